import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_fasta(path):
    """minimal FASTA reader for fixtures: returns (names, seqs) as bytes."""
    names, seqs = [], []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                names.append(line[1:])
                seqs.append(bytearray())
            elif seqs:
                seqs[-1] += line
    return names, [bytes(s) for s in seqs]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def raw_reads():
    return read_fasta(os.path.join(GOLDEN, "raw.fasta"))[1]


@pytest.fixture(scope="session")
def solid_fixture_bytes():
    import gzip
    with open(os.path.join(GOLDEN, "raw.k11.a2.solid"), "rb") as f:
        return gzip.decompress(f.read())


@pytest.fixture(scope="session")
def unit_vectors():
    import json
    with open(os.path.join(GOLDEN, "unit_vectors.json")) as f:
        return json.load(f)
