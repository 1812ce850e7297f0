"""A fixed-seed slice of tools/fuzz_parity.py: random small jobs with odd shapes (k, confirm, max_search, abundance,
read lengths around k, error rates, method chains, group widths, index on/off, sparse / lazy sets) through the HIP
path and the oracle; any difference in a set or a corrected read fails."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [7, 8])
def test_random_jobs_match_the_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "12", str(seed)], capture_output=True,
                       text=True, timeout=300)
    tail = "\n".join(r.stdout.splitlines()[-3:])
    assert r.returncode == 0, tail + r.stderr[-500:]
    assert "no mismatch" in tail
    assert r.stdout.count("\nok case") + r.stdout.startswith("ok case") >= 10
