"""File objects -> file descriptors for the native host pipelines (brx_run_correction_fd, brx_count_fasta_fd).

A plain file (io.BufferedReader / BufferedWriter / FileIO) is handed over as its descriptor.  Anything else
(gzip / bz2 / xz streams, BytesIO, a reader whose sniffed bytes cannot be pushed back) is bridged through a
pipe by a feeder / collector thread; the ctypes call releases the GIL, so the bridge runs alongside."""
from __future__ import annotations

import io
import os
import threading
from contextlib import contextmanager
from typing import BinaryIO

CHUNK = 4 << 20


def _plain_fd(f, writing: bool):
    kinds = (io.BufferedWriter, io.FileIO) if writing else (io.BufferedReader, io.FileIO)
    if type(f) not in kinds:
        return None
    try:
        fd = f.fileno()
        if writing:
            f.flush()
        elif f.seekable():
            os.lseek(fd, f.tell(), os.SEEK_SET)  # the reader may have buffered ahead of its logical position
        else:
            return None  # e.g. stdin after a peek(): its buffered bytes would be lost
        return fd
    except (OSError, ValueError, io.UnsupportedOperation):
        return None


@contextmanager
def input_fd(f: BinaryIO):
    fd = _plain_fd(f, False)
    if fd is not None:
        yield fd
        return
    r, w = os.pipe()
    err = []

    def feed():
        try:
            with os.fdopen(w, "wb", buffering=0) as wf:
                while True:
                    chunk = f.read(CHUNK)
                    if not chunk:
                        break
                    wf.write(chunk)
        except BrokenPipeError:
            pass  # the consumer stopped early (parse error ends the stream, src/lib.rs:35)
        except Exception as e:  # surfaced after the native call returns
            err.append(e)

    t = threading.Thread(target=feed, daemon=True)
    t.start()
    try:
        yield r
    finally:
        os.close(r)
        t.join()
    if err:
        raise err[0]


@contextmanager
def output_fd(f: BinaryIO):
    fd = _plain_fd(f, True)
    if fd is not None:
        yield fd
        return
    r, w = os.pipe()
    err = []

    def collect():
        try:
            with os.fdopen(r, "rb", buffering=0) as rf:
                while True:
                    chunk = rf.read(CHUNK)
                    if not chunk:
                        break
                    f.write(chunk)
        except Exception as e:
            err.append(e)

    t = threading.Thread(target=collect, daemon=True)
    t.start()
    try:
        yield w
    finally:
        os.close(w)
        t.join()
    if err:
        raise err[0]
