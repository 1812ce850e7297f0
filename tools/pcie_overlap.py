#!/usr/bin/env python3
"""What the host link of the box allows: page-locked 82 MB blocks (one 8192-record batch) up, down, both at once on two
streams, and up under a compute kernel.  Bounds what brx_chain_correct_batch_async can overlap (bench.py host_8192)."""
import time, json
import torch
n = 82 << 20
h_up = torch.empty(n, dtype=torch.uint8).pin_memory()
h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def up():
    with torch.cuda.stream(s1):
        d_a.copy_(h_up, non_blocking=True)
def dn():
    with torch.cuda.stream(s2):
        h_dn.copy_(d_b, non_blocking=True)
def kern():
    with torch.cuda.stream(s3):
        big.add_(1)   # 2 GiB of HBM traffic: ~0.5 ms
res = {"bytes": n, "up_ms": timed(up), "down_ms": timed(dn), "up_and_down_ms": timed(lambda: (up(), dn())),
       "kernel_ms": timed(kern), "up_and_kernel_ms": timed(lambda: (up(), kern())),
       "up_down_kernel_ms": timed(lambda: (up(), dn(), kern()))}
res["up_GBps"] = n / res["up_ms"] / 1e6
res["down_GBps"] = n / res["down_ms"] / 1e6
print(json.dumps(res))
