#!/bin/bash
# HBM traffic of the bench's kernels from the PMC counters, collected the way
# /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots) prescribes:
#   - counters in their own runs, with --kernel-trace only (never with sys/hip/hsa traces);
#   - FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one TCC pass);
#   - gfx950 correction: FETCH_SIZE tallies 64 B per request.  A wide coalesced stream is fetched in
#     128-B requests, so its FETCH_SIZE must be doubled; the calibration pass below (tools/probe_bench)
#     shows that an independent random 4-byte load is ONE request = 64 B as counted (no correction),
#     which is the access pattern of the dominant kernel (correct_pass: one set probe per k-mer).
# Run on the GPU box from the repo root:  bash profiles/collect_pmc.sh rNN
set -e
TAG=${1:-r1}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/probe" -o p -- /root/repo/tools/probe_bench 16 64 > "$OUT/probe.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > "$OUT/write.log" 2>&1
python3 /root/repo/profiles/pmc_summary.py "$OUT" "$TAG" || true
# gpurun only merges gpurun_out/ back: re-run `python3 profiles/pmc_summary.py gpurun_out/pmc_$TAG $TAG` in the work tree
