#!/bin/bash
# round 4, lean reverse passes (rev_scan_kernel + verify): parity, per-method A/B by BRX_REV_LEAN, configs[4]'s share A/B
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_lane.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for v in 1 0; do
  BRX_REV_LEAN=$v BRX_TRACE=1 timeout -k 10 200 python tools/method_bench.py 100000 graph gap_size > gpurun_out/r4o_methods_lean$v.jsonl 2> gpurun_out/r4o_methods_lean$v.err || exit 1
done
for v in 1 0; do
  BRX_REV_LEAN=$v BRX_TRACE=1 timeout -k 10 300 python bench.py --config 4 --gpus 1 --steps 2 --no-cpu-baseline 2> gpurun_out/r4o_config4_lean$v.err > gpurun_out/r4o_config4_lean$v.json || exit 1
done
timeout -k 10 200 python tools/fuzz_parity.py 150 2>&1 | tail -1
