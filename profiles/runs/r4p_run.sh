#!/bin/bash
# round 4: unanswered index probes settled inline in the 64-lane loops + lean reverse passes: parity, bench, methods, configs[4]
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_lane.py -x -q -m gpu 2>&1 | tail -2 || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r4p_bench.json 2> gpurun_out/r4p_bench.err || exit 1
timeout -k 10 300 python tools/method_bench.py 100000 > gpurun_out/r4p_methods.jsonl 2> gpurun_out/r4p_methods.err || exit 1
timeout -k 10 300 python bench.py --config 4 --gpus 1 --steps 2 --no-cpu-baseline 2> gpurun_out/r4p_config4.err > gpurun_out/r4p_config4.json || exit 1
timeout -k 10 200 python tools/fuzz_parity.py 150 2>&1 | tail -1
