#!/bin/bash
# round 4, run j: unit start in one trip (fat UnitDesc): parity, bench, walking methods
timeout -k 10 500 python -m pytest tests/test_gpu_lane.py tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
timeout -k 10 200 python tools/fuzz_parity.py 150 78 2>&1 | tail -1 | cut -c1-200
FUZZ_FOCUS=walklane timeout -k 10 200 python tools/fuzz_parity.py 100 79 2>&1 | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline --no-e2e --steps 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels']; print('bench', d['value'], d['ms_per_step'], 'correct_pass', k['correct_pass'], 'apply', k['lane_apply']['avg_ms'], 'sync', k['lane_sync']['avg_ms'], 'units', k['lane_units']['avg_ms'], d['correct_stats']['fixes'])"
python tools/method_bench.py 100000 one graph gap_size 2>/dev/null | cut -c1-330
