#!/bin/bash
# round 4, run i: guided draws (BRX_LANE_GSS) and graded chunks at the end of the batch (BRX_LANE_TAIL), A/B on the bench step
for g in 0 1; do for t in 0 1; do
  BRX_LANE_GSS=$g BRX_LANE_TAIL=$t python bench.py --no-cpu-baseline --no-e2e --steps 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels']; print('GSS=$g TAIL=$t', d['value'], d['ms_per_step'], 'correct_pass', k['correct_pass'], 'apply', k['lane_apply']['avg_ms'], 'sync', k['lane_sync']['avg_ms'], 'units', k['lane_units']['avg_ms'], d['correct_stats']['lane_units'], d['correct_stats']['fixes'])"
done; done
timeout -k 10 500 python -m pytest tests/test_gpu_lane.py tests/test_gpu_scale.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
timeout -k 10 200 python tools/fuzz_parity.py 120 77 2>&1 | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/r4i_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r4i_bench.json')); print(d['value'], d['host_8192'])"
