#!/bin/bash
# round 4, run k: one-round-ahead fetch of the next position's index line (BRX_LANE_PREFETCH), A/B + parity
for v in 0 1 0 1; do
  BRX_LANE_PREFETCH=$v python bench.py --no-cpu-baseline --no-e2e --steps 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels']; print('PREFETCH=$v', d['value'], d['ms_per_step'], 'correct_pass', k['correct_pass']['avg_ms'], 'apply', k['lane_apply']['avg_ms'], 'sync', k['lane_sync']['avg_ms'], d['correct_stats']['fixes'])"
done
timeout -k 10 300 python -m pytest tests/test_gpu_lane.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
timeout -k 10 200 python tools/fuzz_parity.py 100 80 2>&1 | tail -1 | cut -c1-200
