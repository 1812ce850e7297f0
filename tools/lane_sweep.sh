#!/bin/bash
# same-box sweep of the lane pass's knobs at configs[1]: prints ms per step and the lane kernels' times
for spec in "C=0 R=4" "C=256 R=4" "C=512 R=4" "C=768 R=4" "C=2048 R=4" "C=512 R=2" "C=512 R=8" "C=1024 R=2"; do
  eval $spec
  BRX_LANE_CHUNK=$C BRX_LANE_SYNC=$R python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; s=d['correct_stats']
print('$spec', 'ms/step', d['ms_per_step'], 'correct', d['phases']['correct_ms_per_step'], 'pass_avg', k['correct_pass']['avg_ms'], 'sync', k['lane_sync']['avg_ms'], 'apply', k['lane_apply']['avg_ms'], 'units', s['lane_units'], 'redone', s['lane_redone_reads'], 'rounds', s['rounds'])
"
done
