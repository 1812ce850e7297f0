#!/bin/bash
# same-box A/B of libbrx variants on the lane pass: tools/lane_ab.sh "C=512" default w8 ...
spec=$1; shift
eval $spec
for v in "$@"; do
  lib=br_amd/lib/libbrx.so; [ "$v" != default ] && lib=br_amd/lib/ab/libbrx_$v.so
  for rep in 1 2; do
  BRX_LIB_PATH=$lib BRX_LANE_CHUNK=${C:-0} BRX_LANE_SYNC=${R:-4} python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; s=d['correct_stats']
print('$v', '$spec', 'ms/step', d['ms_per_step'], 'correct', d['phases']['correct_ms_per_step'], 'pass_avg', k['correct_pass']['avg_ms'], 'sync', k.get('lane_sync',{}).get('avg_ms'), 'apply', k.get('lane_apply',{}).get('avg_ms'), 'redone', s['lane_redone_reads'])
"
  done
done
