#!/bin/bash
# round 4, run f: where a host batch spends its time; configs[4]'s share with / without Graph's reverse pass in lane form
BRX_TRACE=2 python tools/host_async_trace.py 3 2> gpurun_out/r4f_trace3.err
grep -v amdgpu.ids gpurun_out/r4f_trace3.err | cut -c1-200 | tail -24
BRX_TRACE=0 python tools/host_async_trace.py 2 2>/dev/null
BRX_TRACE=0 python tools/host_async_trace.py 4 2>/dev/null
for v in 0 1; do
  BRX_LANE_REV=$v python bench.py --config 4 --gpus 1 --steps 2 --no-cpu-baseline 2>/dev/null > gpurun_out/r4f_config4_rev$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r4f_config4_rev$v.json')); print('config4 LANE_REV=$v', d['value'], d['ms_per_step'], {k:(v['avg_ms'],v['launches']) for k,v in d['kernels'].items() if k.startswith('lane') or k.startswith('correct') or k.startswith('succ')})"
done
python bench.py --no-cpu-baseline --no-e2e --steps 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('AP_GRID 1M', d['value'], d['kernels']['lane_apply'])"
