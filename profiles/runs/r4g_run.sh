#!/bin/bash
# round 4, run g: kernel durations of ONE 8192-record batch against the full-size set (rocprofv3 kernel trace)
O=$PWD/gpurun_out/r4g
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
METHOD_BENCH_BATCH=8192 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/tools/method_bench.py 100000 one > $O/stats.log 2>&1
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows:
    n=r["Name"]
    if any(t in n for t in ("lane_", "one_kernel", "compact", "scan")):
        print(n[:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
