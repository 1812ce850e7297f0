#!/bin/bash
# HBM read traffic (FETCH_SIZE, KiB, 64 B per request on gfx950) of every kernel of one bench step, one rocprofv3 --pmc
# pass with --kernel-trace only.  Run on the GPU box from the repo root:  bash profiles/collect_fetch.sh TAG [bench args]
TAG=${1:-fetch}
shift
OUT=$PWD/gpurun_out/fetch_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/a" -o a -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e "$@" > "$OUT/a.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
def nm(s): return s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
for f in glob.glob(out + "/a/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[nm(r["Kernel_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-45s launches %3d  fetched %8.3f GB per launch (as counted) = %.3f G requests of 64 B" % (k, len(v), sum(v) / len(v) * 1024 / 1e9, sum(v) / len(v) * 1024 / 64 / 1e9))
PY
