#!/bin/bash
# Per-kernel averages of any counter set for kernels whose name matches a pattern: one rocprofv3 --pmc pass with
# --kernel-trace only per counter group (groups separated by ':').
#   bash profiles/collect_counters.sh TAG 'l1_scatter|l1_hist' 'SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY:WRITE_SIZE:FETCH_SIZE' [bench args]
TAG=$1; PAT=$2; GROUPS_=$3; shift 3
OUT=$PWD/gpurun_out/ctr_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
IFS=':' read -ra GR <<< "$GROUPS_"
for g in "${GR[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$OUT/p$i" -o a -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e "$@" > "$OUT/p$i.log" 2>&1 || exit 1
done
python3 - "$OUT" "$PAT" <<'PY'
import csv, collections, glob, sys, re, json
out, pat = sys.argv[1], re.compile(sys.argv[2])
def nm(s): return s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = nm(r["Kernel_Name"])
        if pat.search(n):
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = nm(r["Kernel_Name"])
        if pat.search(n):
            dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
res = {}
for n, c in agg.items():
    res[n] = {"ms_under_pmc": round(sum(dur[n]) / max(len(dur[n]), 1), 3), "launches": len(dur[n])}
    res[n].update({k: sum(v) / len(v) for k, v in c.items()})
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
