#!/bin/bash
# Quick issue-side picture of the correction kernels: VALU instructions, VALU-busy share of the SIMDs, lanes per
# instruction, per kernel instance.  One rocprofv3 --pmc pass with --kernel-trace only.
# Run on the GPU box from the repo root:  bash profiles/collect_valu.sh TAG [extra bench args]
TAG=${1:-valu}
shift
OUT=$PWD/gpurun_out/valu_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAVES \
  --output-format csv -d "$OUT/a" -o a -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e "$@" > "$OUT/a.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, collections, glob, sys, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def nm(s): return s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
for f in glob.glob(out + "/a/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[nm(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/a/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[nm(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
line = [l for l in open(out + "/a.log") if l.startswith("{")]
stats = json.loads(line[-1])["correct_stats"] if line else {}
res = {"correct_stats": stats}
for name, c in agg.items():
    if not any(t in name for t in ("correct_kernel", "one_kernel", "lane_")):
        continue
    d = {k: sum(v) / len(v) for k, v in c.items()}
    ms = sum(dur[name]) / max(len(dur[name]), 1)
    quad = 1024 * ms * 1e-3 * 2.4e9 / 4
    res[name] = {"ms": round(ms, 3), "launches": len(dur[name]), "valu_insts_G": round(d["SQ_INSTS_VALU"] / 1e9, 3),
                 "salu_insts_G": round(d["SQ_INSTS_SALU"] / 1e9, 3), "lds_insts_M": round(d["SQ_INSTS_LDS"] / 1e6, 1),
                 "valu_busy_of_simd": round(d["SQ_ACTIVE_INST_VALU"] / quad, 3),
                 "lanes_per_valu": round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"], 1),
                 "waves": int(d["SQ_WAVES"]), "wait_any_frac": round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3)}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
