#!/bin/bash
# Shader-side counters of the bench's kernels (issue/wait split of the correction kernel), in their own
# rocprofv3 runs with --kernel-trace only (never combined with sys/hip/hsa traces).
# Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).
# Run on the GPU box from the repo root:  bash profiles/collect_sq.sh TAG [extra bench args]
TAG=${1:-sq}
shift
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d "$OUT/a" -o a -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e "$@" > "$OUT/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAVES \
  --output-format csv -d "$OUT/b" -o b -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e "$@" > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, collections, glob, sys, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/a/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
res = {}
for name, c in agg.items():
    if not any(t in name for t in ("correct_kernel", "one_kernel", "final_count", "scatter", "hash_final", "_hist", "index_insert")):
        continue
    d = {k: sum(v) / len(v) for k, v in c.items()}
    d["ms"] = sum(dur[name]) / max(len(dur[name]), 1)
    if d.get("SQ_BUSY_CYCLES"):
        # 1024 SIMDs; quad-cycle units
        d["valu_busy_frac_of_wave_time"] = d.get("SQ_ACTIVE_INST_VALU", 0) / d.get("SQ_WAVE_CYCLES", 1)
        d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0) / d.get("SQ_WAVE_CYCLES", 1)
    res[name] = d
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, v in res.items():
    print(k, json.dumps({a: (round(b, 3) if b < 100 else int(b)) for a, b in v.items()}))
PY
