#!/bin/bash
# BASELINE configs[2] (10 Gbp, k=19, correct::greedy on one GPU): kernel stats and HBM traffic of the same command.
# Counters in their own rocprofv3 runs with --kernel-trace only; FETCH_SIZE and WRITE_SIZE in separate passes.
# Run on the GPU box from the repo root:  bash profiles/collect_config3.sh TAG
TAG=${1:-r1}
R=$PWD
OUT=$R/gpurun_out/c3_$TAG
mkdir -p "$OUT"
ARGS="--reads 1000000 --method greedy --steps 1 --warmup 1 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 $R/bench.py $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 $R/bench.py $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 $R/bench.py $ARGS > "$OUT/write.log" 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, sys, json, collections
out = sys.argv[1]
def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg
f = per_kernel(out + "/fetch/f_counter_collection.csv", "FETCH_SIZE")
w = per_kernel(out + "/write/w_counter_collection.csv", "WRITE_SIZE")
res = {}
for name in f:
    if "correct_kernel" in name or "compact" in name:
        # the timed step is the LAST launches of the run (warmup first): keep the last two correct passes
        fv, wv = f[name][-2:], w.get(name, [0, 0])[-2:]
        # units as in profiles/pmc_summary.py: KiB; random-probe kernels are used as counted (64 B per request)
        res[name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]] = {
            "launches_kept": len(fv), "fetch_bytes": [v * 1024 for v in fv], "write_bytes": [v * 1024 for v in wv],
            "hbm_bytes_per_launch": (sum(fv) / len(fv) + sum(wv) / max(len(wv), 1)) * 1024}
json.dump(res, open(out + "/pmc_raw.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
