#!/bin/bash
# End of round 4: the bench line, its kernel stats (rocprofv3 of the same command), HBM traffic (PMC passes), every
# corrector at 1 Gbp, the other BASELINE configs on one GPU, the big-set runs.  From the repo root on the GPU box:
#   bash profiles/collect_r4.sh
O=$PWD/gpurun_out/${TAG:-r4z}
mkdir -p $O
R=$PWD
echo "== bench (default)"; timeout -k 10 600 python bench.py --steps 10 --warmup 2 > $O/bench.log 2>$O/bench.err; echo "rc=$?"; tail -1 $O/bench.log > $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
echo "== kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/stats.log 2>&1; echo "rc=$?"
cd $R
echo "== pmc"; timeout -k 10 600 bash profiles/collect_pmc.sh ${TAG:-r4z} > $O/pmc.log 2>&1; echo "rc=$?"
echo "== methods"; timeout -k 10 300 python tools/method_bench.py 100000 > $O/methods_1gbp.jsonl 2>$O/methods.err; echo "rc=$?"
echo "== config 2"; timeout -k 10 400 python bench.py --config 2 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/config2_greedy_10gbp.json 2>$O/config2.err; echo "rc=$?"
echo "== config 3 share"; timeout -k 10 400 python bench.py --config 3 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/config3_share_n1.json 2>$O/config3.err; echo "rc=$?"
echo "== config 4 share"; timeout -k 10 400 python bench.py --config 4 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/config4_share_n1.json 2>$O/config4.err; echo "rc=$?"
echo "== big sets"; timeout -k 10 300 python tools/bigset_bench.py 19 800000 one one:fwd > $O/bigset_k19.jsonl 2>/dev/null; echo "rc=$?"
timeout -k 10 300 python tools/bigset_bench.py 21 800000 graph,gap_size > $O/bigset_k21.jsonl 2>/dev/null; echo "rc=$?"
ls $O
