#!/bin/bash
# End of round 3 (after the solidity mask of the walking correctors): the bench line, its kernel stats, every corrector at
# 1 Gbp, configs[4]'s share.  Run from the repo root on the GPU box: bash profiles/collect_r3c.sh
O=$PWD/gpurun_out/r3c
mkdir -p $O
R=$PWD
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/stats.log 2>&1
cd $R
timeout -k 10 300 python tools/method_bench.py 100000 > $O/methods_1gbp.jsonl 2>$O/methods.err
timeout -k 10 400 python bench.py --config 4 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/config4_share_n1.json 2>$O/config4.err
ls $O
