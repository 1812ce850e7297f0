#!/bin/bash
# Round 3's evidence from one GPU box (run from the repo root): bash profiles/collect_r3.sh TAG
#  1 the contract bench line (CPU baseline, FASTA file -> file leg, 8192-record host batches included)
#  2 rocprofv3 --kernel-trace --stats of the same command (per-kernel durations: must agree with the line's timers)
#  3 PMC passes: FETCH_SIZE, WRITE_SIZE (separate runs, --kernel-trace only) -> HBM bytes per pass
#  4 SQ counters of the correction kernels (VALU instructions, busy share, lanes per instruction)
#  5 every corrector at 1 Gbp; BASELINE configs[2], [3]'s and [4]'s per-GPU shares
TAG=${1:-r3a}
O=$PWD/gpurun_out/$TAG
mkdir -p $O
R=$PWD
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/stats.log 2>&1
cd $R
bash profiles/collect_pmc.sh $TAG > $O/pmc.log 2>&1
bash profiles/collect_valu.sh $TAG > $O/valu.log 2>&1
timeout -k 10 300 python tools/method_bench.py 100000 > $O/methods_1gbp.jsonl 2>$O/methods.err
timeout -k 10 300 python bench.py --config 3 --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/config3_share_n1.json 2>$O/config3.err
timeout -k 10 400 python bench.py --config 4 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/config4_share_n1.json 2>$O/config4.err
timeout -k 10 400 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/config2_greedy_10gbp.json 2>$O/config2.err
ls -la $O
