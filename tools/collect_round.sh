# One round's evidence from one GPU box: bench line, rocprofv3 kernel stats of the same command, fuzz runs.
# usage (repo root, on the GPU box): bash tools/collect_round.sh TAG
set -e
TAG=${1:-r2}
mkdir -p gpurun_out/$TAG
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/$TAG/bench.log 2>&1
tail -1 gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench_n1.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $R/gpurun_out/$TAG/stats.log 2>&1
cd $R
timeout -k 10 200 python tools/fuzz_parity.py 90 77 > gpurun_out/$TAG/fuzz_parity.log 2>&1
BRX_PIPE_BATCH_MB=1 timeout -k 10 200 python tools/fuzz_fasta.py 45 78 > gpurun_out/$TAG/fuzz_fasta.log 2>&1
tail -1 gpurun_out/$TAG/fuzz_parity.log; tail -1 gpurun_out/$TAG/fuzz_fasta.log
find gpurun_out/$TAG/stats -name "*kernel_stats.csv" | head
