set -e
mkdir -p gpurun_out/r1k
timeout -k 10 300 python bench.py > gpurun_out/r1k/bench.log 2>&1
tail -1 gpurun_out/r1k/bench.log > gpurun_out/r1k/bench_n1.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1k/stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1k/stats.log 2>&1
cd $R
timeout -k 10 200 python tools/fuzz_parity.py 90 77 > gpurun_out/r1k/fuzz_parity.log 2>&1
BRX_PIPE_BATCH_MB=1 timeout -k 10 200 python tools/fuzz_fasta.py 45 78 > gpurun_out/r1k/fuzz_fasta.log 2>&1
tail -1 gpurun_out/r1k/fuzz_parity.log; tail -1 gpurun_out/r1k/fuzz_fasta.log
find gpurun_out/r1k/stats -name "*kernel_stats.csv" | head
