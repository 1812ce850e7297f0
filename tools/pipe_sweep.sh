#!/bin/bash
# host-pipeline tuning sweep (run on the GPU box): batch size x GPU workers x writer threads, FASTA -> FASTA on 1 Gbp in /dev/shm
# columns: build_s correct_s parse_s gpu_s(sum over workers) write_s(sum over writers) batches end_to_end_Gbases/s
rm -f gpurun_out/sweep.log
for rep in 1 2; do
for cfg in "32 2 1" "32 2 2" "32 3 2" "64 2 2" "64 3 2" "128 3 2" "16 3 2"; do
  set -- $cfg
  BRX_PIPE_BATCH_MB=$1 BRX_PIPE_WORKERS=$2 BRX_PIPE_WRITERS=$3 timeout -k 10 200 python tools/e2e_cli.py 100000 200 > gpurun_out/e2e_tmp.log 2>&1 || exit 1
  echo "batch_mb=$1 workers=$2 writers=$3 $(tail -1 gpurun_out/e2e_tmp.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["build_s"], d["correct_s"], d["correct_parse_s"], d["correct_gpu_s_2workers"], d["correct_write_s"], d["batches"], d["end_to_end_gbases_per_s"])')" >> gpurun_out/sweep.log
done; done
cat gpurun_out/sweep.log
