#!/bin/bash
# Embedding-bag workload lines: optimizers at one id per bag, then multi-hot bags of 4, 20 and 100 ids.
# usage: scripts/embag_lines.sh TAG   -> gpurun_out/prof/TAG_embag_lines.jsonl
set -e
OUT=gpurun_out/prof
mkdir -p $OUT
F=$OUT/$1_embag_lines.jsonl
: > $F
C="--workload embag --no-cpu-baseline --steps 20 --warmup 3 --stage-reps 10"
timeout -k 10 200 python3 bench.py $C --opt sgd >> $F
timeout -k 10 200 python3 bench.py $C --opt adagrad_rowwise >> $F
timeout -k 10 200 python3 bench.py $C --opt adagrad_rowwise --bag-len 4 >> $F
timeout -k 10 200 python3 bench.py $C --opt adagrad_rowwise --bag-len 20 --batch 16384 >> $F
timeout -k 10 200 python3 bench.py $C --opt adagrad_rowwise --bag-len 100 --batch 4096 >> $F
echo "embag lines done"
