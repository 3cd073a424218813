#!/bin/bash
# The config-2 bench lines the measurement table asks for beside the headline: seeds 0/1/2 with uniform ids,
# Zipf(1.05) ids, and the same step at batch 65536.  One JSON line each into gpurun_out/prof/fm_lines.jsonl.
set -e
OUT=gpurun_out/prof
mkdir -p $OUT
: > $OUT/fm_lines.jsonl
for s in 0 1 2; do
  timeout -k 10 200 python3 bench.py --seed $s --no-cpu-baseline >> $OUT/fm_lines.jsonl
done
for s in 0 1 2; do
  timeout -k 10 200 python3 bench.py --seed $s --dist zipf --no-cpu-baseline >> $OUT/fm_lines.jsonl
done
timeout -k 10 300 python3 bench.py --batch 65536 --no-cpu-baseline >> $OUT/fm_lines.jsonl
timeout -k 10 300 python3 bench.py --batch 65536 --dist zipf --no-cpu-baseline >> $OUT/fm_lines.jsonl
echo "fm lines done"
