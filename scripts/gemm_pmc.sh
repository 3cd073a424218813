#!/bin/bash
# Counter passes over scripts/gemm_one.py; medians into gpurun_out/r3/TAG_diag.txt
# usage: scripts/gemm_pmc.sh TAG "M N K" "counters pass 1" ...
set -e
TAG=$1; ARGS=$2
shift 2
OUT=$(pwd)/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
i=0
FILES=""
for C in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$i -- python3 scripts/gemm_one.py $ARGS > $OUT/${TAG}_pass$i.log 2>&1
  F=$(find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' | head -1)
  cp $F $OUT/${TAG}_pass$i.csv
  FILES="$FILES $OUT/${TAG}_pass$i.csv"
done
python3 scripts/pmc_diag.py "gemm_bf16" $FILES > $OUT/${TAG}_diag.txt
rm -f $OUT/${TAG}_pass*.csv
cat $OUT/${TAG}_diag.txt
