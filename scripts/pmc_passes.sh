#!/bin/bash
# Counter passes (one rocprofv3 run each) over one bench.py command; medians per kernel into gpurun_out/r3/TAG_diag.txt.
# usage: scripts/pmc_passes.sh TAG KERNEL_PATTERN "bench.py arguments" "counters of pass 1" ["counters of pass 2" ...]
set -e
TAG=$1; PAT=$2; ARGS=$3
shift 3
OUT=$(pwd)/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
i=0
FILES=""
for C in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  timeout -s KILL 240 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${TAG}_$i -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/${TAG}_pass$i.log 2>&1
  F=$(find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' | head -1)
  cp $F $OUT/${TAG}_pass$i.csv
  FILES="$FILES $OUT/${TAG}_pass$i.csv"
done
python3 scripts/pmc_diag.py "$PAT" $FILES > $OUT/${TAG}_diag.txt
rm -f $OUT/${TAG}_pass*.csv
cat $OUT/${TAG}_diag.txt
