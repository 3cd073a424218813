#!/bin/bash
# Bench line + rocprofv3 kernel stats + the two HBM counter passes of one bench.py workload, on the GPU box.
# usage: scripts/profile_workload.sh TAG "bench.py arguments" ["short arguments for the counter passes"]
# Writes gpurun_out/prof/TAG_{bench.json,kernel_stats.csv,kstats.txt,fetch.csv,write.csv}; every GPU step runs
# under its own time limit and the script stops at the first failure.
set -e
TAG=$1
ARGS=$2
PMC_ARGS=${3:-$ARGS --steps 20 --warmup 5 --stage-reps 5}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py $ARGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
rm -rf /tmp/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/ks -- python3 bench.py $ARGS --no-cpu-baseline > $OUT/${TAG}_ks.log 2>&1
cp $(find /tmp/prof_$TAG/ks -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
python3 scripts/kstats.py $(find /tmp/prof_$TAG/ks -name '*kernel_trace.csv' | head -1) > $OUT/${TAG}_kstats.txt
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_$TAG/f -- python3 bench.py $PMC_ARGS --no-cpu-baseline > $OUT/${TAG}_f.log 2>&1
cp $(find /tmp/prof_$TAG/f -name '*counter_collection.csv' | head -1) $OUT/${TAG}_fetch.csv
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_$TAG/w -- python3 bench.py $PMC_ARGS --no-cpu-baseline > $OUT/${TAG}_w.log 2>&1
cp $(find /tmp/prof_$TAG/w -name '*counter_collection.csv' | head -1) $OUT/${TAG}_write.csv
echo "profiled $TAG"
