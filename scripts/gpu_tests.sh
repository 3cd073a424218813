#!/bin/bash
# GPU test run on the box: log under gpurun_out/r3, achieved errors printed behind the summary.
# usage: scripts/gpu_tests.sh [pytest arguments, default: the whole tests directory]
mkdir -p gpurun_out/r3
rm -f gpurun_out/achieved_errors.txt
[ $# -eq 0 ] && set -- tests
timeout -k 10 900 python -u -m pytest "$@" -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/r3/t_all.log 2>&1
RC=$?
tail -15 gpurun_out/r3/t_all.log
[ -f gpurun_out/achieved_errors.txt ] && cat gpurun_out/achieved_errors.txt
exit $RC
