#!/bin/bash
# A/B of the FM step under environment switches: prints one line per variant and batch.
# usage: scripts/ab_fm.sh "VAR=1 VAR2=1" ... (each argument one variant's environment; "" = defaults)
OUT=gpurun_out/r3
mkdir -p $OUT
i=0
for V in "$@"; do
  i=$((i+1))
  for b in ${BATCHES:-8192 65536}; do
    env $V timeout -k 10 300 python3 bench.py --batch $b --no-cpu-baseline ${BENCH_ARGS} > $OUT/ab_${i}_b$b.json 2>$OUT/ab_${i}_b$b.err || { tail -5 $OUT/ab_${i}_b$b.err; exit 1; }
    python3 - "$V" $b $OUT/ab_${i}_b$b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print("[%s] B=%s %.1f M/s step %.2f us |" % (sys.argv[1], sys.argv[2], d["value"] / 1e6, d["ms_per_step"] * 1e3),
      " | ".join("%s %.2f" % (s["name"][:14], s["ms"] * 1e3) for s in d["stages"]), "| frac", d["roofline"]["frac"])
PY
  done
done
