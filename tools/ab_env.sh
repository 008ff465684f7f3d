#!/bin/bash
# A/B of environment switches of ONE library on one box, round-robin.   usage: bash tools/ab_env.sh <rounds> "VAR=val ..." "VAR=val ..." ...   ("-" = no switch)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/ab
N=$1; shift
for i in $(seq $N); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
    env $EV timeout -k 10 200 python bench.py --ab > gpurun_out/ab/ab.json 2>/dev/null
    python - "$E" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab/ab.json").read().strip().splitlines()[-1])
print("%-30s value %.3fM us %.2f" % (sys.argv[1], d["value"]/1e6, 1e3*d["ms_per_step"]), {k: round(v,2) for k,v in d["roofline"]["kernel_us"].items()})
PY
  done
done
