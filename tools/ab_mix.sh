#!/bin/bash
# A/B of (library, environment) pairs on ONE box in ONE call, round-robin.   usage: bash tools/ab_mix.sh <rounds> "lib.so|VAR=val VAR2=val" ...
# ("-" for the in-tree library / no switch, e.g. "-|-")
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/ab
N=$1; shift
for i in $(seq $N); do
  for P in "$@"; do
    L="${P%%|*}"; E="${P#*|}"
    if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
    if [ "$L" = "-" ]; then LV=""; else LV="SLODE_LIB_PATH=$PWD/$L"; fi
    env $LV $EV timeout -k 10 200 python bench.py --ab > gpurun_out/ab/ab.json 2>/dev/null
    python - "$P" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab/ab.json").read().strip().splitlines()[-1])
print("%-44s value %.3fM us %.2f" % (sys.argv[1], d["value"]/1e6, 1e3*d["ms_per_step"]), {k: round(v,2) for k,v in d["roofline"]["kernel_us"].items()})
PY
  done
done
