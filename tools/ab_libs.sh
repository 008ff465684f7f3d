#!/bin/bash
# A/B of several builds of libslode.so on ONE box in ONE call (box-to-box spread is 0.5 us on the dominant kernel; the step time repeats to
# +-0.05 us on one box): round-robin over the libraries.   usage: bash tools/ab_libs.sh <rounds> <libA.so> <libB.so> ...
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/ab
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    SLODE_LIB_PATH=$PWD/$L timeout -k 10 200 python bench.py --ab > gpurun_out/ab/ab.json 2>/dev/null
    python - $L <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab/ab.json").read().strip().splitlines()[-1])
print("%-46s value %.3fM us %.2f" % (sys.argv[1], d["value"]/1e6, 1e3*d["ms_per_step"]), {k: round(v,2) for k,v in d["roofline"]["kernel_us"].items()})
PY
  done
done
