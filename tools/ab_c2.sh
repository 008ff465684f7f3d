#!/bin/bash
# A/B of library builds on config[2] (tools/bench_configs.py <case>), round-robin on one box.   usage: bash tools/ab_c2.sh <rounds> <case substring> lib1.so lib2.so ...
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; CASE=$2; shift 2
for i in $(seq $N); do
  for L in "$@"; do
    echo "== $L"
    SLODE_LIB_PATH=$L timeout -k 10 200 python tools/bench_configs.py "$CASE" 2>/dev/null | cut -c 1-20,90-400
  done
done
