#!/bin/bash
# round-3 GPU call 1: new tests, bench line, rocprof cross-check of the per-dispatch clock, N=2 rehearsal, stamp timelines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "dopri5 or config4" --durations=10 > $OUT/t1.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/t1.log
timeout -k 10 300 python bench.py > $OUT/bench1.json 2> $OUT/bench1.err; echo "bench rc=$?"
bash tools/profile_quick.sh a
SLODE_BENCH_REHEARSE=1 timeout -k 10 200 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench_n2.json 2> $OUT/bench_n2.err; echo "rehearse rc=$?"
SLODE_LIB_PATH=structured_latent_odes_amd/libslode_stamps.so timeout -k 10 120 python tools/stamps.py > $OUT/stamps_c1.log 2>&1; echo "stamps rc=$?"
STAMPS_CFG=c2 SLODE_LIB_PATH=structured_latent_odes_amd/libslode_stamps.so timeout -k 10 120 python tools/stamps.py > $OUT/stamps_c2.log 2>&1; echo "stamps c2 rc=$?"
