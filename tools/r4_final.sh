#!/bin/bash
# final evidence run of round 4: full GPU suite, the driver's bench command, profiles, stamp timelines, N=2 rehearsal
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 > $OUT/final_tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/final_tests.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench_driver_cmd.json 2> $OUT/final_bench_driver_cmd.err; echo "bench(driver cmd) rc=$?"
timeout -k 10 400 python bench.py > $OUT/final_bench_default.json 2> $OUT/final_bench_default.err; echo "bench(default) rc=$?"
SLODE_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/final_bench_n2_rehearsal.json 2> $OUT/final_bench_n2_rehearsal.err; echo "rehearse rc=$?"
bash tools/profile_r04.sh > $OUT/final_profile.log 2>&1; echo "profile rc=$?"
# summaries now, on the box (the raw traces are far larger than what gpurun carries back), then the raw output goes
PROFILE_DST=$OUT/profiles_out python tools/profile_r03_collect.py $OUT/prof ${TAG:-r04_h} > $OUT/final_collect.log 2>&1; echo "collect rc=$?"
rm -rf $OUT/prof
for cfg in c1 c2 c2dp5; do
  STAMPS_CFG=$cfg SLODE_LIB_PATH=structured_latent_odes_amd/libslode_stamps.so timeout -k 10 120 python tools/stamps.py > $OUT/final_stamps_$cfg.log 2>&1; echo "stamps $cfg rc=$?"
done
timeout -k 10 300 python tools/batch_sweep.py > $OUT/final_batch_sweep.log 2>&1; echo "sweep rc=$?"
