#!/bin/bash
# Round-2 profiles of the metric step (run on the GPU box through gpurun; writes under gpurun_out/r2/prof, summaries are copied into
# profiles/ by tools/profile_r02_collect.py).  Arms of the fused ODE/ELBO kernel (SLODE_ODE_ALG, read once per handle):
#   0 = product (piecewise-linear heads + prefix/suffix-sum contraction), 1 = direct head evaluation, 2 = direct + MFMA contraction.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counter_names.txt
for alg in 0 1 2; do
  SLODE_ODE_ALG=$alg rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_alg$alg -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_alg$alg.json 2> $OUT/bench_alg$alg.err || echo "stats alg $alg failed"
done
for alg in 0 2; do
  SLODE_ODE_ALG=$alg rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -f csv -d $OUT/pmc_sq_alg$alg -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_sq_alg$alg.json 2> $OUT/pmc_sq_alg$alg.err || echo "pmc sq alg $alg failed"
  SLODE_ODE_ALG=$alg rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -f csv -d $OUT/pmc_mfma_alg$alg -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_mfma_alg$alg.json 2> $OUT/pmc_mfma_alg$alg.err || echo "pmc mfma alg $alg failed"
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err || echo "write failed"
find $OUT -name "*.csv" | head -40
du -sh $OUT
