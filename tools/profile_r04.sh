#!/bin/bash
# Round-4 profiles (run on the GPU box through gpurun; raw output under gpurun_out/r4/prof, summaries copied into profiles/ by
# tools/profile_r03_collect.py (arguments: gpurun_out/r4/prof r04_e)).  Kernel-trace statistics and PMC counters always in SEPARATE rocprofv3 runs.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
M="--steps 100 --warmup 20 --ab"
S="--steps 20 --warmup 5 --repeats 5 --ab"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_main -- python3 bench.py $M > $OUT/bench_main.json 2> $OUT/bench_main.err || echo "stats main failed"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_full -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_full.json 2> $OUT/bench_full.err || echo "stats full failed"
for c in "config[0]" "rk4 (fixed" "dopri5" "config[4]"; do
  tag=$(echo "$c" | tr -dc 'a-z0-9')
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_$tag -- python3 tools/bench_configs.py "$c" > $OUT/bench_$tag.jsonl 2> $OUT/bench_$tag.err || echo "stats $tag failed"
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -f csv -d $OUT/pmc_sq -- python3 bench.py $S > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || echo "pmc sq failed"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -f csv -d $OUT/pmc_mfma -- python3 bench.py $S > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err || echo "pmc mfma failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_fetch -- python3 bench.py $S > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_write -- python3 bench.py $S > $OUT/pmc_write.json 2> $OUT/pmc_write.err || echo "write failed"
find $OUT -name "*stats.csv" | head -20
du -sh $OUT
