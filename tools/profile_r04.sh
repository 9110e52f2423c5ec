# Round-4 profile on the GPU box (counters in their own passes, --pmc with --kernel-trace only, as the pool requires):
#   kernel-trace stats of the bench command; HBM traffic + issue counters of the scan kernels at configs[1] (prof_scan.py), of the
#   chunk-group kernels at the configs[4] / configs[3] bank shapes (prof_shard.py), and of the 64-mini-batch train step (prof_train.py).
# usage: bash tools/profile_r04.sh [parts]   parts: any of "stats scan shard train" (default all); outputs under gpurun_out/r04_*
set -e
PARTS=${1:-"stats scan shard train"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
pass() {  # pass <outdir> <counters> <driver> : env passes through
  rocprofv3 --pmc $2 --kernel-trace -d $R/gpurun_out/$1 -o out --output-format csv -- python3 $R/tools/$3 > $R/gpurun_out/$1.log 2>&1
}
for P in $PARTS; do
  case $P in
    stats) rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_stats -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --train-steps 2 > $R/gpurun_out/r04_bench_under_rocprof.log 2>&1 ;;
    scan) export REPS=1; pass r04_pmc_FETCH_SIZE FETCH_SIZE prof_scan.py; pass r04_pmc_WRITE_SIZE WRITE_SIZE prof_scan.py; pass r04_pmc_sq1 "$SQ1" prof_scan.py; pass r04_pmc_sq2 "$SQ2" prof_scan.py ;;
    shard) export REPS=1; for C in 4 3; do export CFG=$C; pass r04_cg${C}_FETCH_SIZE FETCH_SIZE prof_shard.py; pass r04_cg${C}_WRITE_SIZE WRITE_SIZE prof_shard.py; pass r04_cg${C}_sq1 "$SQ1" prof_shard.py; pass r04_cg${C}_sq2 "$SQ2" prof_shard.py; done ;;
    train) export G=64 ARENA_GB=90 REPS=0; pass r04_train_FETCH_SIZE FETCH_SIZE prof_train.py; pass r04_train_WRITE_SIZE WRITE_SIZE prof_train.py; pass r04_train_sq1 "$SQ1" prof_train.py; pass r04_train_sq2 "$SQ2" prof_train.py ;;
  esac
  echo "part $P done"
done
echo profile-done
