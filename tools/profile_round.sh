# Round profile on the GPU box: kernel-trace stats of the bench command, PMC HBM traffic and issue counters of the scan kernels.
# usage: bash tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>_*; fold them with tools/summarize_*.py afterwards)
# Counters go in their own passes (--pmc with --kernel-trace only), as the pool requires.
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --train-steps 2 > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  REPS=1 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/${TAG}_pmc_$C -o out --output-format csv -- python3 $R/tools/prof_scan.py > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
done
REPS=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/${TAG}_pmc_sq1 -o out --output-format csv -- python3 $R/tools/prof_scan.py > $R/gpurun_out/${TAG}_pmc_sq1.log 2>&1
REPS=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/${TAG}_pmc_sq2 -o out --output-format csv -- python3 $R/tools/prof_scan.py > $R/gpurun_out/${TAG}_pmc_sq2.log 2>&1
echo profile-done
