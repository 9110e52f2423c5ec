# Round profile on the GPU box: kernel-trace stats of the bench command + PMC HBM traffic of the scan kernels.
# usage: bash tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>_*)
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --train-steps 2 > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  REPS=1 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/${TAG}_pmc_$C -o out --output-format csv -- python3 $R/tools/prof_scan.py > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
done
echo profile-done
