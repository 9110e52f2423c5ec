# PMC passes over tools/prof_scan.py (run on the GPU box): [TAG=name] bash tools/pmc_scan.sh "CTR CTR ..." ["CTR ..." ...]
# Counters in their own runs (--kernel-trace only), one rocprofv3 pass per argument; results in gpurun_out/pmc_<TAG><i>/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "$@"; do
  i=$((i+1))
  REPS=${REPS:-1} rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_${TAG}$i -o out --output-format csv -- python3 $R/tools/prof_scan.py > $R/gpurun_out/pmc_${TAG}$i.log 2>&1 || exit 1
done
echo done
