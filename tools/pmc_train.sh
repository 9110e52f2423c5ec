# PMC passes over tools/prof_train.py (run on the GPU box): bash tools/pmc_train.sh "CTR CTR ..." ["CTR ..." ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "$@"; do
  i=$((i+1))
  G=64 ARENA_GB=90 REPS=0 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmct_$i -o out --output-format csv -- python3 $R/tools/prof_train.py > $R/gpurun_out/pmct_$i.log 2>&1 || exit 1
done
echo done
