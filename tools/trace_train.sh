# kernel-trace of the train-step probe: per-kernel averages (usage on the GPU box: bash tools/trace_train.sh [rows])
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
G=${G:-64} ARENA_GB=90 REPS=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tr_stats -o out --output-format csv -- python3 $R/tools/prof_train.py > $R/gpurun_out/tr.log 2>&1 || exit 1
python3 - $R ${1:-24} <<'PY'
import csv, sys
R, n = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(R + "/gpurun_out/tr_stats/out_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print("%-56s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"].split("(")[0][:56], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total ms", tot / 1e6)
PY
