set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-big --no-train --no-cpu --no-extras > $R/gpurun_out/r5_scan_sdwa.json 2>/dev/null
python3 -c "
import json;o=json.loads(open('$R/gpurun_out/r5_scan_sdwa.json').read().strip().splitlines()[-1]);print(o['ms_per_step'],o['ms_per_step_default_mode'],o['kernel_ms_per_step'])"
rm -rf /tmp/g1t /tmp/g1g
rocprofv3 --kernel-trace --output-format csv -d /tmp/g1g -- python3 $R/tools/g1_step.py --steps 30 > $R/gpurun_out/r5_g1_graph.log 2>&1
f=$(find /tmp/g1g -name "*kernel_trace.csv" | head -1)
python3 $R/tools/g1_sequence.py $f 2 > $R/gpurun_out/r5_g1_sequence_graph.txt
MOTIFS_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/g1t -- python3 $R/tools/g1_step.py --steps 30 > $R/gpurun_out/r5_g1_nograph.log 2>&1
f=$(find /tmp/g1t -name "*kernel_trace.csv" | head -1)
python3 $R/tools/g1_sequence.py $f 2 > $R/gpurun_out/r5_g1_sequence_nograph.txt
tail -1 $R/gpurun_out/r5_g1_sequence_graph.txt; tail -1 $R/gpurun_out/r5_g1_sequence_nograph.txt
grep ms_per_step $R/gpurun_out/r5_g1_graph.log $R/gpurun_out/r5_g1_nograph.log
