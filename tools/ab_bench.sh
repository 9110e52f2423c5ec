#!/bin/bash
# A/B of the timed scan step under environment switches: bash tools/ab_bench.sh "VAR=1" "VAR2=1" ...  ("" = defaults)
mkdir -p gpurun_out
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python bench.py --no-train --no-cpu --no-extras --steps 20 --warmup 3 > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err || { tail -5 gpurun_out/ab_$i.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$i.json").read().strip().splitlines()[-1])
print("[$envs]", "ms_per_step", round(j["ms_per_step"],4), {k: round(v,4) for k,v in j["kernel_ms_per_step"].items()}, "cand frac", round(j["roofline"]["frac"],3), flush=True)
PY
done
