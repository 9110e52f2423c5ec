#!/bin/bash
# scan step time against the workspace bound (cells of a super-batch written by scan_cand and read back by stage_hits:
# do they come back from the 256 MiB Infinity Cache when a super-batch is small enough?)
mkdir -p gpurun_out
for mb in 48 96 144 192 384 768 8192; do
  MOTIFS_WS_LIMIT_MB=$mb python bench.py --no-train --no-cpu --no-extras --steps 20 --warmup 3 > gpurun_out/ws_$mb.json 2> gpurun_out/ws_$mb.err || exit 1
  python - <<PY
import json
j=json.loads(open("gpurun_out/ws_$mb.json").read().strip().splitlines()[-1])
print("ws_limit_mb", $mb, "ms_per_step", round(j["ms_per_step"],4), j["kernel_ms_per_step"], flush=True)
PY
done
