#!/bin/bash
# batch-size sweep on the GPU box: fused vs tiled forward family, roofline object per run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for cfg in "256 16 200 20" "2048 3 60 6" "8192 1 20 3"; do
  set -- $cfg
  for path in tiled fused; do
    python bench.py --batch $1 --pool $2 --steps $3 --warmup $4 --path $path --no-cpu-baseline > gpurun_out/sweep_${1}_${path}.json 2> gpurun_out/sweep_${1}_${path}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/sweep_${1}_${path}.json"))
r=d["roofline"]
print("B=$1 path=$path graphs/s=%.0f ms/step=%.3f | %s: %.1f us, %.1f MB -> %.0f GB/s (frac %.3f)" % (d["value"], d["ms_per_step"], r["kernel"][:12], r["avg_launch_us"], r["algorithmic_bytes_per_launch"]/1e6, r["achieved"], r["frac"]))
PY
  done
done
