#!/bin/bash
# per-dispatch kernel trace of a few steps (diagnostics): prints the dispatches of the last complete step
# (DG_WGRAD_SPLIT=1 bash tools/trace.sh shows the weight-gradient kernel one segment per launch)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --truncate-kernels --output-format csv -d $OUT/trace -o tr -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# last complete step: from one conv1 launch to the next (first kernel of a pipelined step)
first="k_gcn_fwd_af" if "k_gcn_fwd_af" in names else "k_prep_fast_a"
idx=[i for i,n in enumerate(names) if n==first]
s=idx[-2]; e=idx[-1]
t0=int(rows[s]["Start_Timestamp"])
for r in rows[s:e]:
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    print(f'{(st-t0)/1000:8.1f}us +{(en-st)/1000:6.1f}us  {r["Kernel_Name"]:18s} grid={r.get("Grid_Size_X","?")} wg={r.get("Workgroup_Size_X","?")}')
PY
rm -rf $OUT/trace
