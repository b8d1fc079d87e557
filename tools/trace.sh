#!/bin/bash
# per-dispatch kernel trace of a few steps (diagnostics): prints the k_wgrad dispatch durations of the last step
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --truncate-kernels --output-format csv -d $OUT/trace -o tr -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# last complete step: find last k_prep_fast_a
idx=[i for i,n in enumerate(names) if n=="k_prep_fast_a"]
s=idx[-2]; e=idx[-1]
t0=int(rows[s]["Start_Timestamp"])
for r in rows[s:e]:
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    print(f'{(st-t0)/1000:8.1f}us +{(en-st)/1000:6.1f}us  {r["Kernel_Name"]:18s} grid={r.get("Grid_Size_X","?")} wg={r.get("Workgroup_Size_X","?")}')
PY
rm -rf $OUT/trace
