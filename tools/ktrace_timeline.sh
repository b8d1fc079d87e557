#!/bin/bash
# start offset / duration of every dispatch of ONE steady-state training step (device timestamps of a rocprofv3 kernel trace):
# tools/ktrace_timeline.sh [bench args...]      -- shows what runs beside what when the side stream is in use
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --truncate-kernels --output-format csv -d $OUT/ktl -o p -- \
  python $R/bench.py "$@" --steps 60 --warmup 10 --min-seconds 0.01 --no-cpu-baseline --no-pmc --no-roofline --large-batch 0 > $OUT/ktl.log 2>&1
F=$(find $OUT/ktl -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts at a k_chain_fwd_q / k_chain_readout_tail / k_gcn_fwd* dispatch on the main queue: take the 3rd-last complete step
starts = [i for i, n in enumerate(names) if n in ("k_chain_fwd_q", "k_chain_readout_tail")]
if len(starts) < 5: sys.exit("no steps found")
a, b = starts[-4], starts[-3]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{r['Kernel_Name']:24s} start {s/1000:8.1f} us  end {e/1000:8.1f} us  dur {(e-s)/1000:7.1f} us  queue {r.get('Queue_Id','?')}")
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0)/1000:.1f} us")
PY
rm -rf $OUT/ktl
