#!/bin/bash
# average duration of a kernel's dispatches by position within the step (e.g. the two k_wgrad launches of a large batch):
# tools/ktrace_parity.sh <kernel> <launches per step> [bench args...]
K=$1; P=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --truncate-kernels --output-format csv -d $OUT/kt -o p -- \
  python $R/bench.py "$@" --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-roofline --large-batch 0 > $OUT/kt.log 2>&1
F=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python - "$F" "$K" "$P" <<'PY'
import csv, sys, collections
f, k, p = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"] == k]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2 // p * p:]          # second half: steady state
acc = collections.defaultdict(list)
for i, r in enumerate(rows): acc[i % p].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for j in range(p): print(f"{k} launch {j} of the step: mean {sum(acc[j])/len(acc[j]):.2f} us over {len(acc[j])} dispatches, grid {rows[j]['Grid_Size'] if 'Grid_Size' in rows[j] else '?'}")
PY
rm -rf $OUT/kt
