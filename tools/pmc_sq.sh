#!/bin/bash
# SQ / LDS / TCP counters of chosen kernels (one rocprofv3 --pmc pass per counter group, kernel-trace only)
# usage (GPU box): bash tools/pmc_sq.sh <tag> "<bench args>" "<kernel name regex>"
TAG=$1; ARGS=$2; KRE=$3
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CGRP=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
)
i=0
for G in "${CGRP[@]}"; do
  rocprofv3 --pmc $G --kernel-trace --truncate-kernels --output-format csv -d $OUT/sq_${TAG}_$i -o p -- \
      python $R/bench.py $ARGS --no-cpu-baseline --no-roofline --no-pmc --large-batch 0 --steps 30 --warmup 10 --pool 4 > $OUT/sq_${TAG}_$i.log 2>&1
  i=$((i+1))
done
python - "$OUT" "$TAG" "$KRE" <<'PY'
import csv, glob, sys, re, collections, json
out, tag, kre = sys.argv[1], sys.argv[2], re.compile(sys.argv[3])
res = collections.defaultdict(dict)
for f in glob.glob(f"{out}/sq_{tag}_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not kre.search(k): continue
        a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in acc.items():
        res[k][c] = v / n
json.dump(res, open(f"{out}/sq_{tag}.json", "w"), indent=1)
for k, d in sorted(res.items()):
    print("==", k)
    for c, v in sorted(d.items()): print(f"   {c:32s} {v:16.1f}")
PY
for d in $OUT/sq_${TAG}_[0-9]*; do [ -d "$d" ] && rm -rf "$d"; done
