#!/bin/bash
# HBM traffic counters of the hot kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slot limits,
# /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only.
# usage: tools/pmc.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for C in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  rocprofv3 --pmc $C --kernel-trace --truncate-kernels --output-format csv -d $OUT/pmc_${TAG}_$C -o p -- \
      python $R/bench.py "$@" --no-cpu-baseline --no-roofline --no-pmc --large-batch 0 --steps 60 --warmup 10 --pool 8 > $OUT/pmc_${TAG}_$C.log 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, json, collections
out, tag = sys.argv[1], sys.argv[2]
res = {}
import os
for c in os.environ.get("PMC_COUNTERS", "FETCH_SIZE WRITE_SIZE").split():
    f = glob.glob(f"{out}/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file for", c); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != c: continue
        k = r["Kernel_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in acc.items():
        res.setdefault(k, {})[c] = v / n
        res[k]["dispatches"] = n
res["_git"] = os.environ.get("GIT_HASH", "unknown")
json.dump(res, open(f"{out}/pmc_{tag}.json", "w"), indent=1)
for k, d in sorted((k, d) for k, d in res.items() if isinstance(d, dict)):
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        fs, wsz = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
        print(f"{k:22s} n={d['dispatches']:5d} FETCH_SIZE={fs:10.1f} KB  WRITE_SIZE={wsz:10.1f} KB  -> (2*F+W)*1024 = {(2*fs+wsz)*1024/1e6:8.2f} MB")
    else:
        print(f"{k:22s} " + " ".join(f"{c}={v:.1f}" for c, v in d.items()))
PY
for C in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do rm -rf $OUT/pmc_${TAG}_$C; done
