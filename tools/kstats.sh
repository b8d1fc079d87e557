#!/bin/bash
# per-kernel rocprofv3 stats of a bench configuration: tools/kstats.sh <tag> [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --truncate-kernels --output-format csv -d $OUT/ks_$TAG -o p -- \
  python $R/bench.py "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-roofline --large-batch 0 > $OUT/ks_$TAG.log 2>&1
find $OUT/ks_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$TAG.csv \;
rm -rf $OUT/ks_$TAG
cut -d, -f1-4 $OUT/kernel_stats_$TAG.csv | head -16
