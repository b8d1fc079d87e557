#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace stats of the same command.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python bench.py "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --truncate-kernels --output-format csv -d $OUT/prof_$TAG -o $TAG -- \
    python $R/bench.py "$@" --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/prof_$TAG.log 2>&1
find $OUT/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$TAG.csv \;
# keep the merge small: drop the full per-dispatch trace
find $OUT/prof_$TAG -name "*kernel_trace.csv" -delete
ls -la $OUT
