#!/bin/bash
# per-kernel rocprofv3 stats of ANY command: tools/kstats_cmd.sh <tag> <command ...>   -> gpurun_out/kernel_stats_<tag>.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
( cd $R && timeout 400 rocprofv3 --kernel-trace --stats --truncate-kernels --output-format csv -d $OUT/ks_$TAG -o p -- "$@" ) > $OUT/ks_$TAG.log 2>&1 < /dev/null
F=$(find $OUT/ks_$TAG -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$F" ]; then cp "$F" $OUT/kernel_stats_$TAG.csv; cut -d, -f1-4,6,7 $OUT/kernel_stats_$TAG.csv | head -12; else echo "no kernel stats for $TAG"; tail -5 $OUT/ks_$TAG.log; fi
rm -rf $OUT/ks_$TAG
