#!/bin/bash
# step time + aggregation-kernel time of library variants (tools/build_variant.sh): tools/variant_sweep.sh <batch> <bench args> -- <names...>
B=$1; shift; ARGS=""; while [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for V in default "$@"; do
  if [ $V = default ]; then unset DGCNN_HIP_LIB; else export DGCNN_HIP_LIB=$R/dgcnn_amd/variants/lib_$V.so; fi
  python bench.py --batch $B $ARGS --steps 100 --warmup 20 --pool 8 --no-cpu-baseline --no-pmc --large-batch 0 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s B=%5d %8.1f us/step | agg kernel %7.2f us frac %.4f' % ('$V', $B, d['ms_per_step']*1e3, r['avg_launch_us'], r['frac']))"
done
