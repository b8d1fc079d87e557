#!/bin/bash
# dense-block vs CSR-gather aggregation: step time per batch size, then per-kernel stats of the dense form
# usage (GPU box): bash tools/agg_sweep.sh <tag> [batch sizes...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
BS=${@:-50 256 2048}
: > $OUT/agg_sweep_$TAG.txt
for B in $BS; do
  for A in sparse dense; do
    python bench.py --batch $B --agg $A --steps 200 --warmup 30 --pool 8 --no-cpu-baseline --no-pmc --large-batch 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('B=%5d %-6s %8.1f us/step %10.0f graphs/s | agg kernel %7.2f us frac %.4f' % ($B, '$A', d['ms_per_step']*1e3, d['value'], r['avg_launch_us'], r['frac']))" >> $OUT/agg_sweep_$TAG.txt
  done
done
cat $OUT/agg_sweep_$TAG.txt
cd /tmp && export TMPDIR=/tmp
for B in 50 2048; do
  rocprofv3 --kernel-trace --stats --truncate-kernels --output-format csv -d $OUT/prof_${TAG}_d$B -o p -- \
    python $R/bench.py --batch $B --agg dense --steps 100 --warmup 20 --pool 8 --no-cpu-baseline --no-pmc --no-roofline --large-batch 0 > /dev/null 2>&1
  find $OUT/prof_${TAG}_d$B -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_${TAG}_dense_b$B.csv \;
  rm -rf $OUT/prof_${TAG}_d$B
  echo "== dense B=$B"; cut -d, -f1-4 $OUT/kernel_stats_${TAG}_dense_b$B.csv | head -14
done
