#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for w in COLLAB MUTAG; do
  timeout 300 python bench.py --workload $w --steps 2000 --warmup 200 --no-cpu-baseline --no-dropin --no-pmc --large-batch 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
done
python tools/phase_step_kernel.py COLLAB 50 2>&1 | grep -E "it3"
python tools/phase_step_kernel.py MUTAG 50 2>&1 | grep -E "it3"
