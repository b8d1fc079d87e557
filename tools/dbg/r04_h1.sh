#!/bin/bash
# bf16 leg through the one-launch kernel: tests + bench A/B
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_chain_tail.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -8
for d in ; do
  timeout 300 python bench.py --steps 2000 --warmup 200 --dtype $d --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', d['ms_per_step'], d['value'])"
done

