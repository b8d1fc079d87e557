#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
Q="--no-cpu-baseline --no-dropin --no-pmc --large-batch 0"
for b in 50 256 1024; do
  timeout 300 python bench.py --batch $b --pool 8 --steps 1000 --warmup 100 $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b', d['ms_per_step'], d['value'])"
done
bash tools/kstats.sh h5_b50 --no-dropin | grep -E "k_wgrad|k_chain_readout"
bash tools/kstats.sh h5_b256 --batch 256 --pool 8 --no-dropin | grep -E "k_wgrad|k_chain_readout"
