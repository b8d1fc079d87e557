#!/bin/bash
# A/B of the step-kernel candidates (docs/rounds/r06.md): parity tests, quick bench line and phase table per variant library
# usage (GPU box): bash tools/r06_ab.sh wpre1 noeb ...      (variants built by tools/build_variant.sh; `default` is always run first)
exec < /dev/null
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out; mkdir -p $O
Q="--min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin --steps 400 --warmup 40"
for V in default "$@"; do
  if [ $V = default ]; then unset DGCNN_HIP_LIB; else export DGCNN_HIP_LIB=$R/dgcnn_amd/variants/lib_$V.so; fi
  echo "=== $V" >> $O/r06_ab.txt
  timeout 900 python -m pytest tests/test_gpu_chain_tail.py tests/test_gpu_model.py tests/test_gpu_eval_kernel.py -q -x 2>&1 | tail -3 >> $O/r06_ab.txt
  for k in 1 2 3; do
    python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$V  %.2f us/step | kernel %.2f us' % (d['ms_per_step']*1e3, r['avg_launch_us']))" >> $O/r06_ab.txt
  done
  python tools/phase_step_kernel.py COLLAB 50 2>&1 | tail -5 >> $O/r06_ab.txt
done
cat $O/r06_ab.txt
