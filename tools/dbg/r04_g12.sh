O=gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_chain_tail.py tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -x -q ) > $O/r04l_gputests.log 2>&1
tail -3 $O/r04l_gputests.log
python tools/phase_step_kernel.py COLLAB 50 2>&1 | grep -E "^it[3]|COLLAB x" | cut -c1-330
python tools/phase_step_kernel.py COLLAB 50 largest 190 2>&1 | grep -E "^it[3]|COLLAB x" | cut -c1-330
B="--steps 400 --warmup 40 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin"
python bench.py $B > $O/r04l_b50.json 2>$O/r04l.err
python bench.py $B --batch 256 --pool 8 > $O/r04l_b256.json 2>>$O/r04l.err
for f in $O/r04l_*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"; done
