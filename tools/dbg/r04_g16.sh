O=gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_chain_tail.py tests/test_gpu_chain.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_properties.py -m gpu -x -q ) > $O/r04q_gputests.log 2>&1
tail -3 $O/r04q_gputests.log
python tools/phase_step_kernel.py COLLAB 50 2>&1 | grep -E "^it[3]|COLLAB x" | cut -c1-420
B="--steps 400 --warmup 40 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin"
python bench.py $B > $O/r04q_b50.json 2>$O/r04q.err
python bench.py $B --batch 256 --pool 8 > $O/r04q_b256.json 2>>$O/r04q.err
for f in $O/r04q_*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"; done
bash tools/kstats.sh r04q_b2048 --batch 2048 --pool 8 --no-pipeline --no-dropin > /dev/null 2>&1; cut -d, -f1-4 $O/kernel_stats_r04q_b2048.csv | grep "chain"
