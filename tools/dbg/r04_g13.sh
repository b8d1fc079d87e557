O=gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_chain_tail.py tests/test_gpu_model.py tests/test_gpu_classifier.py tests/test_gpu_kernels.py -m gpu -x -q ) > $O/r04m_gputests.log 2>&1
tail -2 $O/r04m_gputests.log
bash tools/kstats.sh r04m_b50 --no-dropin --prep dataset > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04m_b50.csv | head -3
bash tools/kstats.sh r04m_b2048 --no-dropin --batch 2048 --pool 8 --no-pipeline > /dev/null 2>&1
grep wgrad $O/kernel_stats_r04m_b2048.csv | cut -d, -f1-4
B="--steps 400 --warmup 40 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin"
python bench.py $B > $O/r04m_b50.json 2>$O/r04m.err
for f in $O/r04m_*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
