O=gpurun_out
bash tools/kstats.sh r04e_b2048 --batch 2048 --pool 8 --no-pipeline --no-dropin > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04e_b2048.csv | head -14
B="--steps 100 --warmup 20 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin --batch 2048 --pool 8"
python bench.py $B > $O/r04e_b2048.json 2>$O/r04e.err
python bench.py $B --prep dataset > $O/r04e_b2048_ds.json 2>>$O/r04e.err
for f in $O/r04e_*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"; done
