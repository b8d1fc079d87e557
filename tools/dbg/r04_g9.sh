O=gpurun_out
B="--steps 100 --warmup 20 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin --batch 2048 --pool 8"
python bench.py $B --prep dataset > $O/r04i_b2048_ds.json 2>$O/r04i.err || tail -5 $O/r04i.err
python bench.py $B > $O/r04i_b2048_pb.json 2>>$O/r04i.err
python bench.py --steps 400 --warmup 40 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin --prep dataset > $O/r04i_b50_ds.json 2>>$O/r04i.err || tail -5 $O/r04i.err
for f in $O/r04i_*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['config']['prep'])"; done
bash tools/kstats.sh r04i_b2048_ds --batch 2048 --pool 8 --no-dropin --prep dataset > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04i_b2048_ds.csv | head -14
