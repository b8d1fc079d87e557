O=gpurun_out
B="--steps 200 --warmup 20 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin --pool 8"
for bs in 128 256 384; do
python bench.py $B --batch $bs > $O/r04r_b${bs}.json 2>$O/r04r.err
DGCNN_HIP_LIB=dgcnn_amd/variants/lib_side128.so python bench.py $B --batch $bs > $O/r04r_b${bs}_side.json 2>>$O/r04r.err
python bench.py $B --batch $bs --prep dataset > $O/r04r_b${bs}_ds.json 2>>$O/r04r.err
python bench.py $B --batch $bs --no-pipeline > $O/r04r_b${bs}_np.json 2>>$O/r04r.err
done
for f in $O/r04r_*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
