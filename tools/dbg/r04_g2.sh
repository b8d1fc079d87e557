O=gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r04b_gputests.log 2>&1
tail -15 $O/r04b_gputests.log
B="--steps 400 --warmup 40 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin"
python bench.py $B > $O/r04b_b50.json 2>$O/r04b.err
DGCNN_STEP_KERNEL=0 python bench.py $B > $O/r04b_b50_old.json 2>>$O/r04b.err
python bench.py $B --batch 256 --pool 8 > $O/r04b_b256.json 2>>$O/r04b.err
DGCNN_STEP_KERNEL=0 python bench.py $B --batch 256 --pool 8 > $O/r04b_b256_old.json 2>>$O/r04b.err
for W in MUTAG PROTEINS DD; do python bench.py $B --workload $W > $O/r04b_$W.json 2>>$O/r04b.err; done
for f in $O/r04b_*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"; done
