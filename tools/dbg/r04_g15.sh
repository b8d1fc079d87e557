O=gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_dense.py tests/test_gpu_properties.py tests/test_prepared_dataset.py tests/test_gpu_classifier.py -m gpu -x -q ) > $O/r04p_gputests.log 2>&1
tail -3 $O/r04p_gputests.log
B="--steps 100 --warmup 20 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin --batch 2048 --pool 8"
python bench.py $B --prep dataset > $O/r04p_b2048_ds.json 2>$O/r04p.err
python bench.py $B > $O/r04p_b2048_pb.json 2>>$O/r04p.err
for f in $O/r04p_*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
bash tools/kstats.sh r04p_b2048_ds --batch 2048 --pool 8 --no-dropin --prep dataset > /dev/null 2>&1; grep "assemble\|prep" $O/kernel_stats_r04p_b2048_ds.csv | cut -d, -f1-4
bash tools/kstats.sh r04p_b2048_np --batch 2048 --pool 8 --no-dropin --no-pipeline > /dev/null 2>&1; grep "prep" $O/kernel_stats_r04p_b2048_np.csv | cut -d, -f1-4
