O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r04a_gputests.log 2>&1
python tools/phase_readout_tail.py > $O/r04a_phase_rt.txt 2>&1
python bench.py --batch 2048 --steps 100 --warmup 20 --pool 8 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin > $O/r04a_b2048_perbatch.json 2>$O/r04a.err
python bench.py --batch 2048 --steps 100 --warmup 20 --pool 8 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin --prep dataset > $O/r04a_b2048_dataset.json 2>>$O/r04a.err
python bench.py --steps 400 --warmup 40 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin --prep dataset > $O/r04a_b50_dataset.json 2>>$O/r04a.err
python tools/epoch_time.py COLLAB 1000 > $O/r04a_epoch.txt 2>&1
tail -3 $O/r04a_gputests.log; cat $O/r04a_phase_rt.txt; cat $O/r04a_epoch.txt; for f in $O/r04a_b*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
