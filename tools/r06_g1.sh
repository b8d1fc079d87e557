#!/bin/bash
# first GPU call of round 6: race case on the three builds, the full GPU suite (no -x: every failure listed), smoke, the driver's bench command
exec < /dev/null
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
H=$(cat tools/.head 2>/dev/null || echo unknown)
echo "HEAD=$H" > $O/r06_race_test.txt
for L in libdgcnn_hip.so variants/lib_racedelay.so variants/lib_racedelay_nobar.so; do
  echo "=== DGCNN_HIP_LIB=dgcnn_amd/$L" >> $O/r06_race_test.txt
  DGCNN_HIP_LIB=$PWD/dgcnn_amd/$L timeout 600 python tests/race_case.py 2>&1 | tail -12 >> $O/r06_race_test.txt
  echo "exit=${PIPESTATUS[0]}" >> $O/r06_race_test.txt
done
echo "HEAD=$H" > $O/r06_gputest.txt
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 >> $O/r06_gputest.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 >> $O/r06_gputest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_a.json 2> $O/r06_bench_a.err
cat $O/r06_race_test.txt; tail -40 $O/r06_gputest.txt; cut -c1-700 $O/r06_bench_a.json
