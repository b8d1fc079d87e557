O=gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r04h_gputests.log 2>&1
tail -3 $O/r04h_gputests.log
bash tools/kstats.sh r04h_b2048_ds --batch 2048 --pool 8 --no-dropin --prep dataset > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04h_b2048_ds.csv | head -14
bash tools/kstats.sh r04h_b2048_pb --batch 2048 --pool 8 --no-dropin > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04h_b2048_pb.csv | head -14
