O=gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_properties.py tests/test_gpu_configs.py -m gpu -x -q ) > $O/r04s_gputests.log 2>&1
tail -2 $O/r04s_gputests.log
bash tools/kstats.sh r04s_b2048 --batch 2048 --pool 8 --no-pipeline --no-dropin > /dev/null 2>&1; cut -d, -f1-4 $O/kernel_stats_r04s_b2048.csv | grep "chain"
