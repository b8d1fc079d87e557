O=gpurun_out
bash tools/kstats.sh r04j_b2048_ds_np --batch 2048 --pool 8 --no-dropin --prep dataset --no-pipeline > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04j_b2048_ds_np.csv | head -12
