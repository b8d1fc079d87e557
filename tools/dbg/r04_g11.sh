O=gpurun_out
bash tools/kstats.sh r04k_b50_np --no-dropin --no-pipeline > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04k_b50_np.csv | head -6
bash tools/kstats.sh r04k_b50_ds --no-dropin --prep dataset > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04k_b50_ds.csv | head -4
