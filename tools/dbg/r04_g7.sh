O=gpurun_out
bash tools/kstats.sh r04g_b50 --no-dropin > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04g_b50.csv | head -8
bash tools/kstats.sh r04g_b256 --batch 256 --pool 8 --no-dropin > /dev/null 2>&1
cut -d, -f1-4 $O/kernel_stats_r04g_b256.csv | head -8
