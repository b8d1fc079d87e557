#!/bin/bash
# Round-2 evidence set (GPU box): bench lines, rocprofv3 kernel stats and PMC summaries -> gpurun_out/, to be copied to profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; cd $R
export GIT_HASH=${GIT_HASH:-unknown}
python bench.py > $OUT/r02_bench.json 2> $OUT/r02_bench.err
python bench.py --dtype bf16 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/r02_bench_bf16.json 2>> $OUT/r02_bench.err
python bench.py --dtype bf16 --batch 2048 --steps 100 --warmup 20 --pool 8 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/r02_bench_bf16_b2048.json 2>> $OUT/r02_bench.err
python bench.py --batch 2048 --steps 100 --warmup 20 --pool 8 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/r02_bench_b2048.json 2>> $OUT/r02_bench.err
python bench.py --batch 256 --steps 200 --warmup 20 --pool 8 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/r02_bench_b256.json 2>> $OUT/r02_bench.err
bash tools/kstats.sh r02_b50 > /dev/null
bash tools/kstats.sh r02_b2048 --batch 2048 --pool 8 > /dev/null
bash tools/kstats.sh r02_bf16_b2048 --batch 2048 --pool 8 --dtype bf16 > /dev/null
bash tools/pmc.sh b50 > $OUT/r02_pmc_b50.txt 2>&1
bash tools/pmc.sh b2048 --batch 2048 > $OUT/r02_pmc_b2048.txt 2>&1
bash tools/pmc_sq.sh r02_b2048 "--batch 2048" "k_gcn_fwd32d|k_gcn_bwd32d|k_gcn_fwd_af_d" > $OUT/r02_sq_b2048.txt 2>&1
bash tools/pmc_sq.sh r02_bf16_b2048 "--batch 2048 --dtype bf16" "k_gcn_fwd32d" > $OUT/r02_sq_bf16_b2048.txt 2>&1
bash tools/pmc_sq.sh r02_b50 "" "k_gcn_fwd32|k_gcn_bwd32|k_readout_tail|k_wgrad" > $OUT/r02_sq_b50.txt 2>&1
ls -la $OUT | tail -30
