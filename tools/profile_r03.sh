#!/bin/bash
# Round-3 evidence set (GPU box): bench lines, rocprofv3 kernel stats and PMC summaries -> gpurun_out/, to be copied to profiles/
# usage: GIT_HASH=<hash> bash tools/profile_r03.sh [tag]     (tag defaults to r03)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; cd $R
T=${1:-r03}
export GIT_HASH=${GIT_HASH:-unknown}
python bench.py > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err
python bench.py --batch 2048 --steps 100 --warmup 20 --pool 8 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/${T}_bench_b2048.json 2>> $OUT/${T}_bench.err
python bench.py --batch 256 --steps 200 --warmup 20 --pool 8 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/${T}_bench_b256.json 2>> $OUT/${T}_bench.err
for W in MUTAG PROTEINS DD; do
  python bench.py --workload $W --steps 200 --warmup 20 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/${T}_bench_$(echo $W | tr A-Z a-z).json 2>> $OUT/${T}_bench.err
done
python bench.py --workload DD --stress-nodes 5748 --steps 200 --warmup 20 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/${T}_bench_dd_stress.json 2>> $OUT/${T}_bench.err
python bench.py --dtype bf16 --steps 400 --warmup 40 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/${T}_bench_bf16.json 2>> $OUT/${T}_bench.err
python bench.py --dtype bf16 --batch 2048 --steps 100 --warmup 20 --pool 8 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 > $OUT/${T}_bench_bf16_b2048.json 2>> $OUT/${T}_bench.err
python bench.py --batch 2048 --steps 100 --warmup 20 --pool 8 --min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-pipeline > $OUT/${T}_bench_b2048_nopipeline.json 2>> $OUT/${T}_bench.err
bash tools/kstats.sh ${T}_b50 > /dev/null
bash tools/kstats.sh ${T}_dd_stress --workload DD --stress-nodes 5748 > /dev/null
bash tools/kstats.sh ${T}_b2048_nopipeline --batch 2048 --pool 8 --no-pipeline > /dev/null
bash tools/kstats.sh ${T}_b2048 --batch 2048 --pool 8 > /dev/null
bash tools/pmc.sh ${T}_b50 > $OUT/${T}_pmc_b50.txt 2>&1
bash tools/pmc.sh ${T}_b2048 --batch 2048 > $OUT/${T}_pmc_b2048.txt 2>&1
bash tools/pmc_sq.sh ${T}_b2048 "--batch 2048" "k_chain_fwd_q|k_chain_bwd_a|k_chain_bwd_b|k_classifier|k_readout_fwd|k_tail_bwd_walk|k_wgrad" > $OUT/${T}_sq_b2048.txt 2>&1
bash tools/pmc_sq.sh ${T}_b50 "" "k_chain_readout_tail|k_gcn_bwd32|k_gcn_bwd1|k_wgrad" > $OUT/${T}_sq_b50.txt 2>&1
bash tools/pmc_sq.sh ${T}_bf16_b50 "--dtype bf16" "k_chain_fwd_q|k_readout_tail|k_gcn_bwd32|k_gcn_bwd1|k_wgrad" > $OUT/${T}_sq_bf16_b50.txt 2>&1
ls -la $OUT | tail -30
