#!/bin/bash
exec < /dev/null
# Round-6 evidence set (GPU box): bench lines, rocprofv3 kernel stats and PMC summaries -> gpurun_out/, to be copied to profiles/
# usage: GIT_HASH=<hash> bash tools/profile_r06.sh [tag]     (tag defaults to r05)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; cd $R
T=${1:-r06}
export GIT_HASH=${GIT_HASH:-unknown}
Q="--min-seconds 1 --no-cpu-baseline --no-pmc --large-batch 0 --no-dropin"
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err          # the driver's command
python bench.py --batch 2048 --steps 100 --warmup 20 --pool 8 $Q > $OUT/${T}_bench_b2048.json 2>> $OUT/${T}_bench.err
python bench.py --batch 2048 --steps 100 --warmup 20 --pool 8 $Q --prep dataset > $OUT/${T}_bench_b2048_dataset.json 2>> $OUT/${T}_bench.err
python bench.py --batch 2048 --steps 100 --warmup 20 --pool 8 $Q --no-pipeline > $OUT/${T}_bench_b2048_nopipeline.json 2>> $OUT/${T}_bench.err
python bench.py --batch 256 --steps 200 --warmup 20 --pool 8 $Q > $OUT/${T}_bench_b256.json 2>> $OUT/${T}_bench.err
python bench.py --steps 400 --warmup 40 $Q --prep dataset > $OUT/${T}_bench_dataset.json 2>> $OUT/${T}_bench.err
DGCNN_STEP_KERNEL=0 python bench.py --steps 400 --warmup 40 $Q > $OUT/${T}_bench_r03form.json 2>> $OUT/${T}_bench.err
for W in MUTAG PROTEINS DD IMDB; do
  python bench.py --workload $W --steps 200 --warmup 20 $Q > $OUT/${T}_bench_$(echo $W | tr A-Z a-z).json 2>> $OUT/${T}_bench.err
done
python bench.py --workload DD --stress-nodes 5748 --steps 200 --warmup 20 $Q > $OUT/${T}_bench_dd_stress.json 2>> $OUT/${T}_bench.err
python bench.py --dtype bf16 --steps 400 --warmup 40 $Q > $OUT/${T}_bench_bf16.json 2>> $OUT/${T}_bench.err
python bench.py --dtype bf16 --batch 2048 --steps 100 --warmup 20 --pool 8 $Q > $OUT/${T}_bench_bf16_b2048.json 2>> $OUT/${T}_bench.err
bash tools/kstats.sh ${T}_b50 --no-dropin > /dev/null
bash tools/kstats.sh ${T}_b256 --batch 256 --pool 8 --no-dropin > /dev/null
bash tools/kstats.sh ${T}_b2048_nopipeline --batch 2048 --pool 8 --no-pipeline --no-dropin > /dev/null
bash tools/kstats.sh ${T}_b2048 --batch 2048 --pool 8 --no-dropin > /dev/null
bash tools/kstats.sh ${T}_b2048_dataset --batch 2048 --pool 8 --no-dropin --prep dataset > /dev/null
bash tools/kstats.sh ${T}_dd --workload DD --no-dropin > /dev/null
bash tools/kstats.sh ${T}_dd_stress --workload DD --stress-nodes 5748 --no-dropin > /dev/null
bash tools/pmc.sh ${T}_b50 --no-dropin > $OUT/${T}_pmc_b50.txt 2>&1
bash tools/pmc.sh ${T}_b2048 --batch 2048 --no-dropin > $OUT/${T}_pmc_b2048.txt 2>&1
bash tools/pmc_sq.sh ${T}_b50 "--no-dropin" "k_chain_readout_tail|k_wgrad" > $OUT/${T}_sq_b50.txt 2>&1
bash tools/pmc_sq.sh ${T}_b2048 "--batch 2048 --no-dropin" "k_chain_fwd_q|k_chain_bwd_a|k_chain_bwd_b|k_classifier|k_readout_fwd|k_tail_bwd_walk|k_wgrad" > $OUT/${T}_sq_b2048.txt 2>&1
python tools/phase_step_kernel.py COLLAB 50 > $OUT/${T}_phase_step_kernel.txt 2>&1
# the data-parallel code path on ONE GPU (1-rank group): one-shot exchange kernel and RCCL route
BENCH_FORCE_DIST=1 python bench.py --steps 400 --warmup 40 $Q --exchange oneshot 2>> $OUT/${T}_bench.err | grep '^{' > $OUT/${T}_bench_dp1_oneshot.json
BENCH_FORCE_DIST=1 python bench.py --steps 400 --warmup 40 $Q --exchange rccl 2>> $OUT/${T}_bench.err | grep '^{' > $OUT/${T}_bench_dp1_rccl.json
python tools/epoch_time.py COLLAB 1000 > $OUT/${T}_epoch_time.txt 2>&1
python tools/epoch_time.py COLLAB 5000 >> $OUT/${T}_epoch_time.txt 2>&1
python tools/eval_time.py COLLAB 50 > $OUT/${T}_eval_time.txt 2>&1
python tools/eval_time.py MUTAG 50 >> $OUT/${T}_eval_time.txt 2>&1
python tools/eval_time.py PROTEINS 50 >> $OUT/${T}_eval_time.txt 2>&1
python tools/eval_time.py COLLAB 256 >> $OUT/${T}_eval_time.txt 2>&1
EVAL_NO_LOOKAHEAD=1 python tools/eval_time.py COLLAB 50 >> $OUT/${T}_eval_time.txt 2>&1      # (every batch prepared by its own call: three launches)
python tools/phase_eval_kernel.py COLLAB 50 > $OUT/${T}_phase_eval_kernel.txt 2>&1
python tools/phase_step_kernel.py COLLAB 50 median >> $OUT/${T}_phase_step_kernel.txt 2>&1   # (the batch's MEDIAN graph beside its largest: fixed latency)
python tools/route_time.py PROTEINS 50 > $OUT/${T}_route_time.txt 2>&1                       # batches with a graph of 257..512 nodes
python bench.py --batch 128 --steps 200 --warmup 20 --pool 8 $Q > $OUT/${T}_bench_b128.json 2>> $OUT/${T}_bench.err
# config 5's per-rank share on 8 GPUs (32 graphs) through the data-parallel code path on ONE GPU (1-rank group)
BENCH_FORCE_DIST=1 python bench.py --batch 32 --steps 400 --warmup 40 $Q --exchange oneshot 2>> $OUT/${T}_bench.err | grep '^{' > $OUT/${T}_bench_dp1_b32_oneshot.json
BENCH_FORCE_DIST=1 python bench.py --batch 32 --steps 400 --warmup 40 $Q --exchange rccl 2>> $OUT/${T}_bench.err | grep '^{' > $OUT/${T}_bench_dp1_b32_rccl.json
python bench.py --batch 32 --steps 400 --warmup 40 $Q > $OUT/${T}_bench_b32.json 2>> $OUT/${T}_bench.err
bash tools/kstats_cmd.sh ${T}_eval python tools/eval_time.py COLLAB 50 > /dev/null
python tools/phase_readout_tail_dd.py > $OUT/${T}_phase_readout_tail_dd.txt 2>&1
# round 6: 256 graphs per step from a prepared dataset (no riders at all) beside the per-batch line
python bench.py --batch 256 --steps 200 --warmup 20 --pool 8 $Q --prep dataset > $OUT/${T}_bench_b256_dataset.json 2>> $OUT/${T}_bench.err
bash tools/kstats.sh ${T}_b256_dataset --batch 256 --pool 8 --no-dropin --prep dataset > /dev/null
python tools/eval_time.py DD 50 >> $OUT/${T}_eval_time.txt 2>&1
ls -la $OUT | tail -40
python tools/eval_route_time.py PROTEINS 50 > $OUT/${T}_eval_route_time.txt 2>&1      # one-launch evaluation of batches with a 257..512-node graph
# DD at batch 50 WITHOUT the scalar narrow gathers of conv4 (round 6 made them the default): the round-5 form beside the default's lines
DGCNN_NARROW_GATHER=1 python bench.py --workload DD --steps 200 --warmup 20 $Q > $OUT/${T}_bench_dd_narrow1.json 2>> $OUT/${T}_bench.err
DGCNN_NARROW_GATHER=1 bash tools/kstats.sh ${T}_dd_narrow1 --workload DD --no-dropin > /dev/null
