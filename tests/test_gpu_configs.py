"""BASELINE.json configs 3 and 5 on the HIP path (the two configs that had not run):
config 5 -- COLLAB-shape, global batch 256, data parallel (per-rank 32 at 8 GPUs): forward+backward parity at 256 and
            32 graphs, the bitwise batch-composition property, a 2-rank run at global batch 256 == 1 rank;
config 3 -- the bf16 leg (hs stored bf16, X.W on the bf16 matrix cores), with its STATED tolerances."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dgcnn_amd import synth
from dgcnn_amd.batch import collate
from oracle import ref_dense
from parity_util import (KEY_TOL, check_backward_parity, check_forward_parity, cpu_state_dict, gpu_xcat, make_model)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("bs", [256, 32])
@pytest.mark.parametrize("agg", ["auto", "dense"])
def test_config5_collab_forward_backward_vs_oracle(bs, agg):
    sh = synth.SHAPES["COLLAB"]
    b = synth.make_batch("COLLAB", bs, start=2000)
    m = make_model(sh.num_features, sh.num_classes)
    if agg != "auto":
        m.agg_mode = agg
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


@pytest.mark.parametrize("agg", ["sparse", "dense"])
def test_config5_global_batch_equals_the_eight_rank_shards_bitwise(agg):
    """a graph's result does not depend on the batch it travels in: the 256-graph global batch and its eight 32-graph
    rank shards give bit-identical log-probabilities (what makes data-parallel evaluation exact)"""
    sh = synth.SHAPES["COLLAB"]
    graphs = synth.make_graphs("COLLAB", 256, start=2000)
    m = make_model(sh.num_features, sh.num_classes)
    m.agg_mode = agg
    m.eval()
    with torch.no_grad():
        full = m(collate(graphs).to("cuda")).clone()
        parts = [m(collate(graphs[k:k + 32]).to("cuda")).clone() for k in range(0, 256, 32)]
    m.check_errors()
    assert torch.equal(torch.cat(parts), full)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from dgcnn_amd import dist as ddist, synth as sy
    from dgcnn_amd.train import Trainer
    from parity_util import make_model as mk
    ddist.init_from_env("gloo")
    torch.cuda.set_device(0)
    sh = sy.SHAPES["COLLAB"]
    m = mk(sh.num_features, sh.num_classes, seed=324)
    m.eval()
    tr = Trainer(m, process_group=dist.group.WORLD)
    fulls = [sy.make_batch("COLLAB", 256, start=3000 + 256 * k) for k in range(2)]
    shards = [ddist.shard_batch(f, rank, world).to("cuda") for f in fulls]
    # evaluation first (identical replicas: every graph's forward is bit-identical to the single-rank run), then ONE
    # training step -- COLLAB-shaped sort keys sit within 1e-6 of each other, so after a step whose gradient sum was
    # taken in a different order a SortPooling near-tie may legitimately flip and trajectories part by O(1e-3)
    ev = tr.test_epoch(shards, num_samples=2 * 256)
    # train_step without global_batch: the Trainer derives it per batch and reduces the metrics itself.
    # (model.eval(): the dropout stream is indexed by the graph's position in the LOCAL batch, so only dropout-free
    # steps are comparable between a sharded and an unsharded run)
    tr.reset_metrics()
    tr.train_step(shards[0], shards[0].y)
    loss, acc = tr.read_metrics()
    torch.cuda.synchronize()
    ret[f"m{rank}"] = (loss, acc) + tuple(ev)
    if rank == 0:
        ret["params"] = m.flat_params.detach().cpu()
    dist.barrier()
    dist.destroy_process_group()


def test_config5_two_rank_data_parallel_global_batch_256_equals_one_rank():
    from dgcnn_amd.train import Trainer
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    sh = synth.SHAPES["COLLAB"]
    m = make_model(sh.num_features, sh.num_classes, seed=324)
    m.eval()
    tr = Trainer(m)
    fulls = [synth.make_batch("COLLAB", 256, start=3000 + 256 * k).to("cuda") for k in range(2)]
    ev = tr.test_epoch(fulls, num_samples=2 * 256)
    tr.reset_metrics()
    tr.train_step(fulls[0], fulls[0].y)
    loss, acc = tr.read_metrics()
    torch.cuda.synchronize()
    assert ret["m0"] == ret["m1"]                           # the reduced (dataset-level) numbers, on every rank
    assert abs(ret["m0"][2] - ev[0]) < 1e-5 and abs(ret["m0"][3] - ev[1]) < 1e-9      # eval: identical forward on every graph
    assert abs(ret["m0"][0] - loss) < 1e-5 and abs(ret["m0"][1] - acc) < 1e-9
    torch.testing.assert_close(ret["params"], m.flat_params.detach().cpu(), rtol=1e-5, atol=2e-6)


# ---- config 3: the bf16 leg --------------------------------------------------------------------------------------
BF16_XCAT_TOL = 2e-2      # per-node activations vs the fp64 oracle: hs rounded to bf16 (2^-9 relative) before each of the
                          # three 32-wide aggregations and the 32 -> 1 one, X.W inputs rounded to bf16; |x| <= 1
BF16_LOGIT_TOL = 5e-2     # log-probabilities vs the oracle evaluated on the kernel's own permutation


@pytest.mark.parametrize("name,bs", [("COLLAB", 50), ("PROTEINS", 24), ("MUTAG", 50)])
def test_config3_bf16_leg_forward_and_gradients_within_stated_tolerance(name, bs):
    sh = synth.SHAPES[name]
    start = 1000
    b = synth.make_batch(name, bs, start=start)
    while b.max_nodes > 512:
        start += bs
        b = synth.make_batch(name, bs, start=start)
    m = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m)
    m.compute_dtype = "bf16"
    m.eval()
    with torch.no_grad():
        logp = m(b.to("cuda")).cpu()
    m.check_errors()
    _, aux = ref_dense.forward_dense(sd, b.x, b.edge_index, b.batch, b.num_graphs, return_all=True)
    xc = gpu_xcat(m)
    err_x = float((xc.double() - aux["xcat"].detach()).abs().max())
    assert 1e-6 < err_x <= BF16_XCAT_TOL, err_x            # (> fp32 noise: the leg really ran in bf16)
    perm = m.last_workspace_view("perm").cpu()
    ok, msg = ref_dense.check_perm_valid(aux["xcat"], aux["ptr"], perm, tol=BF16_XCAT_TOL)
    assert ok, msg
    ref = ref_dense.forward_dense(sd, b.x, b.edge_index, b.batch, b.num_graphs, perm_override=perm).detach()
    assert float((logp.double() - ref).abs().max()) <= BF16_LOGIT_TOL
    # gradients: the backward is fp32 on the activations the bf16 forward saved (straight-through); direction and
    # size must agree with the fp64 gradients evaluated on the same permutation and dropout mask
    m.train()
    bg = b.to("cuda")
    out = m(bg)
    torch.nn.functional.nll_loss(out, bg.y).backward()
    mask = m.last_workspace_view("drop_mask").cpu()
    perm = m.last_workspace_view("perm").cpu()
    _, _, g_ref, _ = ref_dense.loss_and_grads_dense(sd, b.x, b.edge_index, b.batch, b.y, b.num_graphs, dropout_mask=mask,
                                                   perm_override=perm)
    gmax = max(float(v.abs().max()) for v in g_ref.values())
    for k, p in m.named_parameters():
        g, r = p.grad.detach().cpu().double().reshape(-1), g_ref[k].double().reshape(-1)
        # stated tolerance: 10 % of the tensor's gradient norm, plus a floor of 2 % of the largest gradient entry of the
        # model per element (a bias gradient that is a sum of cancelling terms can be ~0 in fp64)
        assert float((g - r).norm()) <= 0.10 * float(r.norm()) + 0.02 * gmax * (r.numel() ** 0.5), k


def test_config3_bf16_leg_is_reproducible_and_refuses_the_gather_form():
    from dgcnn_amd import _lib
    sh = synth.SHAPES["COLLAB"]
    b = synth.make_batch("COLLAB", 16, start=5).to("cuda")
    m = make_model(sh.num_features, sh.num_classes)
    m.compute_dtype = "bf16"
    m.eval()
    with torch.no_grad():
        a = m(b).clone(); c = m(b).clone()
        assert torch.equal(a, c)
        m.agg_mode = "sparse"                     # no bf16 form of the CSR-gather kernels: loud error, never a silent fp32 run
        with pytest.raises(_lib.DgcnnError):
            m(b)
