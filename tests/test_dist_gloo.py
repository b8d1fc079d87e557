"""Data-parallel path on CPU: world_size-2 gloo process groups.  The compute engine
here is the oracle (tests may use it); what is under test is the build's sharding + flat-bucket
all-reduce + global-mean loss scaling: 2 ranks x half the batch == 1 rank x the full batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from dgcnn_amd import dist as ddist, synth
    from oracle import ref_ops
    r, w, _ = ddist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    sh = synth.SHAPES["PROTEINS"]
    full = synth.make_batch("PROTEINS", 12, start=60)
    mine = ddist.shard_batch(full, rank, world)
    torch.manual_seed(324 + rank)                  # replicas start DIFFERENT on purpose ...
    model = ref_ops.RefModel(sh.num_features, sh.num_classes)
    model.eval(); model.stable_sort = True         # eval: no dropout mask to synchronise
    flat0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    ddist.broadcast_parameters(flat0, src=0)       # ... and are made identical by one broadcast
    off = 0
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(flat0[off:off + p.numel()].view(p.shape)); off += p.numel()
    red = ddist.GradAllReduce()
    gb = red.global_batch(mine.num_graphs)
    assert gb == 12
    logp = model(mine)
    # local SUM of label log-probs scaled by 1/B_global == this rank's share of the global mean NLL
    loss = -logp[torch.arange(mine.num_graphs), mine.y].sum() / gb
    loss.backward()
    params = list(model.parameters())
    flat, metas = ddist.flatten_grads(params)
    red(flat)                                       # ONE collective for the whole model
    lsum = loss.detach().clone()
    dist.all_reduce(lsum)
    if rank == 0:
        ret["flat"] = flat.clone()
        ret["loss"] = float(lsum)
        ret["sizes"] = [mine.num_graphs]
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_rank():
    sys.path.insert(0, ROOT)
    from dgcnn_amd import synth
    from oracle import ref_ops
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    # single-rank reference on the full batch with rank 0's initial parameters
    sh = synth.SHAPES["PROTEINS"]
    full = synth.make_batch("PROTEINS", 12, start=60)
    torch.manual_seed(324)
    model = ref_ops.RefModel(sh.num_features, sh.num_classes)
    model.eval(); model.stable_sort = True
    loss = ref_ops.nll_mean(model(full), full.y)
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert abs(ret["loss"] - float(loss.detach())) < 1e-6
    assert torch.allclose(ret["flat"], ref, rtol=1e-4, atol=1e-7)
    assert 1 <= ret["sizes"][0] <= 11


def test_shard_and_range_helpers():
    sys.path.insert(0, ROOT)
    from dgcnn_amd import dist as ddist, synth
    b = synth.make_batch("COLLAB", 9, start=4)
    parts = [ddist.shard_batch(b, r, 4) for r in range(4)]
    assert sum(p.num_graphs for p in parts) == 9
    assert torch.equal(torch.cat([p.y for p in parts]), b.y)
    assert all(p.coalesced_undirected for p in parts) and all(p.max_nodes > 0 and p.max_edges > 0 for p in parts)
    # cost balance: no shard carries more than ~2x the mean cost on this skewed workload
    cost = [p.num_nodes + p.num_edges for p in parts]
    assert max(cost) <= 2.2 * (sum(cost) / 4)
    cover = []
    for r in range(3):
        g0, g1 = ddist.graph_range(10, r, 3)
        cover += list(range(g0, g1))
    assert cover == list(range(10))
    assert ddist.shard_batch(b, 0, 1) is b
