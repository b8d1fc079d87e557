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


def _gpu_worker(rank, world, port, ret):
    """two ranks on the SAME GPU (the box has one): gloo carries the collective, the kernels are the real HIP ones"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from dgcnn_amd import dist as ddist, synth
    from dgcnn_amd.train import Trainer
    from parity_util import make_model
    ddist.init_from_env("gloo")
    torch.cuda.set_device(0)
    sh = synth.SHAPES["PROTEINS"]
    m = make_model(sh.num_features, sh.num_classes, seed=324 + rank)     # replicas start different ...
    ddist.broadcast_parameters(m.flat_params, src=0)                     # ... one broadcast makes them identical
    m.eval()                                                             # no dropout: shards and full batch comparable
    tr = Trainer(m, process_group=dist.group.WORLD)
    fulls = [synth.make_batch("PROTEINS", 14, start=500 + 14 * k) for k in range(3)]
    shards = [ddist.shard_batch(f, rank, world).to("cuda") for f in fulls]
    for k in range(6):          # pipelined data-parallel route: fwd+bwd (+ look-ahead), ONE all-reduce, Adam
        cur, nxt = shards[k % 3], shards[(k + 1) % 3]
        tr.train_step(cur, cur.y, global_batch=14, next_data=nxt)
    torch.cuda.synchronize()
    m.check_errors()
    loss, correct = tr.read_metrics()          # reduced over the ranks inside: the dataset-level numbers, on every rank
    ret[f"metrics{rank}"] = (loss, correct)
    if rank == 0:
        ret["params"] = m.flat_params.detach().cpu()
        ret["loss"], ret["correct"] = loss, correct
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_data_parallel_trainer_on_gpu_equals_single_rank():
    """The whole data-parallel training route on real kernels: 2 ranks x cost-balanced half batches (loss scaled by
    1/B_global in-kernel, one flat all-reduce, replicated Adam) must follow the same trajectory as 1 rank x full
    batches -- parameters after 6 steps, summed loss and #correct."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dgcnn_amd import synth
    from dgcnn_amd.train import Trainer
    from parity_util import make_model
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_worker, args=(2, port, ret), nprocs=2, join=True)
    sh = synth.SHAPES["PROTEINS"]
    m = make_model(sh.num_features, sh.num_classes, seed=324)
    m.eval()
    tr = Trainer(m)
    fulls = [synth.make_batch("PROTEINS", 14, start=500 + 14 * k).to("cuda") for k in range(3)]
    for k in range(6):
        tr.train_step(fulls[k % 3], fulls[k % 3].y, next_data=fulls[(k + 1) % 3])
    torch.cuda.synchronize()
    loss, correct = tr.read_metrics()
    assert ret["metrics0"] == ret["metrics1"]            # Trainer.read_metrics all-reduces under a process group
    assert abs(ret["loss"] - loss) < 1e-4 * max(1.0, abs(loss)) and ret["correct"] == correct
    # same gradients up to summation order; six Adam steps of size 1e-3 amplify that to at most a few 1e-6
    torch.testing.assert_close(ret["params"], m.flat_params.detach().cpu(), rtol=1e-4, atol=1e-5)


def _oneshot_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from dgcnn_amd import dist as ddist, synth
    from dgcnn_amd.train import Trainer
    from parity_util import make_model
    ddist.init_from_env("gloo")
    torch.cuda.set_device(0)
    sh = synth.SHAPES["PROTEINS"]
    fulls = [synth.make_batch("PROTEINS", 14, start=500 + 14 * k) for k in range(3)]
    shards = [ddist.shard_batch(f, rank, world).to("cuda") for f in fulls]
    out = {}
    for route in ("collective", "one_shot"):
        m = make_model(sh.num_features, sh.num_classes, seed=324)
        m.eval()
        tr = Trainer(m, process_group=dist.group.WORLD, one_shot=(route == "one_shot"))
        for k in range(7):
            tr.train_step(shards[k % 3], shards[k % 3].y, global_batch=14, next_data=shards[(k + 1) % 3])
        torch.cuda.synchronize()
        tr.read_metrics()
        out[route] = m.flat_params.detach().cpu().clone()
        if route == "one_shot":
            out["gsum"] = tr._peer.grad_tensor(7).detach().cpu().clone()      # own gradient of the last step (buffer 7 & 1)
        tr.close()
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_one_shot_peer_allreduce_adam_equals_the_collective_route_bit_for_bit():
    """dgcnn_allreduce_adam_step (gradients in hipIpc-mapped fine-grained memory, rank-ordered sum + Adam in one launch per
    rank) against all_reduce + dgcnn_adam_step: two processes sharing the one GPU, 7 pipelined data-parallel steps --
    identical parameters on both ranks and between the two routes, bit for bit (2 ranks: a + b is order-free)."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_oneshot_worker, args=(2, port, ret), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    assert torch.equal(a["collective"], b["collective"]) and torch.equal(a["one_shot"], b["one_shot"])
    assert torch.equal(a["one_shot"], a["collective"])
    assert float(a["gsum"].abs().max()) > 0 and not torch.equal(a["gsum"], b["gsum"])       # ranks hold different shards
