"""GPU-box helper: evaluation-step time (the reference's test() loop body, train.py:59-64, through Trainer.eval_step with look-ahead)
of batches whose largest graph has 257..512 nodes, per route: the library's default (launch per layer), the chain forward forced
(two launches), and the one-launch evaluation kernel's two-tiles-per-wave form (round 6: Model.inference_one_launch).
usage: python tools/eval_route_time.py [workload] [graphs per batch]"""
import sys, time, torch
sys.path.insert(0, ".")
from dgcnn_amd import synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
name, G = (sys.argv[1] if len(sys.argv) > 1 else "PROTEINS"), int(sys.argv[2]) if len(sys.argv) > 2 else 50
sh = synth.SHAPES[name]
big, small = [], []
k = 0
while (len(big) < 8 or len(small) < 8) and k < 400:
    b = synth.make_batch(name, G, start=G * k); k += 1
    if 256 < b.max_nodes <= 512 and len(big) < 8: big.append(b.to("cuda"))
    elif b.max_nodes <= 256 and len(small) < 8: small.append(b.to("cuda"))
print(f"{name} x {G}: {len(big)} batches with 256 < max_nodes <= 512 (max {max(b.max_nodes for b in big)}), {len(small)} with <= 256")
for label, bs, chain, wide in (("<=256 default (one launch)", small, None, False), ("257..512 default", big, None, False),
                               ("257..512 use_chain", big, True, False), ("257..512 one launch", big, None, True),
                               ("mixed, one launch", [x for p in zip(small, big) for x in p], None, True),
                               ("mixed, default", [x for p in zip(small, big) for x in p], None, False)):
    torch.manual_seed(324)
    m = Model(sh.num_features, sh.num_classes).to("cuda"); m.eval()
    if chain is not None: m.use_chain = chain
    m.inference_one_launch = wide
    tr = Trainer(m, exclusive_device=True)
    nb = len(bs)
    for i in range(3 * nb): tr.eval_step(bs[i % nb], bs[i % nb].y, next_data=bs[(i + 1) % nb])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40 * nb
    for i in range(n): tr.eval_step(bs[i % nb], bs[i % nb].y, next_data=bs[(i + 1) % nb])
    torch.cuda.synchronize()
    print(f"  {label:28s} {1e6 * (time.perf_counter() - t0) / n:7.1f} us/step")
    tr.read_metrics()
