"""GPU-box helper: time of the reference's test() loop body (train.py:59-64) through Trainer.eval_step, per batch of 50,
batches resident on the device (raw edge lists: graph preparation of every batch inside the timed region -- it rides on the
previous batch's launch, as in Trainer.test_epoch).
usage: python tools/eval_time.py [workload] [graphs per batch]"""
import os, sys, time, torch
sys.path.insert(0, ".")
from dgcnn_amd import synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
name, G = (sys.argv[1] if len(sys.argv) > 1 else "COLLAB"), int(sys.argv[2]) if len(sys.argv) > 2 else 50
sh = synth.SHAPES[name]
bs = [synth.make_batch(name, G, start=G * k).to("cuda") for k in range(40)]
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda"); m.eval()
tr = Trainer(m, exclusive_device=True)
nb = len(bs)
LA = os.environ.get("EVAL_NO_LOOKAHEAD") is None      # EVAL_NO_LOOKAHEAD=1: every batch prepared by its own call
for k, b in enumerate(bs): tr.eval_step(b, b.y, next_data=bs[(k + 1) % nb] if LA else None)      # (look-ahead as Trainer.test_epoch's)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(25):
        for k, b in enumerate(bs): tr.eval_step(b, b.y, next_data=bs[(k + 1) % nb] if LA else None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (25 * len(bs))
    print(f"{name} x {G}: eval step {1e6 * dt:.1f} us/batch ({G / dt:,.0f} graphs/s)")
