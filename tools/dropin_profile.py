"""GPU-box helper: host-time breakdown of the drop-in route (reference loop body around this build's Model)."""
import sys, time, torch, cProfile, pstats, io
sys.path.insert(0, ".")
from torch import nn
from torch.optim import Adam
from dgcnn_amd import synth
from dgcnn_amd.model import Model
sh = synth.SHAPES["COLLAB"]
batches = [b.to("cuda") for b in synth.make_batches("COLLAB", 500, 50)]
m = Model(sh.num_features, sh.num_classes).to("cuda"); m.train()
opt = Adam(m.parameters()); crit = nn.NLLLoss()
T = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0, "step": 0.0, "zero": 0.0}
def it(i, rec):
    data = batches[i % 10]
    t0 = time.perf_counter(); pred = m(data)
    t1 = time.perf_counter(); loss = crit(pred, data.y)
    t2 = time.perf_counter(); loss.backward()
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter(); opt.zero_grad()
    t5 = time.perf_counter()
    if rec:
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[k] += v
for i in range(50): it(i, False)
torch.cuda.synchronize()
K = 300
for i in range(K): it(i, True)
torch.cuda.synchronize()
print({k: round(1e6 * v / K, 1) for k, v in T.items()}, "us host per step")
pr = cProfile.Profile(); pr.enable()
for i in range(100): it(i, False)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3500])
