"""GPU-box helper: where the drop-in route's host time goes inside this build's own share (Function.forward / backward)."""
import sys, time, torch
sys.path.insert(0, ".")
from torch import nn
from dgcnn_amd import synth, _lib
from dgcnn_amd import model as M
from dgcnn_amd.optim import Adam as FlatAdam
sh = synth.SHAPES["COLLAB"]
batches = [b.to("cuda") for b in synth.make_batches("COLLAB", 500, 50)]
m = M.Model(sh.num_features, sh.num_classes).to("cuda"); m.train()
opt = FlatAdam(m.parameters()); crit = nn.NLLLoss()
T = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
    return w
F = M._DGCNNFunction
F.forward = staticmethod(timed("F.forward", F.forward))
F.backward = staticmethod(timed("F.backward", F.backward))
L = _lib.lib()
class LW:
    def __init__(s, L): s.L = L
    def __getattr__(s, n): return timed("C." + n, getattr(s.L, n))
lw = LW(L)
_lib.lib = lambda: lw
M.Model._grad_views = timed("grad_views", M.Model._grad_views)
def it(i, rec):
    data = batches[i % 10]
    t0 = time.perf_counter(); pred = m(data)
    t1 = time.perf_counter(); loss = crit(pred, data.y)
    t2 = time.perf_counter(); loss.backward()
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter(); opt.zero_grad()
    t5 = time.perf_counter(); a = loss.item(); t6 = time.perf_counter(); c = (pred.argmax(dim=1) == data.y).sum().item()
    t7 = time.perf_counter()
    if rec:
        for k, v in zip(("fwd", "loss", "bwd", "step", "zero", "item1", "acc+item2"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6)): T[k] = T.get(k, 0.0) + v
for i in range(50): it(i, False)
torch.cuda.synchronize(); T.clear()
K = 300
t0 = time.perf_counter()
for i in range(K): it(i, True)
torch.cuda.synchronize()
print("total us/step", round(1e6 * (time.perf_counter() - t0) / K, 1))
print({k: round(1e6 * v / K, 1) for k, v in T.items()})
