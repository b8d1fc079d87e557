"""GPU-box helper: steady-state us per step of the drop-in route (reference loop body verbatim, torch / flat Adam), three interleaved
repeats, with and without autograd multithreading (the first pass is 25-50 % slower than the steady state)."""
import sys, time, torch
sys.path.insert(0, ".")
from torch import nn
from dgcnn_amd import synth
from dgcnn_amd.model import Model
from dgcnn_amd.optim import Adam as NewAdam
sh = synth.SHAPES["COLLAB"]
batches = [b.to("cuda") for b in synth.make_batches("COLLAB", 500, 50)]
def run(mk, K=300):
    torch.manual_seed(324)
    m = Model(sh.num_features, sh.num_classes).to("cuda"); m.train()
    opt, crit = mk(m), nn.NLLLoss()
    def loop(n):
        running, correct = 0.0, 0
        for i in range(n):
            data = batches[i % 10]
            pred = m(data); loss = crit(pred, data.y); loss.backward(); opt.step(); opt.zero_grad()
            running += loss.item(); correct += (pred.argmax(dim=1) == data.y).sum().item()
        return running
    loop(30); torch.cuda.synchronize(); t0 = time.perf_counter(); r = loop(K); torch.cuda.synchronize()
    return round(1e6 * (time.perf_counter() - t0) / K, 1)
for rep in range(3):
    out = {}
    for mt in (True, False):
        with torch.autograd.set_multithreading_enabled(mt):
            out[f"torch mt={mt}"] = run(lambda m: torch.optim.Adam(m.parameters()))
            out[f"flat mt={mt}"] = run(lambda m: NewAdam(m.parameters()))
    print(out)
