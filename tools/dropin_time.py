"""GPU-box helper: cost of the DROP-IN route -- the reference's own loop body (train.py:36-45) around this build's Model,
torch NLLLoss, autograd backward, torch.optim.Adam, the two .item() syncs -- next to the fused Trainer route."""
import sys, time, torch
sys.path.insert(0, ".")
from torch import nn
from torch.optim import Adam
from dgcnn_amd import synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer

sh = synth.SHAPES["COLLAB"]
batches = [b.to("cuda") for b in synth.make_batches("COLLAB", 500, 50)]
torch.manual_seed(324)

def loop(model, opt, crit, K, sync):
    running, correct = 0.0, 0
    for i in range(K):
        data = batches[i % len(batches)]
        pred = model(data)
        loss = crit(pred, data.y)
        loss.backward()
        opt.step(); opt.zero_grad()
        if sync:
            running += loss.item()
            correct += (pred.argmax(dim=1) == data.y).sum().item()
    return running, correct

for name, mk in (("Adam(model.parameters())", lambda m: Adam(m.parameters())),
                 ("Adam(..., fused=True)", lambda m: Adam(m.parameters(), fused=True)),
                 ("dgcnn_amd.optim.Adam(model.parameters())", "flat")):
    m = Model(sh.num_features, sh.num_classes).to("cuda"); m.train()
    if mk == "flat":
        from dgcnn_amd.optim import Adam as FlatAdam
        opt = FlatAdam(m.parameters())
    else:
        opt = mk(m)
    crit = nn.NLLLoss()
    for sync in (True, False):
        loop(m, opt, crit, 50, sync); torch.cuda.synchronize()
        t0 = time.perf_counter(); loop(m, opt, crit, 300, sync); torch.cuda.synchronize()
        print(f"drop-in, {name:40s} item() syncs={sync}: {1e6 * (time.perf_counter() - t0) / 300:7.1f} us/step")
m = Model(sh.num_features, sh.num_classes).to("cuda"); m.train(); tr = Trainer(m, exclusive_device=True)
for i in range(50): tr.train_step(batches[i % 10], batches[i % 10].y, next_data=batches[(i + 1) % 10])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(300): tr.train_step(batches[i % 10], batches[i % 10].y, next_data=batches[(i + 1) % 10])
torch.cuda.synchronize()
print(f"fused Trainer route: {1e6 * (time.perf_counter() - t0) / 300:7.1f} us/step")
