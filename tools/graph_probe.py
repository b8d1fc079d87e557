"""Probe: what does a hipGraph replay of the (unchanged) 13-kernel step cost?  Captures one graph per batch of a
small pool (shapes/scalars baked in -- a timing probe, not a product path) and replays them round-robin.
    python tools/graph_probe.py [B] [pool]"""
import sys, time
import torch
sys.path.insert(0, ".")
from dgcnn_amd import synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
batches = [synth.make_batch("COLLAB", B, start=i * B).to(dev) for i in range(P)]
m = Model(batches[0].x.shape[1], 3).to(dev); m.train()
tr = Trainer(m, exclusive_device=True)
for _ in range(3):
    for b in batches:
        tr.train_step(b, b.y)
torch.cuda.synchronize()

def timed(fn, iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(iters):
        fn(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / iters * 1e6, (t1 - t0) / iters * 1e6

e_tot, e_host = timed(lambda i: tr.train_step(batches[i % P], batches[i % P].y), 2000)
print(f"eager : {e_tot:7.1f} us/step (host enqueue {e_host:6.1f})")
graphs = []
s = torch.cuda.Stream()
try:
    with torch.cuda.stream(s):
        for b in batches:
            g = torch.cuda.CUDAGraph()
            g.capture_begin()
            tr.train_step(b, b.y)
            g.capture_end()
            graphs.append(g)
    torch.cuda.synchronize()
    for g in graphs: g.replay()
    g_tot, g_host = timed(lambda i: graphs[i % P].replay(), 2000)
    print(f"graph : {g_tot:7.1f} us/step (host enqueue {g_host:6.1f})")
except Exception as ex:
    print("capture failed:", repr(ex))
