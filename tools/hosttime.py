"""GPU-box helper: host enqueue time vs device time per training step (is the loop host-bound?)."""
import sys, time, torch
sys.path.insert(0, ".")
from dgcnn_amd import synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
sh = synth.SHAPES["COLLAB"]
batches = [b.to("cuda") for b in synth.make_batches("COLLAB", 500, 50)]
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda"); m.train()
tr = Trainer(m, exclusive_device=True)
for pf in (False, True):
    for i in range(50):
        tr.train_step(batches[i % 10], batches[i % 10].y, next_data=batches[(i + 1) % 10] if pf else None)
    torch.cuda.synchronize()
    K = 400
    t0 = time.perf_counter()
    for i in range(K):
        b = batches[i % 10]
        tr.train_step(b, b.y, next_data=batches[(i + 1) % 10] if pf else None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"pipelined={pf}: host enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")

# pure C-side enqueue cost of one pipelined step (same argument block re-submitted; results are not meaningful)
import ctypes
from dgcnn_amd import _lib
L = _lib.lib()
b = batches[0]
ent = tr._step_args(b, b.y)
a = ent[2]
a.flags &= ~_lib.FLAG_PREPARED
stream = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    a.epoch = (i % 1000) + 1
    L.dgcnn_pipeline_train_step(tr._pipe, ctypes.byref(a), None, stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"C call only (unpipelined, 12 launches): host {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")
