"""GPU-box helper: clock64() phase deltas of workgroup 0 / thread 0 of k_fused_fwd_d (+ the readout body's marks)."""
import sys, torch
sys.path.insert(0, ".")
from dgcnn_amd import _lib, synth
from dgcnn_amd.model import Model
L = _lib.lib()
sh = synth.SHAPES["COLLAB"]
b = synth.make_batch("COLLAB", 50, start=0).to("cuda")
print("n of graph 0:", int((b.batch == 0).sum()), "max_nodes", b.max_nodes)
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda").eval()
dbg = torch.zeros(16 + 2 * 64, dtype=torch.int64, device="cuda")
L.dgcnn_debug_phase_clocks(dbg.data_ptr())
names = ["start", "zero+params", "bitmap+xs", "conv1", "conv2", "conv3", "conv4", "-", "topk", "gather", "conv5", "pool+conv6", "fc1", "fc2+lsm"]
with torch.no_grad():
    for it in range(4):
        m(b); torch.cuda.synchronize()
        v = dbg.cpu().tolist()
        print(f"   prologue: graph_ptr={v[14]-v[0]} zeroing={v[15]-v[14]} loads+barrier={v[1]-v[15]}")
        print(f"it{it} total={v[13]-v[0]} :: " + " ".join(f"{names[k]}={v[k]-v[k-1]}" for k in range(1, 14) if k != 7))
L.dgcnn_debug_phase_clocks(None)
pb = dbg.cpu()[16:16 + 100].view(-1, 2)
if int(pb[:, 0].max()) > 0:
    order = pb[:, 1].argsort()
    print("per-block (n, cycles):", [(int(pb[i, 1]), int(pb[i, 0])) for i in order.tolist()])
