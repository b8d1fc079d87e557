"""GPU-box helper: clock64() phase deltas of workgroup 0 / thread 0 of k_readout_fwd (tiled path)."""
import sys, torch
sys.path.insert(0, ".")
from dgcnn_amd import _lib, synth
from dgcnn_amd.model import Model
L = _lib.lib()
sh = synth.SHAPES["COLLAB"]
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
b = synth.make_batch("COLLAB", BS, start=0).to("cuda")
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda").eval()
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
L.dgcnn_debug_phase_clocks(dbg.data_ptr())
names = {8: "graph_ptr+topk", 9: "gather+Wstage", 10: "conv5", 11: "pool+conv6", 12: "fc1", 13: "fc2+lsm"}
with torch.no_grad():
    for it in range(5):
        m(b); torch.cuda.synchronize()
        v = dbg.cpu().tolist()
        print(f"it{it} total={v[13]-v[7]} :: " + " ".join(f"{names[k]}={v[k]-v[k-1]}" for k in range(8, 14)))
L.dgcnn_debug_phase_clocks(None)
