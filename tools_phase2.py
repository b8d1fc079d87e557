"""GPU-box helper: clock64() phase deltas of workgroup 0 / thread 0 of k_gcn_fwd32 (last of the 3 launches)."""
import sys, torch
sys.path.insert(0, ".")
from dgcnn_amd import _lib, synth
from dgcnn_amd.model import Model
L = _lib.lib()
sh = synth.SHAPES["COLLAB"]
b = synth.make_batch("COLLAB", 50, start=0).to("cuda")
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda").eval()
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
L.dgcnn_debug_phase_clocks(dbg.data_ptr())
names = ["start", "rowptr", "gather", "tanh+store", "barrier", "mfma+store", "end"]
with torch.no_grad():
    for it in range(5):
        m(b); torch.cuda.synchronize()
        v = dbg.cpu().tolist()
        print(f"it{it} total={v[6]-v[0]} :: " + " ".join(f"{names[k]}={v[k]-v[k-1]}" for k in range(1, 7)))
L.dgcnn_debug_phase_clocks(None)
