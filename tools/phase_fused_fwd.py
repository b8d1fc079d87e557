"""GPU-box helper: print clock64() phase deltas of workgroup 0 of the fused forward kernel."""
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
names = ["start", "stage+lin1", "layer1", "layer2", "layer3", "conv4", "-", "-", "topk", "gather+Wstage", "conv5",
         "pool+conv6", "fc1", "fc2+lsm"]
with torch.no_grad():
    for it in range(5):
        m(b)
        torch.cuda.synchronize()
        v = dbg.cpu().tolist()
        idx = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13]
        line = []
        for a, c in zip(idx[:-1], idx[1:]):
            line.append(f"{names[c]}={(v[c]-v[a])}")
        n0 = int((b.batch == 0).sum())
        print(f"it{it} n(graph0)={n0} total={v[13]-v[0]} cycles :: " + " ".join(line))
L.dgcnn_debug_phase_clocks(None)
