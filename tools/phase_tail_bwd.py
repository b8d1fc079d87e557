"""GPU-box helper: clock64() phase deltas of workgroup 0 / thread 0 of k_tail_bwd."""
import sys, torch
sys.path.insert(0, ".")
from dgcnn_amd import _lib, synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
L = _lib.lib()
sh = synth.SHAPES["COLLAB"]
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
b = synth.make_batch("COLLAB", BS, start=0).to("cuda")
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda"); m.train()
tr = Trainer(m, exclusive_device=True)
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
L.dgcnn_debug_phase_clocks(dbg.data_ptr())
names = ["start", "stage+dlogit", "fc2 bwd+partial", "fc1^T (gflat)", "conv6 bwd", "pool/relu", "W5/W6 partials", "scatter"]
for it in range(5):
    tr.train_step(b, b.y); torch.cuda.synchronize()
    v = dbg.cpu().tolist()
    print(f"it{it} total={v[7]-v[0]} :: " + " ".join(f"{names[k]}={v[k]-v[k-1]}" for k in range(1, 8)))
L.dgcnn_debug_phase_clocks(None)
