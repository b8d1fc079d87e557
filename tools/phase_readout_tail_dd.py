import sys, torch
sys.path.insert(0, ".")
from dgcnn_amd import _lib, synth
from dgcnn_amd.batch import collate
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
L = _lib.lib()
sh = synth.SHAPES["DD"]
gs = synth.make_graphs("DD", 50, start=0)
gs.sort(key=lambda g: -g.x.shape[0])
print("sizes", [g.x.shape[0] for g in gs[:5]])
b = collate(gs).to("cuda")
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda"); m.train()
tr = Trainer(m, exclusive_device=True)
dbg = torch.zeros(32, dtype=torch.int64, device="cuda")
L.dgcnn_debug_phase_clocks(dbg.data_ptr())
rn = {8: "topk", 9: "gather+Wstage", 10: "conv5", 11: "pool+conv6", 12: "fc1", 13: "fc2+lsm"}
tn = ["(sync)", "stage+dlogit", "fc2 bwd+partial", "fc1^T", "conv6 bwd", "pool/relu", "W5/W6 partials", "scatter"]
for it in range(3):
    tr.train_step(b, b.y); torch.cuda.synchronize()
    v = dbg.cpu().tolist()
    fw = " ".join(f"{rn[k]}={v[k] - (v[14] if k == 8 else v[k-1])}" for k in range(8, 14))
    bw = " ".join(f"{tn[k]}={v[k] - (v[13] if k == 0 else v[k-1])}" for k in range(0, 8))
    print(f"it{it} total={v[7]-v[14]} :: FWD {fw} :: BWD {bw}")
L.dgcnn_debug_phase_clocks(None)
