"""GPU-box helper: clock64() phase stamps of workgroup 0 of the one-launch evaluation kernel (k_chain_readout_eval) with the
batch's LARGEST graph placed first.   usage: python tools/phase_eval_kernel.py [workload] [graphs]"""
import sys, torch
sys.path.insert(0, ".")
from dgcnn_amd import _lib, synth
from dgcnn_amd.batch import collate
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
L = _lib.lib()
name, G = (sys.argv[1] if len(sys.argv) > 1 else "COLLAB"), int(sys.argv[2]) if len(sys.argv) > 2 else 50
sh = synth.SHAPES[name]
graphs = synth.make_graphs(name, G, start=0)
order = sorted(range(G), key=lambda i: -graphs[i].num_nodes)
b = collate([graphs[i] for i in order]).to("cuda")
b.coalesced_undirected = True
print(f"{name} x {G}: workgroup 0 = graph of {graphs[order[0]].num_nodes} nodes")
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda"); m.eval()
tr = Trainer(m, exclusive_device=True)
dbg = torch.zeros(80, dtype=torch.int64, device="cuda")
L.dgcnn_debug_phase_clocks(dbg.data_ptr())
rn = {8: "topk", 9: "gather+Wstage", 10: "conv5", 11: "pool+conv6", 12: "fc1", 13: "fc2+lsm"}
for it in range(4):
    tr.eval_step(b, b.y); torch.cuda.synchronize()
    v = dbg.cpu().tolist()
    fw = " ".join(f"{rn[k]}={v[k] - (v[14] if k == 8 else v[k-1])}" for k in range(8, 14))
    print(f"it{it} graph workgroup={v[17] - v[15]} chain={v[14] - v[15]} readout={v[16] - v[14]} pair+counter={v[17] - v[16]} :: {fw}")
L.dgcnn_debug_phase_clocks(None)
