"""GPU-box helper: clock64() phase stamps of workgroup 0 of the one-launch training kernel (k_chain_readout_tail) with the
batch's LARGEST graph placed first, i.e. the workgroup whose latency chain is the kernel's duration.
usage: python tools/phase_step_kernel.py [workload] [graphs]"""
import sys, torch
sys.path.insert(0, ".")
from dgcnn_amd import _lib, synth
from dgcnn_amd.batch import collate
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
L = _lib.lib()
name, G = (sys.argv[1] if len(sys.argv) > 1 else "COLLAB"), int(sys.argv[2]) if len(sys.argv) > 2 else 50
sh = synth.SHAPES[name]
force = int(sys.argv[4]) if len(sys.argv) > 4 else None          # force the first graph to this many nodes
graphs = synth.make_graphs(name, G, start=0, force_first_n=force)
order = sorted(range(G), key=lambda i: -graphs[i].num_nodes)
which = sys.argv[3] if len(sys.argv) > 3 else "largest"
if which == "median": order = order[G // 2:] + order[:G // 2]
b = collate([graphs[i] for i in order]).to("cuda")
b.coalesced_undirected = True
print(f"{name} x {G}: workgroup 0 = graph of {graphs[order[0]].num_nodes} nodes (batch max {max(g.num_nodes for g in graphs)})")
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda"); m.train()
tr = Trainer(m, exclusive_device=True)
dbg = torch.zeros(80, dtype=torch.int64, device="cuda")
L.dgcnn_debug_phase_clocks(dbg.data_ptr())
rn = {8: "topk", 9: "gather+Wstage", 10: "conv5", 11: "pool+conv6", 12: "fc1", 13: "fc2+lsm"}
tn = ["(sync)", "stage+dlogit", "fc2 bwd+partial", "fc1^T", "conv6 bwd", "pool/relu", "W5/W6 partials", "scatter"]
gn = {16: "sync", 17: "conv4 bwd", 18: "conv3 bwd", 19: "conv2 bwd", 20: "dW3/dW2 sums", 21: "dW1 sum"}
for it in range(4):
    tr.train_step(b, b.y); torch.cuda.synchronize()
    v = dbg.cpu().tolist()
    fw = " ".join(f"{rn[k]}={v[k] - (v[14] if k == 8 else v[k-1])}" for k in range(8, 14))
    bw = " ".join(f"{tn[k]}={v[k] - (v[13] if k == 0 else v[k-1])}" for k in range(0, 8))
    gb = " ".join(f"{gn[k]}={v[k] - (v[7] if k == 16 else v[k-1])}" for k in range(16, 22)) if v[21] else ""
    if v[29]:      # -DCH_REPEAT_BWD build: slots 24..29 = the first (cold) pass, 16..21 = the second (warm) one
        gb += " || COLD pass: " + " ".join(f"{gn[k]}={v[k + 8] - (v[7] if k == 16 else v[k + 7])}" for k in range(16, 22))
        gb = gb.replace("sync=", "restart=", 1)
    if v[38]:      # -DCH_FINE build: stamps inside conv4's / conv3's backward (wave 0)
        f4 = {40: "loads+stage", 41: "barrier", 42: "product", 43: "gh+ga", 44: "colsums", 45: "image store"}
        f3 = {32: "x2 transpose+loads", 33: "product", 34: "gh transpose", 35: "gx+dW mfma", 36: "epilogue+colsum", 37: "barrier", 38: "image store"}
        gb += " || conv4 fine: " + " ".join(f"{f4[k]}={v[k] - (v[16] if k == 40 else v[k-1])}" for k in range(40, 46))
        gb += " || conv3 fine: " + " ".join(f"{f3[k]}={v[k] - (v[17] if k == 32 else v[k-1])}" for k in range(32, 39))
    if v[48 + 10]:  # -DCH_FINE build: stamps inside the chain forward (wave 0)
        cn = {0: "set-up", 14: "bitmap+dinv stage", 15: "xs split+stage", 1: "zero fill", 2: "barrier", 3: "rows->regs", 4: "conv1", 5: "hs2 store+barrier",
              6: "conv2", 7: "hs3 store+barrier", 8: "conv3", 9: "barrier", 10: "conv4"}
        order = [0, 14, 15, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
        prev = v[15]
        parts = []
        for k in order:
            parts.append(f"{cn[k]}={v[48 + k] - prev}"); prev = v[48 + k]
        gb += " || chain fine: " + " ".join(parts)
    if v[65]:      # -DRD_FINE build: stamps inside the top-k selection
        gb += " || topk fine: " + " ".join(f"{nm}={v[60 + k] - (v[14] if k == 0 else v[59 + k])}" for k, nm in enumerate(["entry", "barrier", "keys+barrier", "count", "shuffle+sel", "barrier"])) + f" rest={v[8] - v[65]}"
    print(f"it{it} kernel={max(v[21], v[17], v[7]) - v[15]} chain={v[14]-v[15]} readout+tail={v[7]-v[14]} :: FWD {fw} :: BWD {bw} :: GCN-BWD {gb}")
L.dgcnn_debug_phase_clocks(None)
