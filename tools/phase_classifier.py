"""GPU-box helper (CL_TIMING build): clock64() phase deltas of workgroup 0 / thread 0 of k_classifier (large batches).
   bash tools/build_variant.sh cltiming "-DCL_TIMING"; DGCNN_HIP_LIB=$PWD/dgcnn_amd/variants/lib_cltiming.so python tools/phase_classifier.py 2048"""
import sys, torch
sys.path.insert(0, ".")
from dgcnn_amd import _lib, synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
L = _lib.lib()
sh = synth.SHAPES["COLLAB"]
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
b = synth.make_batch("COLLAB", BS, start=0).to("cuda")
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda"); m.train()
tr = Trainer(m, exclusive_device=True)
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
L.dgcnn_debug_phase_clocks(dbg.data_ptr())
names = ["start", "loads+stage", "classifier_1 (mfma)+stores", "barrier", "classifier_2", "log_softmax+loss", "gz1+partials", "classifier_1 back"]
for it in range(5):
    dbg.zero_(); dbg[41] = 2**62
    tr.train_step(b, b.y); torch.cuda.synchronize()
    v = dbg.cpu().tolist()[32:]
    print(f"   all workgroups: first start -> last end {(v[8]-v[9])/100:.2f} us; workgroup 0 ends at {(v[10]-v[9])/100:.2f} us")
    print(f"it{it} total={v[7]-v[0]} :: " + " ".join(f"{names[k]}={v[k]-v[k-1]}" for k in range(1, 8)))
L.dgcnn_debug_phase_clocks(None)
