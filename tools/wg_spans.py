"""GPU-box helper (RD_TIMING build): wall-clock spans of every workgroup of k_readout_fwd / k_tail_bwd in one training step.
   bash tools/build_variant.sh rdtiming "-DRD_TIMING"; DGCNN_HIP_LIB=dgcnn_amd/variants/lib_rdtiming.so python tools/wg_spans.py 2048"""
import sys, numpy as np, torch
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
dbg = torch.zeros(1024 + 4 * BS + 64, dtype=torch.int64, device="cuda")
for it in range(3): tr.train_step(b, b.y)
torch.cuda.synchronize()
L.dgcnn_debug_phase_clocks(dbg.data_ptr())
tr.train_step(b, b.y); torch.cuda.synchronize()
L.dgcnn_debug_phase_clocks(None)
d = dbg.cpu().numpy()[1024:1024 + 4 * BS].reshape(BS, 4).astype(np.float64) / 100.0      # 100 MHz -> us
sizes = np.diff(np.searchsorted(b.batch.cpu().numpy(), np.arange(BS + 1)))
for name, c0 in (("k_readout_fwd", 0), ("k_tail_bwd", 2)):
    st, en = d[:, c0], d[:, c0 + 1]
    t0 = st.min()
    dur = en - st
    print(f"{name}: kernel span {en.max()-t0:.1f} us; workgroup life p10 {np.percentile(dur,10):.1f} p50 {np.percentile(dur,50):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f} us; "
          f"sum of lives / (256 CUs x span) = {dur.sum()/(256*(en.max()-t0)):.2f} workgroups resident per CU")
    order = np.argsort(st)
    q = [np.percentile(st - t0, p) for p in (25, 50, 75, 100)]
    print(f"   start quartiles {q[0]:.1f} {q[1]:.1f} {q[2]:.1f} {q[3]:.1f} us; first 256 by start: life mean {dur[order[:256]].mean():.1f}; last 256: {dur[order[-256:]].mean():.1f}; corr(life, nodes) {np.corrcoef(dur, sizes)[0,1]:.2f}")
