"""Per-workgroup phase clocks of k_gcn_fwd32d (needs a -DDGD_TIMING build: tools/build_variant.sh timing "-DDGD_TIMING",
run with DGCNN_HIP_LIB=dgcnn_amd/variants/lib_timing.so).  usage: python tools/dense_timing.py [batch]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgcnn_amd import _lib, synth
from dgcnn_amd.model import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = _lib.lib()
sh = synth.SHAPES["COLLAB"]
b = synth.make_batch("COLLAB", B, start=0).to("cuda")
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).cuda().eval()
m.agg_mode = "dense"
dbg = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(3): m(b)
    torch.cuda.synchronize()
    L.dgcnn_debug_phase_clocks(ctypes.c_void_p(dbg.data_ptr()))
    m(b)                      # conv2, conv3 both write; conv3 (MODE 1) writes last
    torch.cuda.synchronize()
    L.dgcnn_debug_phase_clocks(None)
d = dbg.cpu().numpy().reshape(-1, 8)
d = d[(d[:, 6] > 0) & (d[:, :6].max(1) < 10_000_000)]
names = ["prologue", "barrier", "mfma", "wait+lds store", "issue loads", "epilogue", "stages"]
tot = d[:, :6].sum(1)
t0 = d[:, 7].min()
print(f"{len(d)} workgroups; total cycles per WG: mean {tot.mean():.0f}  min {tot.min()}  max {tot.max()}  p95 {np.percentile(tot,95):.0f}")
print(f"start skew (cycles): max {int((d[:,7]-t0).max())};  end = start+total: max {int((d[:,7]-t0+tot).max())}")
for k, n in enumerate(names):
    print(f"  {n:16s} mean {d[:,k].mean():9.0f}   max {d[:,k].max():9d}   share {100*d[:,k].sum()/tot.sum():5.1f} %" if k < 6 else f"  {n:16s} mean {d[:,k].mean():.2f} max {d[:,k].max()}")
st = d[:, 6]
for s in sorted(set(st.tolist())):
    sel = st == s
    print(f"  WGs with {s:3d} stages: {sel.sum():5d}  mean total {tot[sel].mean():8.0f}")
