"""Per-workgroup phase clocks of the persistent graph-chain forward (needs a -DCH_TIMING build:
tools/build_variant.sh chtiming "-DCH_TIMING"; run with DGCNN_HIP_LIB=dgcnn_amd/variants/lib_chtiming.so).
usage: python tools/chain_timing.py [batch] [workload]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgcnn_amd import _lib, synth
from dgcnn_amd.model import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W = sys.argv[2] if len(sys.argv) > 2 else "COLLAB"
L = _lib.lib()
sh = synth.SHAPES[W]
b = synth.make_batch(W, B, start=0).to("cuda")
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).cuda().eval()
m.agg_mode, m.use_chain = "dense", True
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(3): m(b)
    torch.cuda.synchronize()
    L.dgcnn_debug_phase_clocks(ctypes.c_void_p(dbg.data_ptr()))
    m(b)
    torch.cuda.synchronize()
    L.dgcnn_debug_phase_clocks(None)
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[(d[:, 11] > 0) & (d[:, :12].max(1) < 100_000_000)]       # (the readout kernels stamp the first words of the same buffer)
names = ["setup", "stage(wait+stores)", "barrier0", "prefetch issue", "conv1", "bar1(+image store)", "conv2", "bar2(+image store)", "conv3", "bar3", "conv4"]
tot = d[:, :11].sum(1)
print(f"{len(d)} workgroups, graphs per WG mean {d[:,11].mean():.2f} max {d[:,11].max()}; cycles per WG mean {tot.mean():.0f} min {tot.min()} max {tot.max()}")
ng = d[:, 11].sum()
for k, n in enumerate(names):
    print(f"  {n:20s} per WG {d[:,k].mean():9.0f}  per graph {d[:,k].sum()/ng:8.0f}  share {100*d[:,k].sum()/tot.sum():5.1f} %")

for k, n in ((14, "  stage: bitmap+dv"), (15, "  stage: x split")):
    print(f"  {n:20s} per WG {d[:,k].mean():9.0f}  per graph {d[:,k].sum()/ng:8.0f}")
if d[:, 12].max() > 0:
    t0 = d[:, 12].min()
    st, en = (d[:, 12] - t0) / 100.0, (d[:, 13] - t0) / 100.0      # 100 MHz constant clock -> us
    print(f"start (us): p50 {np.percentile(st,50):.2f} p75 {np.percentile(st,75):.2f} p90 {np.percentile(st,90):.2f} max {st.max():.2f};  end: p50 {np.percentile(en,50):.2f} max {en.max():.2f}")
    print("  starts after 5 us:", int((st > 5).sum()), "of", len(st))
