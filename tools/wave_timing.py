"""Wave-life phase totals of k_gcn_fwd32w (needs a -DDGD_TIMING build: tools/build_variant.sh timing "-DDGD_TIMING",
run with DGCNN_HIP_LIB=dgcnn_amd/variants/lib_timing.so).  usage: python tools/wave_timing.py [batch]"""
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
dbg = torch.zeros(1 << 19, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for _ in range(3): m(b)
    torch.cuda.synchronize()
    L.dgcnn_debug_phase_clocks(ctypes.c_void_p(dbg.data_ptr()))
    m(b)                      # two launches (conv2, conv3) accumulate
    torch.cuda.synchronize()
    L.dgcnn_debug_phase_clocks(None)
d = dbg.cpu().numpy()[65536:].reshape(-1, 8)
livemask = d[:, 7] > 0
lv = d[livemask]
names = ["record", "issue first loads", "first-load latency", "block product", "epilogue"]
tot = lv[:, :5].sum()
print(f"{len(lv)} live waves (x2 launches accumulated); mean life per launch {tot/lv[:,7].sum():.0f} clock64 ticks")
for k, n in enumerate(names): print(f"  {n:20s} mean {lv[:,k].sum()/lv[:,7].sum():9.1f}  share {100*lv[:,k].sum()/tot:5.1f} %")
dead = d[~livemask]
print(f"  dead waves of live items: {(dead[:,5]>0).sum()} mean ticks {dead[:,5][dead[:,5]>0].mean()/2:.0f}; waves of empty workgroups: {(dead[:,6]>0).sum()} mean ticks {dead[:,6][dead[:,6]>0].mean()/2:.0f}")
