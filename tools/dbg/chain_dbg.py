import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, numpy as np
from dgcnn_amd import synth
from parity_util import make_model, cpu_state_dict, gpu_xcat
from oracle import ref_dense
name, bs = sys.argv[1], int(sys.argv[2])
sh = synth.SHAPES[name]
b = synth.make_batch(name, bs, start=1000)
m = make_model(sh.num_features, sh.num_classes); sd = cpu_state_dict(m)
m.eval()
_, aux = ref_dense.forward_dense(sd, b.x, b.edge_index, b.batch, b.num_graphs, return_all=True)
ref = aux["xcat"].detach()
ptr = aux["ptr"]
for mode in ("chain", "dense", "sparse"):
    m.agg_mode = "sparse" if mode == "sparse" else "dense"; m.use_chain = mode == "chain"
    with torch.no_grad(): m(b.to("cuda"))
    xc = gpu_xcat(m).double()
    d = (xc - ref).abs()
    print(mode, "x1 %.2e x2 %.2e x3 %.2e x4 %.2e" % (d[:, :32].max(), d[:, 32:64].max(), d[:, 64:96].max(), d[:, 96].max()))
    if mode == "chain":
        ax = m.last_workspace_view("ax").cpu()
        bad = (d[:, :32].max(1).values > 1e-4).nonzero().flatten().tolist()
        print(" bad x1 rows:", bad[:40], "of", d.shape[0])
        print(" ptr", [int(v) for v in ptr][:12])
        badc = (d[:, :32].max(0).values > 1e-4).nonzero().flatten().tolist()
        print(" bad x1 cols:", badc)
