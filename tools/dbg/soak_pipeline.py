"""GPU-box helper: 360 training steps over batches of 50..700 COLLAB graphs, pipelined (side stream / riders by size) vs plain:
   parameters and metrics must be identical bit for bit."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from dgcnn_amd import synth
from dgcnn_amd.batch import collate
from dgcnn_amd.train import Trainer
from parity_util import make_model
sh = synth.SHAPES["COLLAB"]
graphs = synth.make_graphs("COLLAB", 3000, start=100)
cuts = [0, 700, 1400, 1450, 2100, 2400, 3000]
batches = [collate(graphs[a:b]).to("cuda") for a, b in zip(cuts[:-1], cuts[1:])]
outs = []
for pipelined in (False, True):
    m = make_model(sh.num_features, sh.num_classes)
    m.train(); m._seed_base, m._fwd_count = 9, 0
    tr = Trainer(m)
    for ep in range(60):
        if pipelined: tr.train_epoch(batches, 3000)
        else:
            tr.reset_metrics()
            for b in batches: tr.train_step(b, b.y)
    torch.cuda.synchronize(); m.check_errors()
    outs.append((m.flat_params.clone(), tr.metrics.clone()))
print("steps", 60 * len(batches), "params equal:", torch.equal(outs[0][0], outs[1][0]), "metrics equal:", torch.equal(outs[0][1], outs[1][1]), "finite:", bool(torch.isfinite(outs[1][0]).all()))
