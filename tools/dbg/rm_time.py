import sys, time, torch
sys.path.insert(0, ".")
from dgcnn_amd import synth, _lib
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
sh = synth.SHAPES["COLLAB"]
bs = [b.to("cuda") for b in synth.make_batches("COLLAB", 100, 50)]
m = Model(sh.num_features, sh.num_classes).to("cuda"); tr = Trainer(m)
tr.train_step(bs[0], bs[0].y, next_data=bs[1]); tr.train_step(bs[1], bs[1].y); torch.cuda.synchronize()
for name, fn in (("fused read_metrics", tr.read_metrics),
                 ("three syncs", lambda: (tr.metrics.tolist(), [_lib.ws_view(sl["ws"], "err", *sl["dims"]).cpu().tolist() for sl in tr._slots if sl.get("dims")]))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    print(name, 1e6 * (time.perf_counter() - t0) / 200, "us")
