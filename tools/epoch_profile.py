"""GPU-box helper: host-time profile (cProfile) of real training epochs through DeviceLoader / DeviceLoader(prepared=True)."""
import sys, time, torch, cProfile, pstats, io
sys.path.insert(0, ".")
from dgcnn_amd import synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
from dgcnn_amd.device_data import DeviceDataset, DeviceLoader, PreparedDataset
name, G = "COLLAB", 1000
sh = synth.SHAPES[name]
graphs = synth.make_graphs(name, G, labels="structure")
for prepared in (False, True):
    torch.manual_seed(324)
    m = Model(sh.num_features, sh.num_classes).to("cuda"); tr = Trainer(m, exclusive_device=True)
    gen = torch.Generator().manual_seed(1)
    ld = DeviceLoader(PreparedDataset(graphs) if prepared else DeviceDataset(graphs), 50, shuffle=True, generator=gen, prepared=prepared)
    tr.train_epoch(ld, G); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for _ in range(5): tr.train_epoch(ld, G)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    pr.disable()
    print(f"prepared={prepared}: {1e6 * dt / len(ld):.1f} us/batch (under cProfile)")
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:2600])
