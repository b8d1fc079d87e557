"""GPU-box helper: wall time of a real training epoch (loader + step), host-collated vs device-collated batches."""
import sys, time, torch
sys.path.insert(0, ".")
from dgcnn_amd import synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
from dgcnn_amd.tudataset import GraphLoader
from dgcnn_amd.device_data import DeviceDataset, DeviceLoader, PreparedDataset
name, G = (sys.argv[1] if len(sys.argv) > 1 else "COLLAB"), int(sys.argv[2]) if len(sys.argv) > 2 else 1000
sh = synth.SHAPES[name]
graphs = synth.make_graphs(name, G, labels="structure")
for kind in ("host GraphLoader", "DeviceLoader", "DeviceLoader(prepared)"):
    torch.manual_seed(324)          # same initial weights for every loader: the three runs train the same trajectory
    m = Model(sh.num_features, sh.num_classes).to("cuda"); tr = Trainer(m, exclusive_device=True)
    gen = torch.Generator().manual_seed(1)
    ld = GraphLoader(graphs, 50, shuffle=True, generator=gen, device="cuda") if kind.startswith("host") else \
        (DeviceLoader(PreparedDataset(graphs), 50, shuffle=True, generator=gen, prepared=True) if "prepared" in kind else
         DeviceLoader(DeviceDataset(graphs), 50, shuffle=True, generator=gen))
    tr.train_epoch(ld, G); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): loss, acc = tr.train_epoch(ld, G)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{kind:24s}: {1e3 * dt:8.1f} ms/epoch of {G} graphs = {1e6 * dt / len(ld):7.1f} us/batch  ({G / dt:9.0f} graphs/s)  loss {loss:.3f} acc {acc:.1f}")
