"""GPU-box helper: where the HOST time of a pipelined Trainer step goes (train and eval), per call: wall time per call of a long
unsynchronised loop (= host time when the GPU is faster than the host) and a cProfile breakdown.
usage: python tools/step_host_profile.py [workload] [graphs per batch]"""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, ".")
from dgcnn_amd import synth
from dgcnn_amd.model import Model
from dgcnn_amd.train import Trainer
name, G = (sys.argv[1] if len(sys.argv) > 1 else "MUTAG"), int(sys.argv[2]) if len(sys.argv) > 2 else 50
sh = synth.SHAPES[name]
bs = [synth.make_batch(name, G, start=G * k).to("cuda") for k in range(40)]
nb = len(bs)
torch.manual_seed(324)
m = Model(sh.num_features, sh.num_classes).to("cuda")
tr = Trainer(m, exclusive_device=True)
for mode in ("eval", "train"):
    m.train(mode == "train")
    fn = tr.train_step if mode == "train" else tr.eval_step
    for k, b in enumerate(bs): fn(b, b.y, next_data=bs[(k + 1) % nb])
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(50):
            for k, b in enumerate(bs): fn(b, b.y, next_data=bs[(k + 1) % nb])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n = 50 * nb
        print(f"{name} x {G} {mode}: host {1e6 * (t1 - t0) / n:.1f} us/call, with final sync {1e6 * (t2 - t0) / n:.1f} us/call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        for k, b in enumerate(bs): fn(b, b.y, next_data=bs[(k + 1) % nb])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)
