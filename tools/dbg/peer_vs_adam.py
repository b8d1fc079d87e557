import sys, ctypes, torch
sys.path.insert(0, ".")
from dgcnn_amd import _lib
from dgcnn_amd.dist import PeerExchange
L = _lib.lib()
n = 104128
torch.manual_seed(0)
g = torch.randn(n, device="cuda") * 1e-2
p0 = torch.randn(n, device="cuda"); m0 = torch.randn(n, device="cuda") * 1e-3; v0 = torch.rand(n, device="cuda") * 1e-4
s = torch.cuda.current_stream().cuda_stream
pa, ma, va, ga = p0.clone(), m0.clone(), v0.clone(), g.clone()
_lib.check(L.dgcnn_adam_step(pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), n, 3, 1e-3, 0.9, 0.999, 1e-8, 0, s), "adam")
px = PeerExchange(n, device="cuda:0")
pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
px.grad_tensor(1).copy_(g)
px.step(1, pb, mb, vb, 3, 1e-3, (0.9, 0.999), 1e-8, s)
torch.cuda.synchronize()
print("params equal", torch.equal(pa, pb), float((pa - pb).abs().max()), "m", torch.equal(ma, mb), "v", torch.equal(va, vb))
