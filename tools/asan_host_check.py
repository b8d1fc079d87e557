"""Host-side AddressSanitizer pass over libdgcnn_hip.so's entry points: every host code path that runs
before a kernel launch -- layout queries, argument validation, the step-args plumbing, the host prefix sums of
dgcnn_collate_ids, pipeline create/destroy -- called with valid and invalid arguments.  Runs WITHOUT a GPU (HIP calls
fail cleanly and the entry points return their error codes); launched by tests/test_abi_and_host.py under
LD_PRELOAD=libclang_rt.asan with the -fsanitize=address build of tools/build_variant.sh.  Prints ASAN_HOST_OK at the end."""
import ctypes, os, sys
import numpy as np
lib = ctypes.CDLL(sys.argv[1])
c = ctypes
lib.dgcnn_param_layout.restype = c.c_int64
lib.dgcnn_workspace_bytes.restype = c.c_int64
lib.dgcnn_workspace_offset.restype = c.c_int64
lib.dgcnn_gcn_bwd_scratch_bytes.restype = c.c_int64
lib.dgcnn_dense_table_ints.restype = c.c_int64
offs = (c.c_int64 * 16)()
for F, C in ((1, 3), (8, 2), (90, 2), (512, 64), (0, 2), (513, 2), (5, 65)):
    lib.dgcnn_param_layout(F, C, offs); lib.dgcnn_param_layout(F, C, None)
for N, E, B in ((1, 0, 1), (3800, 140000, 50), (150000, 5700000, 2048), (-1, 0, 1), (10, -5, 2)):
    lib.dgcnn_workspace_bytes(N, E, B, 1, 3)
    for name in (b"err", b"rowptr", b"adjbits", b"dmap", b"wg_t2", b"P32", b"nope"):
        lib.dgcnn_workspace_offset(name, N, E, B, 1, 3)
    lib.dgcnn_dense_table_ints(N, B); lib.dgcnn_gcn_bwd_scratch_bytes(N, 32, 32)
lib.dgcnn_workspace_offset(None, 1, 1, 1, 1, 1)
h = c.c_void_p()
assert lib.dgcnn_pipeline_create(c.byref(h)) == 0
assert lib.dgcnn_pipeline_train_step(h, None, None, None) == -1
assert lib.dgcnn_pipeline_destroy(h) == 0 and lib.dgcnn_pipeline_destroy(None) == -1
# null / bad-size argument validation of the compute entry points (all must return an error code, never touch memory)
assert lib.dgcnn_model_forward(0, 0, 0, 1, 3, None, None, None, None, None, None, 0, c.c_uint64(0), 0, 0, 0, c.c_uint32(1), None) == -1
assert lib.dgcnn_model_backward(5, 0, 1, 1, 3, None, None, None, None, None, None, c.c_float(0), 0, None, None, 0, 0, None) == -1
assert lib.dgcnn_adam_step(None, None, None, None, c.c_int64(4), c.c_int64(1), c.c_float(1e-3), c.c_float(.9), c.c_float(.999), c.c_float(1e-8), 0, None) == -1
assert lib.dgcnn_gcn_fwd(4, None, None, None, None, 1, None, None, 32, None, None, 0, None, None) == -1
assert lib.dgcnn_gcn_bwd(4, None, None, None, None, 32, None, None, 32, 0, None, None, None, None, None, 0, None, None, None, c.c_int64(0), None) == -1
assert lib.dgcnn_allreduce_adam_step(0, 0, None, None, c.c_uint32(1), None, None, None, None, c.c_int64(4), c.c_int64(1), c.c_float(0), c.c_float(0), c.c_float(0), c.c_float(0), None, None) == -1
# dgcnn_collate_ids: the host prefix sums run before any HIP call; a too-small capacity returns -3 with the sizes filled
G = 40
nodes = np.arange(5, 5 + G, dtype=np.int64); edges = nodes * 4
ids = np.array([3, 39, 0, 17, 17], dtype=np.int64)
meta = np.zeros(3 * len(ids) + 2, dtype=np.int64)
sizes = np.zeros(4, dtype=np.int64)
P = lambda a: a.ctypes.data_as(c.c_void_p)
one = c.c_void_p(16)     # non-null dummy device pointers: never dereferenced on the host
rc = lib.dgcnn_collate_ids(len(ids), 1, P(ids), None, P(nodes), P(edges), c.c_int64(G), P(meta), one, None, c.c_int64(10), one, one, one, one,
                           one, c.c_int64(0), c.c_int64(0), None, None, None, one, P(sizes), None)
assert rc == -3 and sizes[0] == int(nodes[ids].sum()) and sizes[2] == int(nodes[ids].max()), (rc, sizes)
bad = np.array([3, 40], dtype=np.int64)
assert lib.dgcnn_collate_ids(2, 1, P(bad), None, P(nodes), P(edges), c.c_int64(G), P(meta), one, None, c.c_int64(10), one, one, one, one,
                             one, c.c_int64(0), c.c_int64(0), None, None, None, one, P(sizes), None) == -1
# prepared dataset: descriptor validation runs on the host before any launch
class DS(c.Structure):
    _fields_ = [("G", c.c_int64), ("Ntot", c.c_int64), ("Etot", c.c_int64), ("F", c.c_int32), ("r", c.c_int32)] + \
               [(n, c.c_void_p) for n in ("node_ptr", "y", "x", "rowptr", "colidx", "dinv", "xs", "adj_bits")]
d = DS(); d.G, d.Ntot, d.Etot, d.F = 10, 100, 400, 1
assert lib.dgcnn_dataset_prepare(None, None, None, None, None, 1, None) == -1
assert lib.dgcnn_dataset_prepare(c.byref(d), one, one, one, one, 1, None) == -1            # output arrays missing
for n in ("node_ptr", "y", "x", "rowptr", "colidx", "dinv", "xs", "adj_bits"): setattr(d, n, 16)
assert lib.dgcnn_dataset_prepare(c.byref(d), one, one, one, one, 0, None) == -3            # no coalesced-undirected promise
d.Ntot = 1 << 31
assert lib.dgcnn_dataset_prepare(c.byref(d), one, one, one, one, 1, None) == -3            # beyond int32 indices
d.Ntot = 100
assert lib.dgcnn_assemble(None, 5, 50, 200, 3, one, one, one, one, one, one, one, 1, 20, c.c_uint32(1), None) == -1
assert lib.dgcnn_assemble(c.byref(d), 5, 50, 200, 3, one, one, one, one, one, one, one, 0, 20, c.c_uint32(1), None) == -3
assert lib.dgcnn_assemble(c.byref(d), 5, 50, 200, 3, None, one, one, one, one, one, one, 1, 20, c.c_uint32(1), None) == -1
assert lib.dgcnn_assemble(c.byref(d), 5, 50, 200, 3, one, one, one, one, one, one, one, 1, 20, c.c_uint32(0), None) == -1
print("ASAN_HOST_OK")
