// api.hip -- extern "C" entry points of libdgcnn_hip.so (see include/dgcnn_hip.h) and the
// orchestration of the whole-model forward / backward as a chain of launches on one stream.
#include "dg_common.h"
#include "dg_prep.h"
#include <cstdlib>

// one-shot, thread-local profiling request (see dgcnn_profile_next_forward)
static thread_local int g_prof_which = -1;
static thread_local hipEvent_t g_prof_a = nullptr, g_prof_b = nullptr;
#define DG_PROF_A(idx) (g_prof_which == (idx) ? g_prof_a : nullptr)
#define DG_PROF_B(idx) (g_prof_which == (idx) ? g_prof_b : nullptr)

#define DG_TRY(expr) do { const int rc__ = (expr); if (rc__ != DGCNN_OK) return rc__; } while (0)

extern "C" {

int dgcnn_profile_next_forward(int which, void* ev_start, void* ev_stop) {
  if (which < 0 || which > 2 || !ev_start || !ev_stop) return DGCNN_EINVAL;
  g_prof_which = which; g_prof_a = (hipEvent_t)ev_start; g_prof_b = (hipEvent_t)ev_stop;
  return DGCNN_OK;
}
int dgcnn_event_create(void** ev) {
  if (!ev) return DGCNN_EINVAL;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return DGCNN_ELAUNCH;
  *ev = (void*)e;
  return DGCNN_OK;
}
int dgcnn_event_record(void* ev, dgcnn_stream_t stream) {
  if (!ev) return DGCNN_EINVAL;
  return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? DGCNN_OK : DGCNN_ELAUNCH;
}
int dgcnn_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
  if (!ev_start || !ev_stop || !ms) return DGCNN_EINVAL;
  if (hipEventSynchronize((hipEvent_t)ev_stop) != hipSuccess) return DGCNN_ELAUNCH;
  return hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop) == hipSuccess ? DGCNN_OK : DGCNN_ELAUNCH;
}
int dgcnn_event_destroy(void* ev) {
  if (!ev) return DGCNN_EINVAL;
  return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? DGCNN_OK : DGCNN_ELAUNCH;
}

int dgcnn_debug_phase_clocks(void* dev_u64x16) {
  dg_fused_set_debug(reinterpret_cast<unsigned long long*>(dev_u64x16));
  return DGCNN_OK;
}

int dgcnn_fused_fits(int max_nodes, int max_edges, int F) {
  if (F < 1 || F > DGCNN_MAX_F) return 0;
  return dg_fused_fits(max_nodes, max_edges, F);
}

int dgcnn_fused_max_nodes(int F) {
  if (F < 1 || F > DGCNN_MAX_F) return 0;
  return dg_fused_max_nodes(F);
}

int dgcnn_version(void) { return DGCNN_ABI_VERSION; }

int64_t dgcnn_param_layout(int F, int C, int64_t offsets[DGCNN_NUM_PARAM_SEGMENTS]) {
  DgParams p;
  const int rc = dg_param_layout(F, C, &p);
  if (rc != DGCNN_OK) return rc;
  if (offsets)
    for (int i = 0; i < DGCNN_NUM_PARAM_SEGMENTS; ++i) offsets[i] = p.off[i];
  return p.total;
}

int64_t dgcnn_workspace_bytes(int N, int E, int B, int F, int C) {
  DgWs w;
  const int rc = dg_ws_layout(N, E, B, F, C, &w);
  return rc != DGCNN_OK ? rc : w.total;
}

int64_t dgcnn_workspace_offset(const char* name, int N, int E, int B, int F, int C) {
  DgWs w;
  if (!name || dg_ws_layout(N, E, B, F, C, &w) != DGCNN_OK) return -1;
#define X(n) if (strcmp(name, #n) == 0) return w.n;
  DG_WS_REGIONS(X)
#undef X
  if (strcmp(name, "P32") == 0) return w.P32;
  if (strcmp(name, "P1") == 0) return w.P1;
  return -1;
}

// DGCNN_FLAG_COALESCED_UNDIRECTED promises a symmetric edge list: its CSR by source equals its CSR by target, so the
// model-level preparation writes ONE copy and the backward's gather kernels read it (the flag must be the same in the calls of
// one batch: prepare / forward / backward -- include/dgcnn_hip.h).
static inline bool dg_csr_symmetric(int flags, int E) { return (flags & DGCNN_FLAG_COALESCED_UNDIRECTED) && E > 0; }

int dgcnn_graph_prep(const int64_t* edge_index, int E, const int64_t* batch, int N, int B,
                     int32_t* rowptr, int32_t* colidx, int32_t* rowptr_t, int32_t* colidx_t,
                     float* dinv, int32_t* graph_ptr, int32_t* scratch, int32_t* err_flag, int flags,
                     uint32_t* adj_bits, int32_t* item_table, dgcnn_stream_t stream) {
  if (!batch || !rowptr || !rowptr_t || !dinv || !graph_ptr || !scratch || !err_flag) return DGCNN_EINVAL;
  if (E > 0 && (!edge_index || !colidx || !colidx_t)) return DGCNN_EINVAL;
  if ((adj_bits == nullptr) != (item_table == nullptr)) return DGCNN_EINVAL;
  if (adj_bits && (!(flags & DGCNN_FLAG_COALESCED_UNDIRECTED) || E <= 0)) return DGCNN_EUNSUPPORTED;   // one bitmap for A and A^T
  // stand-alone entry: plain semantics "err_flag[0..1] != 0 on error" -> clear, then tag with epoch 1
  if (hipMemsetAsync(err_flag, 0, 4 * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return DGCNN_ELAUNCH;
  // graph_eptr (first edge position per graph) is an internal by-product: park it in the scratch tail
  return dg_launch_prep(edge_index, E, batch, N, B, rowptr, colidx, rowptr_t, colidx_t, dinv, graph_ptr,
                        scratch + 2 * (N + 1), scratch, scratch + (N + 1), err_flag, flags, 1u, (hipStream_t)stream,
                        nullptr, nullptr, adj_bits, item_table);
}

int64_t dgcnn_dense_table_ints(int N, int B) { return (N < 0 || B < 0) ? DGCNN_EINVAL : dgd_table_ints(N, B); }
int64_t dgcnn_dense_bitmap_words(int N) { return N < 0 ? DGCNN_EINVAL : 31 * (int64_t)N; }

static bool dg_view_ok(const dgcnn_dense_view* v) { return v && v->graph_ptr && v->item_table && v->adj_bits && v->B > 0; }
static DgDense dg_dense_of(const dgcnn_dense_view* v, int N) {
  DgDense G;
  G.graph_ptr = v->graph_ptr; G.dmap = v->item_table; G.bits = v->adj_bits; G.N = N; G.B = v->B; G.NW = dgd_num_items(N, v->B);
  return G;
}

int dgcnn_gcn_fwd(int N, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                  const float* x, int Fin, const float* W, const float* bias, int Fout,
                  float* out, void* hs_scratch, int flags, const dgcnn_dense_view* dense, dgcnn_stream_t stream) {
  if (!rowptr || !dinv || !x || !W || !bias || !out || !hs_scratch) return DGCNN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int bf16 = (flags & DGCNN_FLAG_BF16) ? 1 : 0;
  const bool use_dense = (flags & DGCNN_FLAG_AGG_DENSE) != 0 || bf16;
  if (use_dense && !dg_view_ok(dense)) return DGCNN_EINVAL;
  if (bf16 && Fout != 32) return DGCNN_EUNSUPPORTED;           // the 32 -> 1 layer's scalar stays fp32
  int rc = dg_launch_lin_first(N, Fin, x, W, dinv, reinterpret_cast<float*>(hs_scratch), Fout, s, bf16);
  if (rc != DGCNN_OK) return rc;
  if (use_dense) {
    const DgDense G = dg_dense_of(dense, N);
    if (Fout == 32) return dg_launch_gcn_fwd32d(2, bf16, 0, &G, dinv, hs_scratch, bias, out, nullptr, nullptr, s);
    if (Fout == 1) return dg_launch_gcn_fwd1d(&G, dinv, reinterpret_cast<const float*>(hs_scratch), bias, out, s);
    return DGCNN_EUNSUPPORTED;
  }
  const float* hs = reinterpret_cast<const float*>(hs_scratch);
  if (Fout == 32) return dg_launch_gcn_fwd32(2, N, rowptr, colidx, dinv, hs, bias, out, nullptr, nullptr, s);
  if (Fout == 1) return dg_launch_gcn_fwd1(N, rowptr, colidx, dinv, hs, bias, out, s);
  return DGCNN_EUNSUPPORTED;
}

// partial-row scratch of dgcnn_gcn_bwd, in floats: [P x 1056] | [P x 32*Fin or 32*Fa] (P = the production grid size)
int64_t dgcnn_gcn_bwd_scratch_bytes(int N, int Fin, int Fout) {
  if (N <= 0 || Fin < 1 || Fin > DGCNN_MAX_F || (Fout != 32 && Fout != 1)) return DGCNN_EINVAL;
  const int64_t P = dg_grid32(N);
  return 4 * (P * 1056 + P * 32 * (int64_t)(Fin > 32 ? Fin : 32));
}

int dgcnn_gcn_bwd(int N, const int32_t* rowptr_t, const int32_t* colidx_t, const float* dinv, const float* gas, int Fout,
                  const float* W, const float* x_prev, int Fin, int first, const float* gp_prev, float* gas_prev,
                  float* gW, float* gb_prev, const float* ax, int Fa, float* gW_af, const dgcnn_dense_view* dense,
                  void* scratch, int64_t scratch_bytes, dgcnn_stream_t stream) {
  if (!rowptr_t || !dinv || !gas || !x_prev || !gW || !scratch || N <= 0) return DGCNN_EINVAL;
  const int64_t need = dgcnn_gcn_bwd_scratch_bytes(N, Fin, Fout);
  if (need < 0 || scratch_bytes < need) return DGCNN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int P = dg_grid32(N);                      // = dg_grid1(N): both production grids
  float* part = reinterpret_cast<float*>(scratch);
  float* part1 = part + (size_t)P * 1056;
  const bool use_dense = dense != nullptr;
  if (use_dense && !dg_view_ok(dense)) return DGCNN_EINVAL;
  // the gather kernels walk the transposed CSR: every branch of the gather form, and conv1's own backward (`first`) in
  // BOTH forms (its operand is the raw [N,Fin] input, there is no dense kernel for it)
  if (!colidx_t && (!use_dense || first)) return DGCNN_EINVAL;
  DgDense G{};
  if (use_dense) G = dg_dense_of(dense, N);
  if (Fout == 1) {                                 // conv4 form: k_gcn_bwd1 / k_gcn_bwd1d
    if (!W || !gp_prev || !gas_prev || !gb_prev || Fin != 32 || first) return DGCNN_EINVAL;
    if (use_dense) DG_TRY(dg_launch_gcn_bwd1d(&G, dinv, gas, W, x_prev, gp_prev, gas_prev, part, P, s));
    else DG_TRY(dg_launch_gcn_bwd1(N, rowptr_t, colidx_t, dinv, gas, W, x_prev, gp_prev, gas_prev, part, P, s));
    const DgRedSeg segs[2] = {{32, P, 64, part, gW}, {32, P, 64, part + 32, gb_prev}};
    return dg_launch_reduce_cols(2, segs, s);
  }
  if (Fout != 32) return DGCNN_EUNSUPPORTED;
  if (first) {                                     // conv1, linear-first: only dW_1 [32,Fin] (the gather kernel in both forms)
    if (Fin < 1 || Fin > DGCNN_MAX_F) return DGCNN_EINVAL;
    DG_TRY(dg_launch_gcn_bwd32(1, N, Fin, rowptr_t, colidx_t, dinv, gas, nullptr, x_prev, nullptr, nullptr, part1, P, s));
    const DgRedSeg seg = {32 * Fin, P, 32 * Fin, part1, gW};
    return dg_launch_reduce_cols(1, &seg, s);
  }
  if (!W || !gp_prev || !gb_prev || Fin != 32) return DGCNN_EINVAL;
  if (ax) {                                        // conv2 form carrying conv1's weight gradient (aggregate-first conv1)
    if (Fa < 1 || Fa > DG_AF_MAX_F || !gW_af) return DGCNN_EINVAL;
    if (use_dense) DG_TRY(dg_launch_gcn_bwd32d(&G, dinv, gas, W, x_prev, gp_prev, nullptr, part, P, s, ax, Fa, part1));
    else DG_TRY(dg_launch_gcn_bwd32(0, N, 32, rowptr_t, colidx_t, dinv, gas, W, x_prev, gp_prev, nullptr, part, P, s, ax, Fa, part1));
    const DgRedSeg segs[3] = {{1024, P, 1056, part, gW}, {32, P, 1056, part + 1024, gb_prev}, {32 * Fa, P, 32 * Fa, part1, gW_af}};
    return dg_launch_reduce_cols(3, segs, s);
  }
  if (!gas_prev) return DGCNN_EINVAL;
  if (use_dense) DG_TRY(dg_launch_gcn_bwd32d(&G, dinv, gas, W, x_prev, gp_prev, gas_prev, part, P, s));
  else DG_TRY(dg_launch_gcn_bwd32(0, N, 32, rowptr_t, colidx_t, dinv, gas, W, x_prev, gp_prev, gas_prev, part, P, s));
  const DgRedSeg segs[2] = {{1024, P, 1056, part, gW}, {32, P, 1056, part + 1024, gb_prev}};
  return dg_launch_reduce_cols(2, segs, s);
}

int dgcnn_sortpool_fwd(int N, int B, const int32_t* graph_ptr, const float* x1, const float* x2,
                       const float* x3, const float* x4, float* pooled, int32_t* perm,
                       dgcnn_stream_t stream) {
  if (!graph_ptr || !x1 || !x2 || !x3 || !x4 || !pooled || !perm) return DGCNN_EINVAL;
  return dg_launch_sortpool_fwd(N, B, graph_ptr, x1, x2, x3, x4, pooled, perm, (hipStream_t)stream);
}

int dgcnn_sortpool_bwd(int N, int B, const int32_t* graph_ptr, const int32_t* perm, const float* gpooled,
                       float* g1, float* g2, float* g3, float* g4, dgcnn_stream_t stream) {
  if (!graph_ptr || !perm || !gpooled || !g1 || !g2 || !g3 || !g4) return DGCNN_EINVAL;
  return dg_launch_sortpool_bwd(N, B, graph_ptr, perm, gpooled, g1, g2, g3, g4, (hipStream_t)stream);
}

// Aggregation form of a batch: dense per-graph blocks on the matrix cores (gcn_dense.hip) or CSR gather (gcn.hip).
// A pure function of host-known numbers, so graph preparation (which builds the bitmap only for the dense form) and
// the forward / backward calls of the same batch always agree.  Dense needs the coalesced + undirected promise (one
// bitmap serves A and A^T) and a per-graph node bound <= 512.  Cost model (MI355X, measured): the block
// product costs ~K_g SIMD-cycles per node row (K_g = n_g rounded up to 32), the gather ~28 per edge.
static bool dg_use_dense(int N, int E, int B, int flags, int max_nodes) {
  if (flags & DGCNN_FLAG_AGG_SPARSE) return false;
  if (!(flags & DGCNN_FLAG_COALESCED_UNDIRECTED) || E <= 0) return false;
  if (max_nodes <= 0 || max_nodes > DGD_MAXN) return false;
  if (dgd_num_items(N, B) > 100000) return false;        // (a workgroup caches at most 128 item records: gcn_dense.hip)
  if (flags & DGCNN_FLAG_AGG_DENSE) return true;
  if (flags & DGCNN_FLAG_BF16) return true;              // the bf16 leg exists in the dense form only
  // Automatic choice.  (1) The dense kernels are persistent pipelines with a fixed prologue (~2.5 us: item records,
  // first stage) and their bitmap costs extra graph-preparation work; below a few hundred work items per launch the
  // one-tile-per-workgroup gather kernels are as fast or faster (measured, COLLAB shape: 50 graphs 5.3 vs 8.2 us per
  // aggregation, 256 graphs 13.5 vs 12.9 us but 138 vs 171 us per step, 2048 graphs 69 vs 25 us and 706 vs 570 us).
  // (2) Cost model: K_g SIMD-cycles per node row (K_g = n_g rounded up to 64) against ~28 per edge of the gather.
  if (N < DG_DENSE_MIN_NODES) return false;
  int64_t kest = 2 * ((int64_t)N / B) + 64;       // ~ size-weighted mean graph size of a spread distribution
  if (kest > max_nodes + 63) kest = max_nodes + 63;
  kest = (kest / 64) * 64;
  if (kest < 64) kest = 64;
  return (int64_t)N * kest <= (int64_t)DG_DENSE_EDGE_COST * ((int64_t)E + N);
}
// Graph-chain forward (gcn_chain.hip): conv1..conv4 of every graph inside one workgroup, one launch instead of four, hs never
// leaves the CU.  Needs the bitmap + graph schedule of the dense structures (so: the coalesced + undirected promise and
// a node bound <= 512) and a raw feature width that admits the aggregate-first conv1; independent of the form the
// BACKWARD takes (dense per-layer kernels for large batches, CSR gather for small ones: dg_use_dense).
static bool dg_use_chain(int N, int E, int B, int F, int flags, int max_nodes) {
  if (flags & (DGCNN_FLAG_NO_CHAIN | DGCNN_FLAG_AGG_SPARSE | DGCNN_FLAG_FORCE_FUSED | DGCNN_FLAG_FORCE_TILED))
    return false;      // (each of these names another kernel family)
  if ((flags & DGCNN_FLAG_BF16) && max_nodes > dg_chain_small_rows(B)) return false;      // (no bf16 form of the size-class kernel)
  if (F > DG_AF_MAX_F) return false;
  if (!(flags & DGCNN_FLAG_COALESCED_UNDIRECTED) || E <= 0) return false;
  if (max_nodes <= 0 || max_nodes > DGD_MAXN) return false;
  if (dgd_num_items(N, B) > 100000) return false;
  if (flags & DGCNN_FLAG_CHAIN) return true;
  // small batches (one workgroup per graph): a graph above 256 nodes would run two tiles per wave and set the launch's
  // duration; large batches: graphs above the persistent kernel's size class take a second launch, worth it there
  if (max_nodes <= 256 || N >= DG_DENSE_MIN_NODES) return true;
  // forward-only use of a small batch with a graph of 257..512 nodes (DGCNN_FLAG_INFERENCE, round 6): the one-launch evaluation
  // kernel takes it (two tiles per wave) -- there is no backward whose gather kernels would want the CSR route instead
  return (flags & DGCNN_FLAG_INFERENCE) && B <= dg_chain_train_max_b() && B <= dg_readout_tail_max_b() &&
         max_nodes <= dg_chain_eval_max_nodes();
}
struct DgForm { bool dense, chain, bitmap, plan; int edge_check; };
#ifndef DG_INSYM_MIN_B
#define DG_INSYM_MIN_B 96
#endif
// one-launch evaluation / inference kernel (k_chain_readout_eval): every graph in the chain form, one workgroup per graph
static int g_eval_kernel = 1;      // dgcnn_eval_kernel_enable (tests / measurement A-B): 0 keeps chain forward + readout as two launches
int dgcnn_eval_kernel_enable(int on) { const int prev = g_eval_kernel; g_eval_kernel = on <= 0 ? 0 : (on >= 2 ? 2 : 1); return prev; }
// (a pure function of the batch's numbers: what graph PREPARATION may rely on -- the process-wide switch is read by the forward only)
static bool dg_one_launch_shape(const DgForm& f, int B, int max_nodes) {
  return f.chain && !f.dense && B <= dg_chain_train_max_b() && B <= dg_readout_tail_max_b() && max_nodes > 0 &&
         max_nodes <= dg_chain_train_max_nodes();
}
// Evaluation / inference forward in one launch.  Graphs of <= 256 nodes: whenever the batch has the one-launch shape.  A batch with
// a graph of 257..512 nodes (round 6: the kernel's two-tiles-per-wave form; test() on PROTEINS-like sets stays one launch): when
// the caller names the forward-only use -- DGCNN_FLAG_INFERENCE, which is also what makes dg_use_chain build the bitmap for such a
// small batch -- or the switch is at 2.  The switch only chooses between two routes over the SAME prepared structures
// (f.chain => the bitmap exists): it may change between preparation and forward.
static bool dg_eval_kernel_admits(const DgForm& f, int B, int max_nodes, int flags) {
  if (g_eval_kernel == 0) return false;
  if (dg_one_launch_shape(f, B, max_nodes)) return true;
  return (g_eval_kernel >= 2 || (flags & DGCNN_FLAG_INFERENCE)) && f.chain && !f.dense && B <= dg_chain_train_max_b() &&
         B <= dg_readout_tail_max_b() && max_nodes > dg_chain_train_max_nodes() && max_nodes <= dg_chain_eval_max_nodes();
}
static DgForm dg_form(int N, int E, int B, int F, int flags, int max_nodes) {
  DgForm f;
  f.chain = dg_use_chain(N, E, B, F, flags, max_nodes);
  // the bf16 leg changes the FORWARD's arithmetic only (hs in bf16, X.W on the bf16 matrix cores): with the chain forward
  // taking it, the backward -- fp32 in either case -- follows the fp32 rule; without, the leg exists in the dense per-layer
  // form only (dg_use_dense says yes whenever that form is admissible)
  f.dense = dg_use_dense(N, E, B, (f.chain ? flags & ~DGCNN_FLAG_BF16 : flags), max_nodes);
  f.bitmap = f.dense || f.chain;
  f.plan = f.dense || (f.chain && dg_chain_needs_schedule(B));      // item table + graph schedule (one workgroup of phase B)
  // reverse-edge check (the coalesced + undirected promise):
  //   0  on the finished bitmap by a launch of its own (k_prep_sym: dense forms, every row of every graph available there);
  //   1  per edge in phase B (binary search in the target's row): chain forward over a gather backward -- no third launch;
  //   2  by the one-launch training / evaluation kernel itself on its LDS image of the graph's bitmap, from DG_INSYM_MIN_B graphs
  //      on: there phase B is a rider of k_wgrad (it no longer fits beside the graph workgroups), and its per-edge searches
  //      were that launch's duration (256 COLLAB graphs: 13.6 us of 17.8); below, phase B runs on idle CUs for free.
  //      Decided from the batch's numbers and flags ALONE: preparation and forward of a batch are separate calls, and a
  //      process-wide switch (dgcnn_eval_kernel_enable) may change between them; a forward that then takes the two-launch chain
  //      route for such a batch runs the bitmap check as a launch of its own (dg_model_forward_impl).  DGCNN_FLAG_FORCE_FUSED /
  //      _FORCE_TILED / _AGG_SPARSE / _NO_CHAIN never get here: dg_use_chain says no, and without a bitmap phase B searches every edge.
  f.edge_check = (f.bitmap && !f.dense) ? ((B >= DG_INSYM_MIN_B && dg_one_launch_shape(f, B, max_nodes)) ? 2 : 1) : 0;
  return f;
}
// Sparse many-node batches on the launch-per-layer gather route (the narrow kernels' regime, gcn.hip: DD at the reference's
// batch of 50): the readout backward leaves the SortPooling-gradient slabs gp1..gp3 SPARSE -- rows of the <= 30 selected nodes of
// a graph + one flag word per node in the h4s region (dead once conv4's forward has run) -- instead of writing 3 x 128 B of zeros
// for every other node (a 661-node graph's workgroup spent 6.4 k of its 40 k cycles storing zeros, the 5748-node stress graph's
// 50 k), and the three consumers (k_gcn_bwd1, k_gcn_bwd32 / k_gcn_bwd32n) select on the flag.  A pure function of the numbers the
// forward and the backward call of a batch share, so producer and consumers always agree.
static bool dg_sparse_gp_gather(int N, int E, int B, int F, int flags, int max_nodes) {
  if (flags & DGCNN_FLAG_FORCE_FUSED) return false;
  const DgForm f = dg_form(N, E, B, F, flags, max_nodes);
  return !f.dense && !f.chain && dg_narrow_applies(N, E) != 0;
}
// the backward of a batch takes the form its forward took (same flags and max_nodes; the fused graph-per-workgroup
// forward never builds the bitmap)
static bool dg_backward_dense(int N, int E, int B, int flags, int max_nodes) {
  if (flags & DGCNN_FLAG_FORCE_FUSED) return false;
  return dg_use_dense(N, E, B, flags, max_nodes);
}
static DgDense dg_dense_view(const void* ws, const DgWs& wl, int N, int B) {
  DgDense G;
  G.graph_ptr = dg_cptr<int32_t>(ws, wl.graph_ptr); G.dmap = dg_cptr<int32_t>(ws, wl.dmap);
  G.bits = dg_cptr<uint32_t>(ws, wl.adjbits); G.N = N; G.B = B; G.NW = dgd_num_items(N, B);
  return G;
}

// DGCNN_STEP_KERNEL=0 in the environment: the one-launch training kernel stops after conv4's backward and conv3 / conv2 / conv1
// run as the two gather launches (the round-3 form; measurement A/B and a second route for the tests)
static int g_step_kernel = -1;
static bool dg_step_kernel_enabled() {
  if (g_step_kernel < 0) { const char* e = getenv("DGCNN_STEP_KERNEL"); g_step_kernel = (e && e[0] == '0') ? 0 : 1; }
  return g_step_kernel != 0;
}
int dgcnn_step_kernel_enable(int on) {
  const int prev = dg_step_kernel_enabled() ? 1 : 0;
  g_step_kernel = on ? 1 : 0;
  return prev;
}
int dgcnn_narrow_gather_enable(int on) { return dg_narrow_gather_enable(on); }
int dgcnn_forward_form(int N, int E, int B, int F, int flags, int max_nodes) {
  if (N <= 0 || B <= 0 || E < 0 || F < 1 || F > DGCNN_MAX_F) return DGCNN_EINVAL;
  const DgForm f = dg_form(N, E, B, F, flags, max_nodes);
  const bool chain_tail = f.chain && !f.dense && B <= dg_chain_train_max_b() && B <= dg_readout_tail_max_b() &&
                          max_nodes <= dg_chain_train_max_nodes();
  const bool step = chain_tail && dg_step_kernel_enabled() && B <= dg_grid1(N) && B <= dg_grid32(N) && dg_wgrad_takes_rider(B);
  // evaluation / inference (no labels-with-backward): chain forward + readout in one launch under the same admissibility
  const bool eval1 = dg_eval_kernel_admits(f, B, max_nodes, flags);
  return (f.dense ? DGCNN_FORM_DENSE : 0) | (f.chain ? DGCNN_FORM_CHAIN : 0) | (chain_tail ? DGCNN_FORM_CHAIN_TAIL : 0) |
         (step ? DGCNN_FORM_STEP : 0) | (eval1 ? DGCNN_FORM_EVAL : 0);
}

int dgcnn_model_prepare(int N, int E, int B, int F, int C, const float* x, const int64_t* edge_index,
                        const int64_t* batch, void* ws, int flags, int max_nodes, uint32_t epoch, dgcnn_stream_t stream) {
  if (!batch || !ws || N <= 0 || B <= 0 || E < 0 || epoch == 0) return DGCNN_EINVAL;
  if (E > 0 && !edge_index) return DGCNN_EINVAL;
  if (F <= DG_AF_MAX_F && !x) return DGCNN_EINVAL;       // the pre-scaled features are part of the preparation
  DgWs wl;
  DG_TRY(dg_ws_layout(N, E, B, F, C, &wl));
  DgLinFirst lf; lf.x = x; lf.W = nullptr; lf.hs = dg_ptr<float>(ws, wl.hsA); lf.F = F;
  const DgForm fm = dg_form(N, E, B, F, flags, max_nodes);
  const bool dense = fm.bitmap;
  const bool sym = dg_csr_symmetric(flags, E);      // (then the backward reads the CSR by target: no second copy is written)
  return dg_launch_prep(edge_index, E, batch, N, B, dg_ptr<int32_t>(ws, wl.rowptr), dg_ptr<int32_t>(ws, wl.colidx),
                        sym ? nullptr : dg_ptr<int32_t>(ws, wl.rowptr_t), sym ? nullptr : dg_ptr<int32_t>(ws, wl.colidx_t),
                        dg_ptr<float>(ws, wl.dinv),
                        dg_ptr<int32_t>(ws, wl.graph_ptr), dg_ptr<int32_t>(ws, wl.graph_eptr),
                        dg_ptr<int32_t>(ws, wl.cnt_in), dg_ptr<int32_t>(ws, wl.cnt_out), dg_ptr<int32_t>(ws, wl.err),
                        flags, epoch, (hipStream_t)stream, F <= DG_AF_MAX_F ? &lf : nullptr, nullptr,
                        dense ? dg_ptr<uint32_t>(ws, wl.adjbits) : nullptr, fm.plan ? dg_ptr<int32_t>(ws, wl.dmap) : nullptr,
                        fm.edge_check, max_nodes);
}

// ---- batches from a PREPARED dataset (dg_assemble.h) -------------------------------------------------------------
// descriptor of one batch's assembly into its workspace: which structures the batch's kernels will read follows from the
// SAME form selection the forward / backward use (dg_form) -- bitmap rows only for the dense / chain forms, a CSR only where a
// gather kernel runs, the item table / graph schedule (*dmap) only where a persistent kernel deals itself graphs
static int dg_fill_assemble(const dgcnn_dataset* ds, const int64_t* ids, const int32_t* onode, const int32_t* oedge, int N, int E,
                            int B, int C, void* ws, float* x, int64_t* batch, int64_t* y, int flags, int max_nodes, uint32_t epoch,
                            DgAssemble* A, int32_t** dmap) {
  if (!ds || !ids || !onode || !oedge || !ws || !x || !A || !dmap || N <= 0 || B <= 0 || E < 0 || epoch == 0) return DGCNN_EINVAL;
  if (!ds->node_ptr || !ds->x || !ds->rowptr || !ds->dinv || ds->G <= 0 || ds->Ntot <= 0 || ds->F < 1 || ds->F > DGCNN_MAX_F ||
      (ds->Etot > 0 && !ds->colidx))
    return DGCNN_EINVAL;
  if (!(flags & DGCNN_FLAG_COALESCED_UNDIRECTED)) return DGCNN_EUNSUPPORTED;      // (prepared datasets are verified undirected)
  const int F = ds->F;
  if (F <= DG_AF_MAX_F && !ds->xs) return DGCNN_EINVAL;
  DgWs wl;
  DG_TRY(dg_ws_layout(N, E, B, F, C, &wl));
  const DgForm fm = dg_form(N, E, B, F, flags, max_nodes);
  if (fm.bitmap && !ds->adj_bits) return DGCNN_EINVAL;
  A->node_ptr = ds->node_ptr; A->ds_rowptr = ds->rowptr; A->ds_colidx = ds->colidx; A->ds_dinv = ds->dinv;
  A->ds_xs = F <= DG_AF_MAX_F ? ds->xs : nullptr; A->ds_x = ds->x; A->ds_bits = ds->adj_bits; A->ds_y = ds->y;
  A->G = ds->G; A->Ntot = ds->Ntot;
  A->ids = ids; A->onode = onode; A->oedge = oedge; A->N = N; A->E = E; A->B = B; A->F = F;
  // a CSR wherever a gather kernel can run for this batch.  The dense form's own forward and backward read the bitmap only,
  // BUT (1) above the aggregate-first width conv1's own backward is the gather kernel in BOTH forms (dg_model_backward_impl:
  // dW1 = gh^T x over the transposed CSR = the CSR under the symmetric promise), and (2) a forced graph-per-workgroup forward
  // (k_fused_fwd / k_fused_fwd_d) stages its CSR slice whatever dg_form says, with the gather backward behind it
  const bool csr = !fm.dense || F > DG_AF_MAX_F || (flags & DGCNN_FLAG_FORCE_FUSED) != 0;
  A->rowptr = csr ? dg_ptr<int32_t>(ws, wl.rowptr) : nullptr;
  A->colidx = csr && E > 0 ? dg_ptr<int32_t>(ws, wl.colidx) : nullptr;
  A->dinv = dg_ptr<float>(ws, wl.dinv); A->xs = dg_ptr<float>(ws, wl.hsA); A->x = x; A->batch = batch; A->y = ds->y ? y : nullptr;
  A->graph_ptr = dg_ptr<int32_t>(ws, wl.graph_ptr); A->graph_eptr = dg_ptr<int32_t>(ws, wl.graph_eptr);
  A->bits = fm.bitmap ? dg_ptr<uint32_t>(ws, wl.adjbits) : nullptr;
  A->err = dg_ptr<unsigned int>(ws, wl.err); A->epoch = epoch;
  *dmap = fm.plan ? dg_ptr<int32_t>(ws, wl.dmap) : nullptr;
  return DGCNN_OK;
}
static int dg_assemble_args(const dgcnn_step_args* a, int flags, uint32_t epoch, hipStream_t s) {
  DgAssemble A; int32_t* dmap = nullptr;
  DG_TRY(dg_fill_assemble(a->ds, a->ds_ids, a->ds_onode, a->ds_oedge, a->N, a->E, a->B, a->C, a->ws, const_cast<float*>(a->x),
                          const_cast<int64_t*>(a->batch), const_cast<int64_t*>(a->y), flags, a->max_nodes, epoch, &A, &dmap));
  return dg_launch_assemble(&A, dmap, s);
}

int dgcnn_assemble(const dgcnn_dataset* ds, int B, int N, int E, int C, const int64_t* ids, const int32_t* onode,
                   const int32_t* oedge, void* ws, float* x, int64_t* batch, int64_t* y, int flags, int max_nodes,
                   uint32_t epoch, dgcnn_stream_t stream) {
  DgAssemble A; int32_t* dmap = nullptr;
  DG_TRY(dg_fill_assemble(ds, ids, onode, oedge, N, E, B, C, ws, x, batch, y, flags, max_nodes, epoch, &A, &dmap));
  return dg_launch_assemble(&A, dmap, (hipStream_t)stream);
}

int dgcnn_dataset_prepare(const dgcnn_dataset* ds, const int64_t* edge_index_global, const int64_t* batch_all,
                          int32_t* scratch, int32_t* err4, int flags, dgcnn_stream_t stream) {
  if (!ds || !batch_all || !scratch || !err4 || !ds->node_ptr || !ds->x || !ds->rowptr || !ds->dinv || ds->G <= 0 || ds->Ntot <= 0 ||
      ds->Etot < 0 || ds->F < 1 || ds->F > DGCNN_MAX_F)
    return DGCNN_EINVAL;
  if (ds->Ntot > 0x3fffffffLL || ds->Etot > 0x7fffffffLL || ds->G > 0x7ffffff0LL) return DGCNN_EUNSUPPORTED;      // int32 indices inside
  if (ds->Etot > 0 && (!edge_index_global || !ds->colidx)) return DGCNN_EINVAL;
  if (!(flags & DGCNN_FLAG_COALESCED_UNDIRECTED) || ds->Etot <= 0) return DGCNN_EUNSUPPORTED;
  if (ds->F <= DG_AF_MAX_F && !ds->xs) return DGCNN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(err4, 0, 4 * sizeof(int32_t), s) != hipSuccess) return DGCNN_ELAUNCH;
  const int N = (int)ds->Ntot, E = (int)ds->Etot, G = (int)ds->G;
  DgLinFirst lf; lf.x = ds->x; lf.W = nullptr; lf.hs = ds->xs; lf.F = ds->F;
  // the batch-level preparation over the whole dataset as ONE block-diagonal batch: same kernels, hence the same bits per graph.
  // edge_check = 1: every edge's reverse is looked up in its row (graphs above 512 nodes have no bitmap whose symmetry could
  // stand in for it); max_nodes = -1: such graphs are not an error here, they simply get no bitmap rows
  return dg_launch_prep(edge_index_global, E, batch_all, N, G, ds->rowptr, ds->colidx, nullptr, nullptr, ds->dinv, scratch,
                        scratch + (G + 1), nullptr, nullptr, err4, flags, 1u, s, ds->F <= DG_AF_MAX_F ? &lf : nullptr, nullptr,
                        ds->adj_bits, nullptr, 1, -1);
}

// rider_a != null: append phase A of another batch's graph preparation to the readout launch (tiled path only;
// *rode = 1 when it was attached).  tt != null (training step with labels): the readout forward and the readout
// backward run as ONE launch when the batch allows it; *tail_done = 1 then tells the backward to skip its first launch.
struct DgTrainTail { const int64_t* y; float loss_scale; };
// evaluation with labels: the batch's loss sum / #correct go to the device accumulator `metrics` (train.py:63-64); done = 1 when
// the forward's own launch folded them in (k_chain_readout_eval; needs the pipeline's counter `ctr` + its host mirror), else the
// caller launches k_eval_metrics behind it
struct DgEvalTail { const int64_t* y; float* metrics; float loss_scale; int done; unsigned int* ctr; unsigned int* ctr_host; };
// Pipelined large-batch step: the point of the step's launch sequence at which the side stream's graph preparation of the NEXT
// batch is forked (an event recorded on the caller's stream right behind launch `at`; dgcnn_pipeline_train_step sets and
// clears the request around its forward / backward calls).  Points: 1 chain forward, 2 readout forward, 3 classifier,
// 4 readout backward, 5 GCN backward a, 6 GCN backward b.
struct DgForkRequest { hipEvent_t ev; int at; bool done; };
static thread_local DgForkRequest g_fork{nullptr, 0, false};
static inline int dg_fork_point(int k, hipStream_t s) {
  if (g_fork.ev && !g_fork.done && g_fork.at == k) {
    if (hipEventRecord(g_fork.ev, s) != hipSuccess) return DGCNN_ELAUNCH;
    g_fork.done = true;
  }
  return DGCNN_OK;
}
static int dg_model_forward_impl(int N, int E, int B, int F, int C, const float* params,
                                 const float* x, const int64_t* edge_index, const int64_t* batch,
                                 void* ws, float* logp, int training, uint64_t seed, int flags, int max_nodes,
                                 int max_edges, uint32_t epoch, dgcnn_stream_t stream, const DgPrepRider* rider_a,
                                 int* rode, const DgTrainTail* tt = nullptr, int* tail_done = nullptr, DgEvalTail* et = nullptr) {
  if (!params || !x || !ws || !logp || N <= 0 || B <= 0 || E < 0 || epoch == 0) return DGCNN_EINVAL;
  if (!(flags & DGCNN_FLAG_PREPARED) && (!batch || (E > 0 && !edge_index))) return DGCNN_EINVAL;      // (a prepared batch's structures are in ws)
  DgParams pl; DgWs wl;
  DG_TRY(dg_param_layout(F, C, &pl));
  DG_TRY(dg_ws_layout(N, E, B, F, C, &wl));
  hipStream_t s = (hipStream_t)stream;
  int32_t* rowptr = dg_ptr<int32_t>(ws, wl.rowptr);
  int32_t* colidx = dg_ptr<int32_t>(ws, wl.colidx);
  float* dinv = dg_ptr<float>(ws, wl.dinv);
  float* hsA = dg_ptr<float>(ws, wl.hsA);
  float* hsB = dg_ptr<float>(ws, wl.hsB);
  float* h4s = dg_ptr<float>(ws, wl.h4s);
  float *x1 = dg_ptr<float>(ws, wl.x1), *x2 = dg_ptr<float>(ws, wl.x2), *x3 = dg_ptr<float>(ws, wl.x3),
        *x4 = dg_ptr<float>(ws, wl.x4);

  // Path choice.  One workgroup per graph only pays when there are enough graphs to occupy the chip
  // (256 CUs): at the reference's batch of 50 the tiled kernels spread each graph's nodes over all CUs
  // and are faster; from a few hundred graphs per batch the LDS-resident kernel wins (no L2 gathers).
  const bool want_fused = (flags & DGCNN_FLAG_FORCE_FUSED) ||
                          (!(flags & DGCNN_FLAG_FORCE_TILED) && B >= DGCNN_FUSED_MIN_GRAPHS);
  const bool fused = want_fused && !(flags & DGCNN_FLAG_AGG_DENSE) && max_nodes > 0 && max_edges > 0 &&
                     dg_fused_fits(max_nodes, max_edges, F);
  // graph-per-workgroup forward in the dense block form: small batches of small graphs (the reference's batch of 50),
  // where four per-layer launches + the readout launch cost ~5 us of dispatch and cold-read latency EACH
  const bool fd_ok = !fused && !(flags & (DGCNN_FLAG_FORCE_TILED | DGCNN_FLAG_BF16)) && F <= DG_AF_MAX_F && max_nodes > 0 &&
                     max_nodes <= dg_fused_d_max_nodes();
  const bool fused_d = fd_ok && (((flags & DGCNN_FLAG_FORCE_FUSED) && (flags & DGCNN_FLAG_AGG_DENSE)) ||
                                 (!(flags & (DGCNN_FLAG_AGG_SPARSE | DGCNN_FLAG_AGG_DENSE)) && B <= DG_FUSED_D_MAX_B));
  const DgForm fm = dg_form(N, E, B, F, flags, max_nodes);
  const bool dense = !fused && !fused_d && fm.dense;
  const bool chain = !fused && !fused_d && fm.chain;
  const bool bitmap = dense || chain;
  const int bf16 = (flags & DGCNN_FLAG_BF16) ? 1 : 0;
  if (bf16 && !dense && !chain) return DGCNN_EUNSUPPORTED;          // the bf16 leg runs in the chain / dense block forms only
  const DgDense G = dg_dense_view(ws, wl, N, B);
  const bool af = F <= DG_AF_MAX_F;    // conv1 aggregate-first: prep leaves xs = dinv*x in hsA, no linear at all
  DgLinFirst lf; lf.x = x; lf.W = af ? nullptr : params + pl.off[0]; lf.hs = hsA; lf.F = F;
  const bool use_lf = af;              // aggregate-first conv1: graph prep also leaves xs = dinv*x
  int lin_done = 0;
  // graph structure, once per batch (the reference recomputes the normalisation in all 4 layers)
  if (!(flags & DGCNN_FLAG_PREPARED))
  DG_TRY(dg_launch_prep(edge_index, E, batch, N, B, rowptr, colidx,
                        dg_csr_symmetric(flags, E) ? nullptr : dg_ptr<int32_t>(ws, wl.rowptr_t),
                        dg_csr_symmetric(flags, E) ? nullptr : dg_ptr<int32_t>(ws, wl.colidx_t), dinv, dg_ptr<int32_t>(ws, wl.graph_ptr),
                        dg_ptr<int32_t>(ws, wl.graph_eptr), dg_ptr<int32_t>(ws, wl.cnt_in), dg_ptr<int32_t>(ws, wl.cnt_out),
                        dg_ptr<int32_t>(ws, wl.err), flags, epoch, s, use_lf ? &lf : nullptr, &lin_done,
                        bitmap ? dg_ptr<uint32_t>(ws, wl.adjbits) : nullptr,
                        (bitmap && fm.plan) ? dg_ptr<int32_t>(ws, wl.dmap) : nullptr, fm.edge_check, max_nodes));
  if (fused_d) {
    DG_TRY(dg_launch_fused_fwd_d(N, B, F, C, params, &pl, x, rowptr, colidx, dinv, dg_ptr<int32_t>(ws, wl.graph_ptr),
                                 dg_ptr<float>(ws, wl.ax), x1, x2, x3, x4, dg_ptr<float>(ws, wl.pooled),
                                 dg_ptr<int32_t>(ws, wl.perm), dg_ptr<float>(ws, wl.a5), dg_ptr<float>(ws, wl.a6),
                                 dg_ptr<float>(ws, wl.a1d), dg_ptr<uint8_t>(ws, wl.drop_mask), logp, training, seed,
                                 dg_ptr<int32_t>(ws, wl.err), epoch, s, rider_a,
                                 g_prof_which >= 0 ? g_prof_a : nullptr, g_prof_which >= 0 ? g_prof_b : nullptr));
    g_prof_which = -1;
    if (rider_a && rode) *rode = 1;
    return DGCNN_OK;
  }
  if (fused) {
    // graph-per-workgroup path: conv1..conv4 + SortPooling + tail in ONE launch, activations in LDS
    const int nmax = ((max_nodes + 15) / 16) * 16;
    DG_TRY(dg_launch_fused_fwd(N, B, F, C, nmax, max_edges > 0 ? max_edges : 0, params, &pl, x, rowptr, colidx, dinv,
                               dg_ptr<int32_t>(ws, wl.graph_ptr), dg_ptr<int32_t>(ws, wl.graph_eptr), dg_ptr<float>(ws, wl.ax), x1, x2, x3, x4,
                               dg_ptr<float>(ws, wl.pooled),
                               dg_ptr<int32_t>(ws, wl.perm), dg_ptr<float>(ws, wl.a5), dg_ptr<float>(ws, wl.a6),
                               dg_ptr<float>(ws, wl.a1d), dg_ptr<uint8_t>(ws, wl.drop_mask), logp, training, seed,
                               dg_ptr<int32_t>(ws, wl.err), epoch, s,
                               g_prof_which >= 0 ? g_prof_a : nullptr, g_prof_which >= 0 ? g_prof_b : nullptr));
    g_prof_which = -1;
    return DGCNN_OK;
  }
  // conv1 linear (raw features), then 4 aggregation launches; each one also produces the next
  // layer's pre-scaled linear output on MFMA, so X.W never takes a launch of its own after this.
  if (chain && tt && tail_done && !dense && B <= dg_chain_train_max_b() && B <= dg_readout_tail_max_b() &&
      max_nodes <= dg_chain_train_max_nodes()) {
    // small training batch: chain forward + readout forward + readout backward + the whole GCN backward of every graph in ONE
    // launch (one partial row per graph for k_wgrad, the step's only other launch)
    const bool step_kernel = dg_step_kernel_enabled() && B <= wl.P1 && B <= wl.P32 && dg_wgrad_takes_rider(B);
    int fused_b = 0;      // (phase B of the rider joined this launch: *rode = 2, nothing is left to ride on k_wgrad)
    DG_TRY(dg_launch_chain_readout_tail(N, B, F, C, dg_ptr<int32_t>(ws, wl.graph_ptr), G.bits, dinv, hsA, params, &pl,
                                        dg_ptr<float>(ws, wl.ax), x1, x2, x3, x4, dg_ptr<float>(ws, wl.pooled),
                                        dg_ptr<int32_t>(ws, wl.perm), dg_ptr<float>(ws, wl.a5), dg_ptr<float>(ws, wl.a6),
                                        dg_ptr<float>(ws, wl.a1d), dg_ptr<uint8_t>(ws, wl.drop_mask), logp, training, seed, tt->y,
                                        tt->loss_scale, dg_ptr<float>(ws, wl.dlogit), dg_ptr<float>(ws, wl.gz1),
                                        dg_ptr<float>(ws, wl.gz6), dg_ptr<float>(ws, wl.gz5), dg_ptr<float>(ws, wl.gp1),
                                        dg_ptr<float>(ws, wl.gp2), dg_ptr<float>(ws, wl.gp3), dg_ptr<float>(ws, wl.gas4),
                                        dg_ptr<float>(ws, wl.gb4p), dg_ptr<float>(ws, wl.lossv), dg_ptr<float>(ws, wl.ptail),
                                        dg_ptr<int32_t>(ws, wl.err), epoch, dg_ptr<float>(ws, wl.gasA), dg_ptr<float>(ws, wl.pa4),
                                        wl.P1, s, rider_a, g_prof_which >= 0 ? g_prof_a : nullptr, g_prof_which >= 0 ? g_prof_b : nullptr,
                                        step_kernel ? dg_ptr<float>(ws, wl.pb3) : nullptr, step_kernel ? dg_ptr<float>(ws, wl.pb2) : nullptr,
                                        step_kernel ? dg_ptr<float>(ws, wl.pb1) : nullptr, bf16,
                                        (rider_a && rode && (flags & DGCNN_FLAG_EXCLUSIVE_DEVICE)) ? &fused_b : nullptr,
                                        fm.edge_check == 2 ? 1 : 0));
    g_prof_which = -1;
    // 2: conv4's backward (gas3 in gasA, {dW4, db3} partials) rode along too; 3: the whole GCN backward did (row b of pa4 / pb3 /
    // pb2 / pb1 = graph b's partials: k_wgrad sums B rows)
    *tail_done = step_kernel ? 3 : (B <= wl.P1 ? 2 : 1);
    if (rider_a && rode) *rode = fused_b > 0 ? 2 : 1;
    return DGCNN_OK;
  }
  if (chain && !(tt && tail_done) && dg_eval_kernel_admits(fm, B, max_nodes, flags)) {
    // evaluation / inference (and the drop-in route's forward: everything the backward reads is saved as by the two launches this
    // replaces): chain forward + readout forward of every graph in ONE launch; with labels, the batch's metrics too
    int fused_b = 0;
    const bool in_launch = et && et->y && et->metrics && et->ctr && et->ctr_host;      // metrics folded in by the launch's last workgroup
    DG_TRY(dg_launch_chain_readout_eval(N, B, F, C, dg_ptr<int32_t>(ws, wl.graph_ptr), G.bits, dinv, hsA, params, &pl,
                                        dg_ptr<float>(ws, wl.ax), x1, x2, x3, x4, dg_ptr<float>(ws, wl.pooled),
                                        dg_ptr<int32_t>(ws, wl.perm), dg_ptr<float>(ws, wl.a5), dg_ptr<float>(ws, wl.a6),
                                        dg_ptr<float>(ws, wl.a1d), dg_ptr<uint8_t>(ws, wl.drop_mask), logp, training, seed,
                                        in_launch ? et->y : nullptr, in_launch ? et->loss_scale : 0.f, dg_ptr<float>(ws, wl.lossv),
                                        in_launch ? et->ctr : nullptr, in_launch ? et->ctr_host : nullptr,
                                        in_launch ? et->metrics : nullptr, dg_ptr<int32_t>(ws, wl.err), epoch, s, rider_a,
                                        g_prof_which >= 0 ? g_prof_a : nullptr, g_prof_which >= 0 ? g_prof_b : nullptr, bf16,
                                        (rider_a && rode && (flags & DGCNN_FLAG_EXCLUSIVE_DEVICE)) ? &fused_b : nullptr,
                                        fm.edge_check == 2 ? 1 : 0, max_nodes));
    g_prof_which = -1;
    if (in_launch) et->done = 1;
    if (rider_a && rode) *rode = fused_b > 0 ? 2 : 1;
    return DGCNN_OK;
  }
  if (chain) {
    // the preparation of this batch left the reverse-edge check to a one-launch kernel (edge_check 2) and this forward is not one:
    // the one-launch evaluation kernel is switched off (dgcnn_eval_kernel_enable(0)).  Check the bitmap here, as the dense forms do.
    if (fm.edge_check == 2)
      DG_TRY(dg_launch_prep_sym(edge_index, E, N, B, batch, dg_ptr<int32_t>(ws, wl.graph_ptr), G.bits, dg_ptr<int32_t>(ws, wl.err),
                                epoch, s));
    DG_TRY(dg_launch_chain_fwd(N, B, F, max_nodes, dg_ptr<int32_t>(ws, wl.graph_ptr), G.bits, dinv, hsA, params, &pl,
                               dg_ptr<float>(ws, wl.ax), x1, x2, x3, x4, fm.plan ? dg_ptr<int32_t>(ws, wl.dmap) : nullptr, bf16, s,
                               g_prof_which >= 0 ? g_prof_a : nullptr,
                               g_prof_which >= 0 ? g_prof_b : nullptr));
    g_prof_which = -1;
  } else if (dense) {
    // dense per-graph block form (gcn_dense.hip): same four launches, A.H on the matrix cores from the bit-packed adjacency
    if (af) {
      DG_TRY(dg_launch_gcn_fwd_af_d(bf16, &G, F, dinv, hsA, params + pl.off[0], params + pl.off[1],
                                    dg_ptr<float>(ws, wl.ax), x1, params + pl.off[2], hsB, s, DG_PROF_A(0), DG_PROF_B(0)));
    } else {
      if (!lin_done) DG_TRY(dg_launch_lin_first(N, F, x, params + pl.off[0], dinv, hsA, 32, s));
      DG_TRY(dg_launch_gcn_fwd32d(0, 0, bf16, &G, dinv, hsA, params + pl.off[1], x1, params + pl.off[2], hsB, s,
                                  DG_PROF_A(0), DG_PROF_B(0)));
    }
    DG_TRY(dg_launch_gcn_fwd32d(0, bf16, bf16, &G, dinv, hsB, params + pl.off[3], x2, params + pl.off[4], hsA, s,
                                DG_PROF_A(1), DG_PROF_B(1)));
    DG_TRY(dg_launch_gcn_fwd32d(1, bf16, 0, &G, dinv, hsA, params + pl.off[5], x3, params + pl.off[6], h4s, s,
                                DG_PROF_A(2), DG_PROF_B(2)));
    g_prof_which = -1;
    DG_TRY(dg_launch_gcn_fwd1d(&G, dinv, h4s, params + pl.off[7], x4, s));
  } else {
  if (af) {
    DG_TRY(dg_launch_gcn_fwd_af(N, F, rowptr, colidx, dinv, hsA, params + pl.off[0], params + pl.off[1],
                                dg_ptr<float>(ws, wl.ax), x1, params + pl.off[2], hsB, s, DG_PROF_A(0), DG_PROF_B(0)));
  } else {
    if (!lin_done) DG_TRY(dg_launch_lin_first(N, F, x, params + pl.off[0], dinv, hsA, 32, s));
    DG_TRY(dg_launch_gcn_fwd32(0, N, rowptr, colidx, dinv, hsA, params + pl.off[1], x1, params + pl.off[2], hsB, s,
                               DG_PROF_A(0), DG_PROF_B(0), E));
  }
  DG_TRY(dg_launch_gcn_fwd32(0, N, rowptr, colidx, dinv, hsB, params + pl.off[3], x2, params + pl.off[4], hsA, s,
                             DG_PROF_A(1), DG_PROF_B(1), E));
  DG_TRY(dg_launch_gcn_fwd32(1, N, rowptr, colidx, dinv, hsA, params + pl.off[5], x3, params + pl.off[6], h4s, s,
                             DG_PROF_A(2), DG_PROF_B(2), E));
  g_prof_which = -1;
  DG_TRY(dg_launch_gcn_fwd1(N, rowptr, colidx, dinv, h4s, params + pl.off[7], x4, s, E));
  }
  if (tt && tail_done && !dense && B <= dg_readout_tail_max_b()) {
    // training step: readout forward + backward in one launch (operands of the backward stay on the CU that made them)
    DG_TRY(dg_launch_readout_tail(N, B, C, params, &pl, dg_ptr<int32_t>(ws, wl.graph_ptr), x1, x2, x3, x4,
                                  dg_ptr<float>(ws, wl.pooled), dg_ptr<int32_t>(ws, wl.perm), dg_ptr<float>(ws, wl.a5),
                                  dg_ptr<float>(ws, wl.a6), dg_ptr<float>(ws, wl.a1d), dg_ptr<uint8_t>(ws, wl.drop_mask), logp,
                                  training, seed, dinv, tt->y, tt->loss_scale, dg_ptr<float>(ws, wl.dlogit),
                                  dg_ptr<float>(ws, wl.gz1), dg_ptr<float>(ws, wl.gz6), dg_ptr<float>(ws, wl.gz5),
                                  dg_ptr<float>(ws, wl.gp1), dg_ptr<float>(ws, wl.gp2), dg_ptr<float>(ws, wl.gp3),
                                  dg_ptr<float>(ws, wl.gas4), dg_ptr<float>(ws, wl.gb4p), dg_ptr<float>(ws, wl.lossv),
                                  dg_ptr<float>(ws, wl.ptail), s, rider_a,
                                  dg_sparse_gp_gather(N, E, B, F, flags, max_nodes) ? dg_ptr<int32_t>(ws, wl.h4s) : nullptr));
    *tail_done = 1;
    if (rider_a && rode) *rode = 1;
    return DGCNN_OK;
  }
  // SortPooling + the whole dense tail: one launch, one workgroup per graph -- up to conv6's output when the batch is large
  // enough for classifier_1 / classifier_2 to run as GEMMs over graphs (classifier.hip: with their backward when this is a
  // training step with labels; *tail_done = 4 then tells the backward that gz6 already holds conv6's output gradient)
  const bool batched_head = dg_classifier_batched(B);
  DG_TRY(dg_fork_point(1, s));
  DG_TRY(dg_launch_readout_fwd(N, B, C, params, &pl, dg_ptr<int32_t>(ws, wl.graph_ptr), x1, x2, x3, x4,
                               dg_ptr<float>(ws, wl.pooled), dg_ptr<int32_t>(ws, wl.perm), dg_ptr<float>(ws, wl.a5),
                               dg_ptr<float>(ws, wl.a6), dg_ptr<float>(ws, wl.a1d),
                               dg_ptr<uint8_t>(ws, wl.drop_mask), logp, training, seed, s, rider_a, !batched_head));
  DG_TRY(dg_fork_point(2, s));
  if (batched_head) {
    const bool with_bwd = tt && tail_done;
    DG_TRY(dg_launch_classifier(B, C, params, &pl, dg_ptr<float>(ws, wl.a6), dg_ptr<float>(ws, wl.a1d),
                                dg_ptr<uint8_t>(ws, wl.drop_mask), logp, training, seed, with_bwd ? tt->y : nullptr,
                                with_bwd ? tt->loss_scale : 0.f, dg_ptr<float>(ws, wl.dlogit), dg_ptr<float>(ws, wl.gz1),
                                dg_ptr<float>(ws, wl.gz6), dg_ptr<float>(ws, wl.lossv), dg_ptr<float>(ws, wl.ptail), s));
    if (with_bwd) *tail_done = 4;
  }
  DG_TRY(dg_fork_point(3, s));
  if (rider_a && rode) *rode = 1;
  return DGCNN_OK;
}

int dgcnn_model_forward(int N, int E, int B, int F, int C, const float* params,
                        const float* x, const int64_t* edge_index, const int64_t* batch,
                        void* ws, float* logp, int training, uint64_t seed, int flags, int max_nodes,
                        int max_edges, uint32_t epoch, dgcnn_stream_t stream) {
  return dg_model_forward_impl(N, E, B, F, C, params, x, edge_index, batch, ws, logp, training, seed, flags, max_nodes,
                               max_edges, epoch, stream, nullptr, nullptr);
}

// form of a batch's backward: `dense` = the dense per-layer kernels (else CSR gather); `chain` = conv4 + conv3 as one
// graph-chain launch (gcn_chain.hip: needs the bitmap the forward's preparation built and graphs of <= 256 nodes);
// `plan` = the item table / graph schedule exists (batches above one graph per persistent workgroup)
struct DgBwdForm { bool dense, chain, plan, sym, sparse_gp; };
static DgBwdForm dg_backward_form(int N, int E, int B, int F, int flags, int max_nodes) {
  DgBwdForm b{false, false, false, dg_csr_symmetric(flags, E), false};
  b.sparse_gp = dg_sparse_gp_gather(N, E, B, F, flags, max_nodes);
  if (flags & DGCNN_FLAG_FORCE_FUSED) return b;            // (the fused graph-per-workgroup forward never builds the bitmap)
  const DgForm f = dg_form(N, E, B, F, flags, max_nodes);
  b.dense = f.dense; b.plan = f.plan;
  // the backward chain pays where the batch fills the chip (the dense form's regime: 2048 COLLAB graphs 62 -> 35 us for the two
  // layers); at the reference's batch of 50 the largest graph's critical path makes it no faster than the two gather
  // launches it replaces (10.6 vs 10.2 us) -- DGCNN_FLAG_CHAIN asks for it anyway (tests)
  b.chain = f.bitmap && (f.dense || (flags & DGCNN_FLAG_CHAIN)) && !(flags & DGCNN_FLAG_NO_CHAIN) &&
            max_nodes > 0 && max_nodes <= dg_chain_bwd_max_nodes() && (f.plan || !dg_chain_needs_schedule(B));
  return b;
}
static int dg_model_backward_impl(int N, int E, int B, int F, int C, const float* params, const float* x,
                                  void* ws, const float* logp, const float* glogp, const int64_t* y,
                                  float loss_scale, int training, float* grads, float* metrics,
                                  const DgAdam* adam, hipStream_t s, const DgBwdForm& bf, const DgPrepRider* rider_b = nullptr,
                                  int tail_done = 0) {
  const bool dense = bf.dense;
  const bool head_done = (tail_done & 4) != 0;      // the batched classifier ran its backward inside the forward half of the step
  tail_done &= 3;
  DgParams pl; DgWs wl;
  DG_TRY(dg_param_layout(F, C, &pl));
  DG_TRY(dg_ws_layout(N, E, B, F, C, &wl));
  const int32_t* rowptr_t = dg_cptr<int32_t>(ws, bf.sym ? wl.rowptr : wl.rowptr_t);
  const int32_t* colidx_t = dg_cptr<int32_t>(ws, bf.sym ? wl.colidx : wl.colidx_t);
  const float* dinv = dg_cptr<float>(ws, wl.dinv);
  float *gasA = dg_ptr<float>(ws, wl.gasA), *gasB = dg_ptr<float>(ws, wl.gasB), *gas4 = dg_ptr<float>(ws, wl.gas4);
  float *gp1 = dg_ptr<float>(ws, wl.gp1), *gp2 = dg_ptr<float>(ws, wl.gp2), *gp3 = dg_ptr<float>(ws, wl.gp3);
  const float *x1 = dg_cptr<float>(ws, wl.x1), *x2 = dg_cptr<float>(ws, wl.x2), *x3 = dg_cptr<float>(ws, wl.x3),
              *x4 = dg_cptr<float>(ws, wl.x4);

  // readout forward + backward ran as one launch (which carried phase A of the next batch's preparation): phase B rides
  // on the LAST launch of the step, the weight-gradient kernel, when that is the single-launch form
  // large batches whose three GCN backward layers are the two chain launches: the SortPooling-gradient slabs gp1..gp3 stay
  // SPARSE -- rows of the selected nodes + a flag word per node (in the h4s region, which nothing of a chain step uses)
  // instead of 3 x 128 B of zeros for every other node (2048 COLLAB graphs: 57 MB less written by k_tail_bwd, 35 MB less read)
  int32_t* gpsel = ((head_done && dense && bf.chain && F <= DG_AF_MAX_F) || (bf.sparse_gp && !dense && !bf.chain && tail_done < 2))
                       ? dg_ptr<int32_t>(ws, wl.h4s) : nullptr;
  const bool wg_rider = tail_done && rider_b && !dense && dg_wgrad_takes_rider(B);
  // large batches (two-stage weight gradients) behind the batched classifier, nothing riding: the walking form of the readout
  // backward -- one conv5 / conv6 partial row per workgroup of DG_TAIL_WALK graphs
  const bool walk = head_done && !rider_b && B > dg_wg_two_stage_b();
  if (walk)
    DG_TRY(dg_launch_tail_bwd_walk(N, B, C, params, &pl, dg_cptr<int32_t>(ws, wl.graph_ptr), dg_cptr<int32_t>(ws, wl.perm), dinv, x4,
                                   dg_cptr<float>(ws, wl.a5), dg_cptr<float>(ws, wl.a6), dg_ptr<float>(ws, wl.gz6),
                                   dg_ptr<float>(ws, wl.gz5), gp1, gp2, gp3, gas4, dg_ptr<float>(ws, wl.gb4p),
                                   dg_ptr<float>(ws, wl.ptail), dg_cptr<float>(ws, wl.pooled), s, gpsel));
  else if (!tail_done)
  DG_TRY(dg_launch_tail_bwd(N, B, C, params, &pl, dg_cptr<int32_t>(ws, wl.graph_ptr), dg_cptr<int32_t>(ws, wl.perm),
                            dinv, x4, dg_cptr<float>(ws, wl.a5), dg_cptr<float>(ws, wl.a6),
                            dg_cptr<float>(ws, wl.a1d), logp, glogp, y, loss_scale, training,
                            dg_ptr<float>(ws, wl.dlogit), dg_ptr<float>(ws, wl.gz1), dg_ptr<float>(ws, wl.gz6),
                            dg_ptr<float>(ws, wl.gz5), gp1, gp2, gp3, gas4, dg_ptr<float>(ws, wl.gb4p),
                            dg_ptr<float>(ws, wl.lossv), dg_ptr<float>(ws, wl.ptail),
                            dg_cptr<float>(ws, wl.pooled), s, rider_b, !head_done, gpsel));
  DG_TRY(dg_fork_point(4, s));
  if (dense) {
    // dense block form (the forward of this batch took it: the bitmap is in the workspace); F > 32 keeps the gather
    // kernel for conv1's own backward (its operand is the raw [N,F] input)
    const DgDense G = dg_dense_view(ws, wl, N, B);
    if (bf.chain) {
      // conv4 + conv3 backward of every graph inside one workgroup: gas4 -> gas2 (gasB), partial {dW4, db3}, {dW3, db2}
      DG_TRY(dg_launch_chain_bwd_a(N, B, G.graph_ptr, G.bits, dinv, gas4, params + pl.off[6], params + pl.off[4], x3, gp3, x2, gp2,
                                   gasB, dg_ptr<float>(ws, wl.pa4), wl.P1, dg_ptr<float>(ws, wl.pb3), wl.P32,
                                   bf.plan ? dg_ptr<int32_t>(ws, wl.dmap) : nullptr, s, gpsel));
    } else {
    DG_TRY(dg_launch_gcn_bwd1d(&G, dinv, gas4, params + pl.off[6], x3, gp3, gasA, dg_ptr<float>(ws, wl.pa4), wl.P1, s));
    DG_TRY(dg_launch_gcn_bwd32d(&G, dinv, gasA, params + pl.off[4], x2, gp2, gasB, dg_ptr<float>(ws, wl.pb3), wl.P32, s));
    }
    DG_TRY(dg_fork_point(5, s));
    if (F <= DG_AF_MAX_F && bf.chain) {
      DG_TRY(dg_launch_chain_bwd_b(N, B, F, G.graph_ptr, G.bits, dinv, gasB, params + pl.off[2], x1, gp1, dg_cptr<float>(ws, wl.ax),
                                   dg_ptr<float>(ws, wl.pb2), dg_ptr<float>(ws, wl.pb1), wl.P32,
                                   bf.plan ? dg_ptr<int32_t>(ws, wl.dmap) : nullptr, s, gpsel));
    } else if (F <= DG_AF_MAX_F) {
      DG_TRY(dg_launch_gcn_bwd32d(&G, dinv, gasB, params + pl.off[2], x1, gp1, gasA, dg_ptr<float>(ws, wl.pb2), wl.P32, s,
                                  dg_cptr<float>(ws, wl.ax), F, dg_ptr<float>(ws, wl.pb1)));
    } else {
      DG_TRY(dg_launch_gcn_bwd32d(&G, dinv, gasB, params + pl.off[2], x1, gp1, gasA, dg_ptr<float>(ws, wl.pb2), wl.P32, s));
      DG_TRY(dg_launch_gcn_bwd32(1, N, F, rowptr_t, colidx_t, dinv, gasA, nullptr, x, nullptr, nullptr,
                                 dg_ptr<float>(ws, wl.pb1), wl.P32, s));
    }
  } else if (tail_done == 3) {
    // the one-launch training kernel ran the whole GCN backward: nothing to launch here
  } else {
  const bool bwd1_hosts_rider = tail_done && !wg_rider && rider_b;      // (then conv4's backward launch carries prep phase B)
  if (tail_done == 2 && !bwd1_hosts_rider) {
    // conv4's backward ran inside the one-launch training kernel: gas3 is in gasA, {dW4, db3} in pa4
    DG_TRY(dg_launch_gcn_bwd32(0, N, 32, rowptr_t, colidx_t, dinv, gasA, params + pl.off[4], x2, gp2, gasB,
                               dg_ptr<float>(ws, wl.pb3), wl.P32, s, nullptr, 0, nullptr, E));
  } else if (bf.chain && !bwd1_hosts_rider) {
    const DgDense G = dg_dense_view(ws, wl, N, B);
    DG_TRY(dg_launch_chain_bwd_a(N, B, G.graph_ptr, G.bits, dinv, gas4, params + pl.off[6], params + pl.off[4], x3, gp3, x2, gp2,
                                 gasB, dg_ptr<float>(ws, wl.pa4), wl.P1, dg_ptr<float>(ws, wl.pb3), wl.P32,
                                 bf.plan ? dg_ptr<int32_t>(ws, wl.dmap) : nullptr, s));
  } else {
  // conv4 backward (+ start of conv3's): gas4 -> gas3 (in gasA), partial {dW4, db3}
  DG_TRY(dg_launch_gcn_bwd1(N, rowptr_t, colidx_t, dinv, gas4, params + pl.off[6], x3, gp3, gasA,
                            dg_ptr<float>(ws, wl.pa4), wl.P1, s, (tail_done && !wg_rider) ? rider_b : nullptr, gpsel, E));
  // conv3 backward: gas3 (gasA) -> gas2 (gasB), partial {dW3, db2}
  DG_TRY(dg_launch_gcn_bwd32(0, N, 32, rowptr_t, colidx_t, dinv, gasA, params + pl.off[4], x2, gp2, gasB,
                             dg_ptr<float>(ws, wl.pb3), wl.P32, s, nullptr, 0, nullptr, E, gpsel));
  }
  // conv2 backward: gas2 (gasB) -> gas1 (gasA), partial {dW2, db1}
  if (F <= DG_AF_MAX_F) {
    // ... carrying conv1's whole backward: dW1 = ga1^T . (A_hat x), from the ax slab the forward saved
    DG_TRY(dg_launch_gcn_bwd32(0, N, 32, rowptr_t, colidx_t, dinv, gasB, params + pl.off[2], x1, gp1, gasA,
                               dg_ptr<float>(ws, wl.pb2), wl.P32, s, dg_cptr<float>(ws, wl.ax), F,
                               dg_ptr<float>(ws, wl.pb1), E, gpsel));
  } else {
    DG_TRY(dg_launch_gcn_bwd32(0, N, 32, rowptr_t, colidx_t, dinv, gasB, params + pl.off[2], x1, gp1, gasA,
                               dg_ptr<float>(ws, wl.pb2), wl.P32, s, nullptr, 0, nullptr, E, gpsel));
    // conv1 backward: gas1 (gasA) -> partial dW1 (data.x needs no gradient)
    DG_TRY(dg_launch_gcn_bwd32(1, N, F, rowptr_t, colidx_t, dinv, gasA, nullptr, x, nullptr, nullptr,
                               dg_ptr<float>(ws, wl.pb1), wl.P32, s, nullptr, 0, nullptr, E));
  }
  }
  DG_TRY(dg_fork_point(6, s));
  // every weight gradient (tail + GCN partial reductions) in ONE launch, fixed-order reductions, optional Adam.
  // (Running the tail half on a second stream concurrently with the GCN chain was measured SLOWER: its
  // ~2400 workgroups starve the latency-bound 1024-thread GCN workgroups of CU slots: 111 -> 137 us/step.)
  DG_TRY(dg_launch_wgrad(3, N, B, F, C, &pl, &wl, ws, grads, (y != nullptr) ? metrics : nullptr, adam, s,
                         wg_rider ? rider_b : nullptr, walk ? dg_tail_walk_rows(B) : 0, (tail_done == 3 && !dense) ? B : 0));
  return DGCNN_OK;
}

int dgcnn_model_backward(int N, int E, int B, int F, int C, const float* params,
                         const float* x, void* ws, const float* logp,
                         const float* glogp, const int64_t* y, float loss_scale, int training,
                         float* grads, float* metrics, int flags, int max_nodes, dgcnn_stream_t stream) {
  if (!params || !x || !ws || !logp || !grads || N <= 0 || B <= 0) return DGCNN_EINVAL;
  if ((glogp == nullptr) == (y == nullptr)) return DGCNN_EINVAL;
  return dg_model_backward_impl(N, E, B, F, C, params, x, ws, logp, glogp, y, loss_scale, training ? 1 : 0,
                                grads, metrics, nullptr, (hipStream_t)stream, dg_backward_form(N, E, B, F, flags, max_nodes));
}

int dgcnn_model_backward_step(int N, int E, int B, int F, int C, float* params, const float* x, void* ws,
                              const float* logp, const int64_t* y, float loss_scale, int training, float* grads,
                              float* metrics, float* exp_avg, float* exp_avg_sq, int64_t step, float lr,
                              float beta1, float beta2, float eps, int flags, int max_nodes, dgcnn_stream_t stream) {
  if (!params || !x || !ws || !logp || !y || !grads || !exp_avg || !exp_avg_sq || N <= 0 || B <= 0 || step < 1)
    return DGCNN_EINVAL;
  DgAdam ad;
  ad.params = params; ad.exp_avg = exp_avg; ad.exp_avg_sq = exp_avg_sq;
  ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps; ad.step = step;
  return dg_model_backward_impl(N, E, B, F, C, params, x, ws, logp, nullptr, y, loss_scale, training ? 1 : 0,
                                grads, metrics, &ad, (hipStream_t)stream, dg_backward_form(N, E, B, F, flags, max_nodes));
}

// ---- pipelined training step ------------------------------------------------------------------------
// The next batch's graph preparation rides on this step's two graph-per-workgroup launches (k_readout_fwd carries
// phase A, k_tail_bwd phase B; see dg_prep.h): same stream, no events, no second queue.  (A side stream + events
// was measured SLOWER than no overlap at all -- 97 vs 86 us/step: each cross-queue dependency costs ~5 us here.)
struct DgPipeline {
  const void* prep_ws = nullptr;         // workspace holding a prepared-but-not-yet-consumed graph structure
  int pN = 0, pE = 0, pB = 0, pflags = 0, pmaxn = 0;
  uint32_t pepoch = 0;
  // large batches: the next batch's graph preparation runs as launches of its own on a SIDE stream, forked from the caller's
  // stream at the start of the step and joined at its end (created at the first such step, destroyed with the pipeline)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // small batches: BOTH phases of the next batch's preparation ride on the one-launch training kernel; the phase-B workgroups wait on
  // this device counter for the phase-A workgroups of the same launch (dg_prep.h).  Created at the first rider, freed with the pipeline.
  unsigned int* sync_ctr = nullptr;
  unsigned int sync_count = 0;
  // evaluation steps in the one-launch form: counter of finished graph workgroups (the last one of a launch folds the batch's
  // metrics into the accumulator); a line of its own in the same allocation, monotonic, `ev_count` = its value after every launch
  // issued so far
  unsigned int* ev_ctr = nullptr;
  unsigned int ev_count = 0;
};
// the pipeline's device counters (one 128-byte allocation, zeroed once on `s`): false = not available (callers keep the forms
// that need none)
static bool dg_pipeline_counters(DgPipeline* h, hipStream_t s) {
  if (h->sync_ctr) return true;
  unsigned int* c = nullptr;
  if (hipMalloc(&c, 128) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMemsetAsync(c, 0, 128, s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(c); return false; }
  h->sync_ctr = c; h->sync_count = 0; h->ev_ctr = c + 16; h->ev_count = 0;
  return true;
}
// From this many graphs per step on the next batch's preparation leaves the rider slots of the step's launches for the side
// stream.  Why: at 2048 COLLAB graphs every launch of the step fills the chip, so rider blocks are ADDED time (k_readout_fwd
// 34 -> 59 us with phase A behind its graph workgroups, k_tail_bwd 35 -> 53 us with phase B), while most launches of the step
// are latency chains that leave the memory system idle (k_chain_fwd_q moves 66 MB in 43 us): the preparation's 79 us of
// launches (phase A is an HBM stream of the int64 edge list) overlap with the step's ~230 us instead.  At the reference's
// batch of 50 the riders fill CUs the graph workgroups leave empty and cost nothing: kept there.
// Measured crossover (COLLAB graphs per step, side stream vs riders, us): 384: 118.9 vs 117.3, 512: 124.4 vs 124.4, 768: 149.1 vs
// 156.0, 1024: 172.2 vs 184.5, 2048: 272.1 vs 286.8.
#ifndef DG_SIDE_PREP_MIN_B
#define DG_SIDE_PREP_MIN_B 512
#endif
#ifndef DG_SIDE_FORK_AT
#define DG_SIDE_FORK_AT 4           // fork behind the readout backward (dg_fork_point)
#endif

int dgcnn_pipeline_create(void** handle) {
  if (!handle) return DGCNN_EINVAL;
  *handle = new DgPipeline();
  return DGCNN_OK;
}

int dgcnn_pipeline_destroy(void* handle) {
  if (!handle) return DGCNN_EINVAL;
  DgPipeline* h = static_cast<DgPipeline*>(handle);
  if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->sync_ctr) (void)hipFree(h->sync_ctr);
  delete h;
  return DGCNN_OK;
}

// the next batch's preparation as rider workgroups of this step's launches (below the side-stream regime): its assembly from a
// prepared dataset (ONE phase), or phases A and B of the graph preparation from its coalesced undirected edge list.
// *rider = rd when there is something to ride, else nullptr (general edge lists: prepared in-stream behind the step).
static int dg_pipeline_rider(DgPipeline* h, const dgcnn_step_args* next, bool side_prep, hipStream_t s, DgPrepRider& rd,
                             const DgPrepRider** rider_out) {
  const DgPrepRider* rider = nullptr;
  if (!side_prep && next && next->ds) {
    // the next batch comes from a prepared dataset: its whole assembly (a copy with offset adds, dg_assemble.h) rides where
    // phase A of a per-batch preparation rides; there is no phase B.  (A batch that needs the planning workgroup -- a forced
    // dense form below the side-stream regime -- is assembled in-stream after the step instead.)
    int32_t* ndmap = nullptr;
    DG_TRY(dg_fill_assemble(next->ds, next->ds_ids, next->ds_onode, next->ds_oedge, next->N, next->E, next->B, next->C, next->ws,
                            const_cast<float*>(next->x), const_cast<int64_t*>(next->batch), const_cast<int64_t*>(next->y),
                            next->flags, next->max_nodes, next->epoch, &rd.as, &ndmap));
    if (!ndmap) {
      rd.mode = 1;
      rd.nblk = dg_cdiv(dg_assemble_work(next->N, next->E, next->B, rd.as.colidx != nullptr), 1024);
      rd.nblk_b = 0;
      rider = &rd;
    }
  } else if (!side_prep && next && (next->flags & DGCNN_FLAG_COALESCED_UNDIRECTED) && next->E > 0) {
    DgWs nl;
    DG_TRY(dg_ws_layout(next->N, next->E, next->B, next->F, next->C, &nl));
    rd.ei = next->edge_index; rd.batch = next->batch; rd.E = next->E; rd.N = next->N; rd.B = next->B;
    rd.rowptr = dg_ptr<int32_t>(next->ws, nl.rowptr); rd.colidx = dg_ptr<int32_t>(next->ws, nl.colidx);
    rd.rowptr_t = nullptr; rd.colidx_t = nullptr;      // (riders exist for undirected edge lists only: dg_csr_symmetric)
    rd.graph_ptr = dg_ptr<int32_t>(next->ws, nl.graph_ptr); rd.graph_eptr = dg_ptr<int32_t>(next->ws, nl.graph_eptr);
    rd.dinv = dg_ptr<float>(next->ws, nl.dinv); rd.err = dg_ptr<unsigned int>(next->ws, nl.err);
    const bool naf = next->F <= DG_AF_MAX_F;
    rd.x = naf ? next->x : nullptr; rd.xs = naf ? dg_ptr<float>(next->ws, nl.hsA) : nullptr; rd.F = next->F;
    rd.epoch = next->epoch;
    const DgForm nf = dg_form(next->N, next->E, next->B, next->F, next->flags, next->max_nodes);
    if (nf.bitmap) {
      rd.bits = dg_ptr<unsigned int>(next->ws, nl.adjbits); rd.dmap = nf.plan ? dg_ptr<int>(next->ws, nl.dmap) : nullptr;
      rd.edge_check = nf.edge_check;
      rd.max_nodes = next->max_nodes;
    }
    rd.nblk = dg_cdiv(dg_prep_fast_work(next->E, next->N, next->B, rd.bits != nullptr), 1024);
    rd.nblk_b = dg_cdiv(dg_prep_fast_work_b(next->E, next->N, next->B, rd.bits != nullptr, rd.edge_check == 1), 1024);
    (void)dg_pipeline_counters(h, s);      // (once per pipeline; a failure only means that phase B keeps riding on k_wgrad)
    rd.sync_ctr = h->sync_ctr; rd.sync_host = h->sync_ctr ? &h->sync_count : nullptr;
    rider = &rd;
  }
  *rider_out = rider;
  return DGCNN_OK;
}

// flags that take part in dg_form / the forward's choice of kernel family: a batch prepared under one set is not "prepared" for a
// step that names another
static inline bool dg_form_flags_differ(int a, int b) {
  const int m = DGCNN_FLAG_COALESCED_UNDIRECTED | DGCNN_FLAG_FORCE_FUSED | DGCNN_FLAG_FORCE_TILED | DGCNN_FLAG_AGG_SPARSE |
                DGCNN_FLAG_AGG_DENSE | DGCNN_FLAG_CHAIN | DGCNN_FLAG_NO_CHAIN | DGCNN_FLAG_BF16 | DGCNN_FLAG_INFERENCE;
  return ((a ^ b) & m) != 0;
}
int dgcnn_pipeline_train_step(void* handle, const dgcnn_step_args* cur, const dgcnn_step_args* next,
                              dgcnn_stream_t stream) {
  if (!handle || !cur) return DGCNN_EINVAL;
  DgPipeline* h = static_cast<DgPipeline*>(handle);
  hipStream_t s = (hipStream_t)stream;
  if (!cur->params || !cur->x || (!cur->batch && !cur->ds) || !cur->y || !cur->ws || !cur->logp || !cur->grads || cur->N <= 0 ||
      cur->B <= 0 || cur->E < 0 || cur->epoch == 0)
    return DGCNN_EINVAL;
  if (cur->exp_avg && (!cur->exp_avg_sq || cur->step < 1)) return DGCNN_EINVAL;
  if (next && (next->ws == cur->ws || !next->ws || (!next->batch && !next->ds) || !next->x || next->N <= 0 || next->B <= 0 ||
               next->E < 0 || next->epoch == 0 || (next->E > 0 && !next->edge_index && !next->ds)))
    return DGCNN_EINVAL;
  // (max_nodes takes part in the choice of the aggregation form, which decides what the preparation built)
  const bool match = h->prep_ws == cur->ws && h->pN == cur->N && h->pE == cur->E && h->pB == cur->B &&
                     h->pmaxn == cur->max_nodes;
  bool prepared = (cur->flags & DGCNN_FLAG_PREPARED) != 0;           // the host says so explicitly ...
  if (prepared && !match) return DGCNN_EINVAL;                       // ... and it must be the batch we prepared
  // ... FOR the kernel family this step runs: the preparation's form (bitmap or not, who checks the reverse edges)
  // followed from the flags it was given; a step that names another family now prepares again under its own flags
  if (prepared && dg_form_flags_differ(cur->flags, h->pflags)) prepared = false;
  int flags = cur->flags & ~DGCNN_FLAG_PREPARED;
  uint32_t epoch = cur->epoch;
  if (prepared) {
    flags = h->pflags | DGCNN_FLAG_PREPARED | (cur->flags & DGCNN_FLAG_EXCLUSIVE_DEVICE);
    epoch = h->pepoch;        // the error words of this workspace carry the preparation's tag
  }
  // a look-ahead preparation survives a step that neither consumes it, nor writes its workspace, nor prepares another batch
  // (an evaluation step between two training steps, say): the one after that still finds it
  if (h->prep_ws == cur->ws || next) h->prep_ws = nullptr;

  static const bool no_side = dg_knob("DG_NO_SIDE_PREP");      // A/B switch (DG_DEBUG_KNOBS builds only)
  // (a batch of a PREPARED dataset is never assembled on the side stream: the copy takes 16 us alone at 2048 graphs, but beside the
  //  GCN backward it stretched k_chain_bwd_b from 31 to 52 us and itself to 50 -- in-stream behind the step: 250.7 -> see DESIGN)
  bool side_prep = next && cur->B >= DG_SIDE_PREP_MIN_B && !no_side && !next->ds;
  if (side_prep && !h->side) {
    int prio_least = 0, prio_greatest = 0;      // the LOWEST priority: the step's own launches take the CUs first
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // all three or nothing: a stream without its events would make every later large step fail in hipEventRecord
    hipStream_t st = nullptr; hipEvent_t e1 = nullptr, e2 = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_least) != hipSuccess ||
        hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      if (e1) (void)hipEventDestroy(e1);
      if (e2) (void)hipEventDestroy(e2);
      if (st) (void)hipStreamDestroy(st);
      side_prep = false;                                         // (no side stream to be had: riders / in-stream as before)
    } else { h->side = st; h->ev_fork = e1; h->ev_join = e2; }
  }
  DgPrepRider rd{};
  const DgPrepRider* rider = nullptr;
  if (cur->ds && !prepared) {        // first batch of a loop drawn from a prepared dataset: assemble in-stream, then run as prepared
    DG_TRY(dg_assemble_args(cur, flags, epoch, s));
    flags |= DGCNN_FLAG_PREPARED;
  }
  DG_TRY(dg_pipeline_rider(h, next, side_prep, s, rd, &rider));
  int rode = 0, tail_done = 0;
  DgTrainTail tt;
  tt.y = cur->y; tt.loss_scale = cur->loss_scale;
  g_fork = DgForkRequest{nullptr, 0, false};
  // whatever way this call returns (every DG_TRY below may), the thread-local fork request is disarmed: a later plain
  // dgcnn_model_forward on this thread must not record into an event of a pipeline that may be gone by then
  struct ForkGuard { ~ForkGuard() { g_fork = DgForkRequest{nullptr, 0, false}; } } fork_guard;
  if (side_prep) {
    int at = DG_SIDE_FORK_AT;
#ifdef DG_DEBUG_KNOBS
    if (const char* e = getenv("DG_FORK_AT")) at = atoi(e);
#endif
    g_fork = DgForkRequest{h->ev_fork, at, false};
    if (at == 0) { if (hipEventRecord(h->ev_fork, s) != hipSuccess) return DGCNN_ELAUNCH; g_fork.done = true; }
  }
  DG_TRY(dg_model_forward_impl(cur->N, cur->E, cur->B, cur->F, cur->C, cur->params, cur->x, cur->edge_index, cur->batch,
                               cur->ws, cur->logp, cur->training, cur->seed, flags, cur->max_nodes, cur->max_edges,
                               epoch, stream, rider, &rode, &tt, &tail_done));
  DgAdam ad;
  const DgAdam* adam = nullptr;
  if (cur->exp_avg) {
    ad.params = cur->params; ad.exp_avg = cur->exp_avg; ad.exp_avg_sq = cur->exp_avg_sq;
    ad.lr = cur->lr; ad.beta1 = cur->beta1; ad.beta2 = cur->beta2; ad.eps = cur->eps; ad.step = cur->step;
    adam = &ad;
  }
  DG_TRY(dg_model_backward_impl(cur->N, cur->E, cur->B, cur->F, cur->C, cur->params, cur->x, cur->ws, cur->logp, nullptr,
                                cur->y, cur->loss_scale, cur->training ? 1 : 0, cur->grads, cur->metrics, adam, s,
                                dg_backward_form(cur->N, cur->E, cur->B, cur->F, flags, cur->max_nodes),
                                (rode == 1 && rd.mode == 0) ? rider : nullptr, tail_done));      // (rode == 2: phase B already ran)
  if (next && rode && rd.mode == 0 && rd.bits && !rd.edge_check)      // dense next batch: its reverse-edge check on the bitmap the riders just built
    DG_TRY(dg_launch_prep_sym(next->edge_index, next->E, next->N, next->B, next->batch, rd.graph_ptr, rd.bits,
                              reinterpret_cast<int32_t*>(rd.err), rd.epoch, s));
  if (next) {
    // no rider possible (general edge list, or this step took the graph-per-workgroup forward): prepare in-stream now
    if (side_prep) {
      if (!g_fork.done && hipEventRecord(h->ev_fork, s) != hipSuccess) return DGCNN_ELAUNCH;      // (a route without that launch)
      g_fork = DgForkRequest{nullptr, 0, false};
      // The preparation runs beside the GCN backward chain kernels -- LDS / matrix-core latency chains that move little memory
      // -- and the weight-gradient launches.  Fork point swept at 2048 COLLAB graphs (step, us; riders instead: 289.8):
      // start of the step 287.2, behind the chain forward 291.7, readout forward 298.8, classifier 313.2, READOUT BACKWARD
      // 277.7, GCN backward a 295.0, b 308.9 -- beside the readout kernels, whose graph workgroups are chains of dependent
      // loads, the preparation's HBM stream stretches them by more than it saves.  Join: whatever the caller enqueues next
      // sees the prepared structure.  (next->ws was last read by the previous step, which precedes the fork.)
      if (hipStreamWaitEvent(h->side, h->ev_fork, 0) != hipSuccess) return DGCNN_ELAUNCH;
      if (next->ds) DG_TRY(dg_assemble_args(next, next->flags, next->epoch, h->side));
      else
      DG_TRY(dgcnn_model_prepare(next->N, next->E, next->B, next->F, next->C, next->x, next->edge_index, next->batch,
                                 next->ws, next->flags, next->max_nodes, next->epoch, (dgcnn_stream_t)h->side));
      if (hipEventRecord(h->ev_join, h->side) != hipSuccess || hipStreamWaitEvent(s, h->ev_join, 0) != hipSuccess) return DGCNN_ELAUNCH;
    } else if (!rode) {
      if (next->ds) DG_TRY(dg_assemble_args(next, next->flags, next->epoch, s));
      else
      DG_TRY(dgcnn_model_prepare(next->N, next->E, next->B, next->F, next->C, next->x, next->edge_index, next->batch,
                                 next->ws, next->flags, next->max_nodes, next->epoch, stream));
    }
    h->prep_ws = next->ws; h->pN = next->N; h->pE = next->E; h->pB = next->B; h->pflags = next->flags;
    h->pepoch = next->epoch; h->pmaxn = next->max_nodes;
  }
  return DGCNN_OK;
}

int dgcnn_pipeline_eval_step(void* handle, const dgcnn_step_args* cur, const dgcnn_step_args* next, dgcnn_stream_t stream) {
  if (!handle || !cur) return DGCNN_EINVAL;
  DgPipeline* h = static_cast<DgPipeline*>(handle);
  hipStream_t s = (hipStream_t)stream;
  if (!cur->params || !cur->x || (!cur->batch && !cur->ds) || !cur->ws || !cur->logp || cur->N <= 0 || cur->B <= 0 || cur->E < 0 ||
      cur->epoch == 0)
    return DGCNN_EINVAL;
  if (next && (next->ws == cur->ws || !next->ws || (!next->batch && !next->ds) || !next->x || next->N <= 0 || next->B <= 0 ||
               next->E < 0 || next->epoch == 0 || (next->E > 0 && !next->edge_index && !next->ds)))
    return DGCNN_EINVAL;
  const bool match = h->prep_ws == cur->ws && h->pN == cur->N && h->pE == cur->E && h->pB == cur->B && h->pmaxn == cur->max_nodes;
  bool prepared = (cur->flags & DGCNN_FLAG_PREPARED) != 0;
  if (prepared && !match) return DGCNN_EINVAL;
  if (prepared && dg_form_flags_differ(cur->flags, h->pflags)) prepared = false;      // (see dgcnn_pipeline_train_step)
  int flags = cur->flags & 0xFFFF & ~DGCNN_FLAG_PREPARED;
  uint32_t epoch = cur->epoch;
  if (prepared) {
    flags = h->pflags | DGCNN_FLAG_PREPARED | (cur->flags & DGCNN_FLAG_EXCLUSIVE_DEVICE);
    epoch = h->pepoch;        // the error words of this workspace carry the preparation's tag
  }
  // a look-ahead preparation survives a step that neither consumes it, nor writes its workspace, nor prepares another batch
  // (an evaluation step between two training steps, say): the one after that still finds it
  if (h->prep_ws == cur->ws || next) h->prep_ws = nullptr;
  if (cur->ds && !prepared) {        // first batch of a loop drawn from a prepared dataset: assemble in-stream, then run as prepared
    DG_TRY(dg_assemble_args(cur, flags, epoch, s));
    flags |= DGCNN_FLAG_PREPARED;
  }
  // riders below the regime where every launch fills the chip (there the next batch is prepared in-stream behind the forward:
  // an evaluation loop has no backward under which a side stream's preparation could hide)
  DgPrepRider rd{};
  const DgPrepRider* rider = nullptr;
  DG_TRY(dg_pipeline_rider(h, next, cur->B >= DG_SIDE_PREP_MIN_B, s, rd, &rider));
  int rode = 0;
  const bool with_metrics = cur->y && cur->metrics;
  DgEvalTail et{cur->y, cur->metrics, cur->loss_scale, 0, nullptr, nullptr};
  if (with_metrics && dg_pipeline_counters(h, s)) { et.ctr = h->ev_ctr; et.ctr_host = &h->ev_count; }
  DG_TRY(dg_model_forward_impl(cur->N, cur->E, cur->B, cur->F, cur->C, cur->params, cur->x, cur->edge_index, cur->batch, cur->ws,
                               cur->logp, 0, 0, flags, cur->max_nodes, cur->max_edges, epoch, stream, rider, &rode, nullptr, nullptr,
                               with_metrics ? &et : nullptr));
  if (with_metrics && !et.done) DG_TRY(dg_launch_eval_metrics(cur->B, cur->C, cur->logp, cur->y, cur->metrics, cur->loss_scale, s));
  if (next) {
    if (rode && rd.mode == 0) {
      if (rode == 1) DG_TRY(dg_launch_prep_phase_b(&rd, s));          // (rode == 2: phase B ran inside the forward's launch)
      if (rd.bits && !rd.edge_check)      // dense next batch: its reverse-edge check on the bitmap just built
        DG_TRY(dg_launch_prep_sym(next->edge_index, next->E, next->N, next->B, next->batch, rd.graph_ptr, rd.bits,
                                  reinterpret_cast<int32_t*>(rd.err), rd.epoch, s));
    } else if (!rode) {
      if (next->ds) DG_TRY(dg_assemble_args(next, next->flags, next->epoch, s));
      else
      DG_TRY(dgcnn_model_prepare(next->N, next->E, next->B, next->F, next->C, next->x, next->edge_index, next->batch, next->ws,
                                 next->flags, next->max_nodes, next->epoch, stream));
    }
    h->prep_ws = next->ws; h->pN = next->N; h->pE = next->E; h->pB = next->B; h->pflags = next->flags;
    h->pepoch = next->epoch; h->pmaxn = next->max_nodes;
  }
  return DGCNN_OK;
}

int dgcnn_model_eval_step(const dgcnn_step_args* a, dgcnn_stream_t stream) {
  if (!a || !a->params || !a->x || (!a->batch && !a->ds) || !a->ws || !a->logp || a->N <= 0 || a->B <= 0 || a->E < 0 || a->epoch == 0)
    return DGCNN_EINVAL;
  int eflags = a->flags & 0xFFFF & ~DGCNN_FLAG_PREPARED;
  if (a->ds) {      // batch from a prepared dataset: assemble, then the forward finds its structures in the workspace
    DG_TRY(dg_assemble_args(a, eflags, a->epoch, (hipStream_t)stream));
    eflags |= DGCNN_FLAG_PREPARED;
  }
  DgEvalTail et{a->y, a->metrics, a->loss_scale, 0, nullptr, nullptr};      // (no pipeline, no counter: k_eval_metrics behind the forward)
  const bool with_metrics = a->y && a->metrics;
  DG_TRY(dg_model_forward_impl(a->N, a->E, a->B, a->F, a->C, a->params, a->x, a->edge_index, a->batch, a->ws, a->logp, 0, 0,
                               eflags, a->max_nodes, a->max_edges, a->epoch, stream, nullptr, nullptr, nullptr, nullptr,
                               with_metrics ? &et : nullptr));
  if (with_metrics && !et.done)
    DG_TRY(dg_launch_eval_metrics(a->B, a->C, a->logp, a->y, a->metrics, a->loss_scale, (hipStream_t)stream));
  return DGCNN_OK;
}

int dgcnn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                    float lr, float beta1, float beta2, float eps, int zero_grads, dgcnn_stream_t stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq) return DGCNN_EINVAL;
  return dg_launch_adam(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, zero_grads,
                        (hipStream_t)stream);
}

int dgcnn_collate(int B, int F, int64_t N, int64_t E, int64_t Etot, const int64_t* graph_ids, const int64_t* out_node_ptr,
                  const int64_t* out_edge_ptr, const float* x_all, const int64_t* ei_all, const int64_t* node_ptr,
                  const int64_t* edge_ptr, const int64_t* y_all, float* x, int64_t* edge_index, int64_t* batch,
                  int64_t* y, dgcnn_stream_t stream) {
  if (!graph_ids || !out_node_ptr || !out_edge_ptr || !x_all || !node_ptr || !edge_ptr || !y_all || !x || !batch || !y)
    return DGCNN_EINVAL;
  if (E > 0 && (!ei_all || !edge_index)) return DGCNN_EINVAL;
  return dg_launch_collate(B, F, N, E, Etot, graph_ids, out_node_ptr, out_edge_ptr, x_all, ei_all, node_ptr, edge_ptr,
                           y_all, x, edge_index, batch, y, (hipStream_t)stream);
}

int dgcnn_collate_ids(int B, int F, const int64_t* ids_host, const int64_t* ids_dev, const int64_t* nodes_per_graph_host,
                      const int64_t* edges_per_graph_host, int64_t num_graphs, int64_t* meta_host, int64_t* meta_dev,
                      void* ev_uploaded, int64_t Etot, const float* x_all, const int64_t* ei_all, const int64_t* node_ptr,
                      const int64_t* edge_ptr, const int64_t* y_all, int64_t cap_nodes, int64_t cap_edges, float* x,
                      int64_t* edge_index, int64_t* batch, int64_t* y, int64_t* out_sizes, dgcnn_stream_t stream) {
  if (B <= 0 || F < 1 || !ids_host || !nodes_per_graph_host || !edges_per_graph_host || !meta_host || !meta_dev ||
      !x_all || !node_ptr || !edge_ptr || !y_all || !y || !out_sizes)
    return DGCNN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t ev = (hipEvent_t)ev_uploaded;
  // ids already on the device (one upload per epoch) and a small batch: the kernel rebuilds the prefix sums itself,
  // nothing is uploaded per batch; otherwise the prefix sums go up through the pinned staging buffer
  const bool scan = ids_dev != nullptr && B <= 256;
  // the previous asynchronous upload from this staging buffer must have been consumed before it is rewritten
  if (!scan && ev && hipEventSynchronize(ev) != hipSuccess) return DGCNN_ELAUNCH;
  // (scan mode never touches meta_host: an earlier batch's asynchronous upload from this ring slot may still be
  // queued -- the guard above is skipped in scan mode -- and rewriting the staging buffer would corrupt it)
  int64_t nsum = 0, esum = 0, nmax = 0, emax = 0;
  if (!scan) { meta_host[0] = 0; meta_host[B + 1] = 0; }
  for (int k = 0; k < B; ++k) {
    const int64_t g = ids_host[k];
    if (g < 0 || g >= num_graphs) return DGCNN_EINVAL;
    const int64_t n = nodes_per_graph_host[g], e = edges_per_graph_host[g];
    nsum += n; esum += e;
    if (n > nmax) nmax = n;
    if (e > emax) emax = e;
    if (!scan) {
      meta_host[k + 1] = nsum;
      meta_host[B + 2 + k] = esum;
      meta_host[2 * B + 2 + k] = g;
    }
  }
  out_sizes[0] = nsum; out_sizes[1] = esum; out_sizes[2] = nmax; out_sizes[3] = emax;
  if (nsum > cap_nodes || esum > cap_edges) return DGCNN_EUNSUPPORTED;     // caller grows its buffers and calls again
  if (nsum <= 0 || !x || !batch) return DGCNN_EINVAL;
  if (esum > 0 && (!ei_all || !edge_index)) return DGCNN_EINVAL;
  if (scan)
    return dg_launch_collate_scan(B, F, nsum, esum, Etot, ids_dev, x_all, ei_all, node_ptr, edge_ptr, y_all, x, edge_index,
                                  batch, y, s);
  if (hipMemcpyAsync(meta_dev, meta_host, sizeof(int64_t) * (3 * (size_t)B + 2), hipMemcpyHostToDevice, s) != hipSuccess)
    return DGCNN_ELAUNCH;
  if (ev && hipEventRecord(ev, s) != hipSuccess) return DGCNN_ELAUNCH;
  return dg_launch_collate(B, F, nsum, esum, Etot, meta_dev + 2 * B + 2, meta_dev, meta_dev + B + 1, x_all, ei_all, node_ptr,
                           edge_ptr, y_all, x, edge_index, batch, y, s);
}

int dgcnn_accumulate_metrics(int B, const void* ws, int N, int E, int F, int C, float* metrics,
                             dgcnn_stream_t stream) {
  DgWs wl;
  if (!ws || !metrics) return DGCNN_EINVAL;
  DG_TRY(dg_ws_layout(N, E, B, F, C, &wl));
  return dg_launch_metrics(B, dg_cptr<float>(ws, wl.lossv), metrics, (hipStream_t)stream);
}

}  // extern "C"
