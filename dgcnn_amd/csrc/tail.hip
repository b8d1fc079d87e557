// tail.hip -- SortPooling readout and the dense tail, forward and backward (gfx950).
//
// One workgroup per graph (graphs are independent in every op below):
//   SortAggregation(k=30)                     /root/reference/model.py:17,35  [PyG]
//   Conv1d(1,16,97,97)+ReLU, MaxPool1d(2,2)   model.py:18,20,37-38
//   Conv1d(16,32,5,1)+ReLU, flatten 352       model.py:19,39-40
//   Linear(352,128)+ReLU, Dropout(.5)         model.py:21-22,41-42
//   Linear(128,C), log_softmax                model.py:23,43
// and their autograd (what `loss.backward()`, /root/reference/train.py:40, runs for these ops),
// plus nn.NLLLoss() (train.py:39,98) when labels are handed in directly.
//
// The per-graph sort runs entirely in LDS: keys are packed as (descending-ordered float bits of
// channel 96, node index) 64-bit words, so one ascending integer sort gives "key descending,
// ties by lower node index" -- this build's documented tie-break (the reference's is undefined).
//   n <= 256  : rank sort (each thread counts smaller keys; no barriers)
//   n <= 4096 : bitonic network in LDS
//   n  > 4096 : k rounds of workgroup-wide arg-min selection (keys stay in HBM/L2)
#include "dg_common.h"
#include <stdlib.h>
#include "dg_readout.h"
#include "dg_prep.h"
#ifndef DG_TAIL_BIG_MIN_B
#define DG_TAIL_BIG_MIN_B 257         // more graphs than CUs: the readout pair uses its two-workgroups-per-CU forms
#endif

__global__ void __launch_bounds__(SP_THREADS)
k_sortpool_fwd(const int* __restrict__ graph_ptr, const float* __restrict__ x1, const float* __restrict__ x2,
               const float* __restrict__ x3, const float* __restrict__ x4, float* __restrict__ pooled,
               int* __restrict__ perm) {
  __shared__ unsigned long long keys[SP_LDS_KEYS];
  __shared__ unsigned long long red[16];
  __shared__ int sel[DGCNN_K];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n0 = graph_ptr[b], n = graph_ptr[b + 1] - n0;
  dg_select_topk(x4, n0, n, keys, red, sel);
  if (tid < DGCNN_K) perm[b * DGCNN_K + tid] = sel[tid] >= 0 ? n0 + sel[tid] : -1;
  float* out = pooled + (size_t)b * KCAT;
  for (int o = tid; o < KCAT; o += SP_THREADS) {
    const int s = o / DGCNN_CAT, c = o - s * DGCNN_CAT;
    const int ln = sel[s];
    out[o] = ln >= 0 ? dg_cat_load(x1, x2, x3, x4, n0 + ln, c) : 0.f;   // zero padding (fill -> 0)
  }
}

int dg_launch_sortpool_fwd(int N, int B, const int32_t* graph_ptr, const float* x1, const float* x2,
                           const float* x3, const float* x4, float* pooled, int32_t* perm, hipStream_t s) {
  if (B <= 0 || N <= 0) return DGCNN_EINVAL;
  hipLaunchKernelGGL(k_sortpool_fwd, dim3(B), dim3(SP_THREADS), 0, s, graph_ptr, x1, x2, x3, x4, pooled, perm);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// stand-alone SortPooling backward: dense per-node gradient slabs (zero for unselected nodes)
__global__ void __launch_bounds__(SP_THREADS)
k_sortpool_bwd(const int* __restrict__ graph_ptr, const int* __restrict__ perm, const float* __restrict__ gpooled,
               float* __restrict__ g1, float* __restrict__ g2, float* __restrict__ g3, float* __restrict__ g4) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n0 = graph_ptr[b], n = graph_ptr[b + 1] - n0;
  for (int t = tid; t < n * 32; t += SP_THREADS) {
    g1[(size_t)n0 * 32 + t] = 0.f; g2[(size_t)n0 * 32 + t] = 0.f; g3[(size_t)n0 * 32 + t] = 0.f;
  }
  for (int t = tid; t < n; t += SP_THREADS) g4[n0 + t] = 0.f;
  __syncthreads();
  const int m = n < DGCNN_K ? n : DGCNN_K;
  for (int o = tid; o < m * DGCNN_CAT; o += SP_THREADS) {
    const int s = o / DGCNN_CAT, c = o - s * DGCNN_CAT;
    const int node = perm[b * DGCNN_K + s];
    const float v = gpooled[(size_t)b * KCAT + o];
    if (c < 32) g1[(size_t)node * 32 + c] = v;
    else if (c < 64) g2[(size_t)node * 32 + c - 32] = v;
    else if (c < 96) g3[(size_t)node * 32 + c - 64] = v;
    else g4[node] = v;
  }
}

int dg_launch_sortpool_bwd(int N, int B, const int32_t* graph_ptr, const int32_t* perm, const float* gpooled,
                           float* g1, float* g2, float* g3, float* g4, hipStream_t s) {
  if (B <= 0 || N <= 0) return DGCNN_EINVAL;
  hipLaunchKernelGGL(k_sortpool_bwd, dim3(B), dim3(SP_THREADS), 0, s, graph_ptr, perm, gpooled, g1, g2, g3, g4);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// readout forward = SortPooling + the whole dense tail, ONE workgroup of 1024 threads per graph.
// LDS plan (bytes): region A 32 KiB = sort keys, afterwards reused for the pooled rows (11640),
// conv5 weights (6208) and conv6 weights (10240); small activations after it.
// ---------------------------------------------------------------------------------------------
// Which blocks of a launch are the RIDER (the next batch's graph preparation, dg_prep.h) and which are graph workgroups:
// graphs first, rider blocks behind them.  `interleave` mixes them at the ratio nblk : B -- measured at 2048 COLLAB graphs it
// is SLOWER for both readout launches (k_readout_fwd 59.5 -> 72 us with phase A's 160 MB edge stream between the graph
// workgroups, k_tail_bwd 51 -> 56 us with phase B): the graph workgroups are chains of dependent loads whose latency grows
// under a saturated memory system.  Kept as a switch for measurement; no launch uses it.
struct DgRole { bool rider; int idx; };
__device__ __forceinline__ DgRole dg_block_role(int i, int B, int nblk, bool interleave) {
  if (!interleave || nblk <= 0) return DgRole{i >= B, i >= B ? i - B : i};
  const long long T = (long long)B + nblk;
  const int r0 = (int)(((long long)i * nblk) / T), r1 = (int)(((long long)(i + 1) * nblk) / T);      // riders among blocks [0, i), [0, i]
  return DgRole{r1 > r0, r1 > r0 ? r0 : i - r0};
}

// BIG (many graphs): registers capped at 64 so that two workgroups share a CU and hide each other's latency chain
template <bool BIG, bool HEAD = true>
__global__ void __launch_bounds__(RD_THREADS) __attribute__((amdgpu_waves_per_eu(BIG ? 8 : 4)))
k_readout_fwd(int C, TailW w, const int* __restrict__ graph_ptr, const float* __restrict__ x1,
              const float* __restrict__ x2, const float* __restrict__ x3, const float* __restrict__ x4,
              float* __restrict__ pooled, int* __restrict__ perm, float* __restrict__ a5g, float* __restrict__ a6g,
              float* __restrict__ a1dg, uint8_t* __restrict__ maskg, float* __restrict__ logp, int training,
              uint64_t seed, unsigned long long* dbg, int B, DgPrepRider rd) {
  const DgRole role = dg_block_role((int)blockIdx.x, B, rd.nblk, false);
  if (role.rider) {    // rider blocks: phase A of the NEXT batch's graph preparation (dg_prep.h)
    dg_rider_phase_a(role.idx * RD_THREADS + (int)threadIdx.x, rd);
    return;
  }
  if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[7] = clock64();
  __shared__ __attribute__((aligned(16))) unsigned long long region0[RD_REGION0_BYTES / 8];
  __shared__ __attribute__((aligned(16))) char small[RD_SMALL_BYTES];
  const RdSmem M = dg_rd_carve(region0, small);
  const int b = role.idx;
  const int n0 = graph_ptr[b], n = graph_ptr[b + 1] - n0;
#ifdef RD_TIMING      // measurement builds: wall-clock start / end of every workgroup (tools/wg_spans.py)
  if (dbg && threadIdx.x == 0) dbg[1024 + 4 * b] = wall_clock64();
#endif
  dg_readout_fwd_body<BIG, HEAD>(M, b, n0, n, C, w, x4, n0, x1, x2, x3, x4, pooled, perm, a5g, a6g, a1dg, maskg, logp,
                                 training, seed, dbg);
#ifdef RD_TIMING
  if (dbg && threadIdx.x == 0) dbg[1024 + 4 * b + 1] = wall_clock64();
#endif
}

int dg_launch_readout_fwd(int N, int B, int C, const float* params, const DgParams* pl, const int32_t* graph_ptr,
                          const float* x1, const float* x2, const float* x3, const float* x4, float* pooled,
                          int32_t* perm, float* a5, float* a6, float* a1d, uint8_t* drop_mask, float* logp,
                          int training, uint64_t seed, hipStream_t s, const DgPrepRider* rider, bool head) {
  if (B <= 0 || N <= 0 || C < 1 || C > DGCNN_MAX_C) return DGCNN_EINVAL;
  DgPrepRider rd{};
  if (rider) rd = *rider;
  static const bool nobig = dg_knob("DG_NO_BIG_READOUT");      // A/B switch (DG_DEBUG_KNOBS builds only)
  if (!head)       // conv6's output only: the classifier runs batched over graphs (classifier.hip)
    hipLaunchKernelGGL((k_readout_fwd<true, false>), dim3(B + rd.nblk), dim3(RD_THREADS), 0, s, C, dg_tail_w(params, pl), graph_ptr,
                       x1, x2, x3, x4, pooled, perm, a5, a6, a1d, drop_mask, logp, training, seed, dg_debug_buffer(), B, rd);
  else if (B >= DG_TAIL_BIG_MIN_B && !nobig)
    hipLaunchKernelGGL(k_readout_fwd<true>, dim3(B + rd.nblk), dim3(RD_THREADS), 0, s, C, dg_tail_w(params, pl), graph_ptr,
                       x1, x2, x3, x4, pooled, perm, a5, a6, a1d, drop_mask, logp, training, seed, dg_debug_buffer(), B, rd);
  else
    hipLaunchKernelGGL(k_readout_fwd<false>, dim3(B + rd.nblk), dim3(RD_THREADS), 0, s, C, dg_tail_w(params, pl), graph_ptr,
                       x1, x2, x3, x4, pooled, perm, a5, a6, a1d, drop_mask, logp, training, seed, dg_debug_buffer(), B, rd);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

#include "dg_tail_body.h"

template <bool BIG, bool HEAD = true>
__global__ void __launch_bounds__(RD_THREADS) __attribute__((amdgpu_waves_per_eu(BIG ? 8 : 4)))
k_tail_bwd(int B, int C, TailW w, const int* __restrict__ graph_ptr, const int* __restrict__ perm,
           const float* __restrict__ dinv, const float* __restrict__ x4, const float* __restrict__ a5g,
           const float* __restrict__ a6g, const float* __restrict__ a1dg, const float* __restrict__ logp,
           const float* __restrict__ glogp, const int64_t* __restrict__ y, float loss_scale, int training,
           float* __restrict__ dlogit, float* __restrict__ gz1g, float* __restrict__ gz6g,
           float* __restrict__ gz5g, float* __restrict__ gp1, float* __restrict__ gp2, float* __restrict__ gp3,
           float* __restrict__ gas4, float* __restrict__ gb4p, float* __restrict__ lossv,
           float* __restrict__ ptail, const float* __restrict__ pooled, unsigned long long* dbg, DgPrepRider rd, int* gpsel) {
  DgRole role = dg_block_role((int)blockIdx.x, B, rd.nblk_b, false);
  // the rider block that also plans the next dense batch (dg_prep_dense_plan: ONE workgroup, ~8 us at 2048 graphs) trades
  // places with graph 0: dispatched first it runs beside the graph workgroups, dispatched last it was the launch's tail
  if (rd.nblk_b > 0 && rd.dmap) {
    if (blockIdx.x == 0) role = DgRole{true, 0};
    else if ((int)blockIdx.x == B) role = DgRole{false, 0};
  }
  if (role.rider) {    // rider blocks: phase B of the NEXT batch's graph preparation (phase A rode on the
                       // readout launch of this step's forward, complete by now)
    dg_prep_fast_b_body(role.idx * RD_THREADS + (int)threadIdx.x, rd.ei, rd.E, rd.N, rd.B, rd.rowptr,
                        rd.colidx, rd.graph_ptr, rd.graph_eptr, rd.dinv, rd.err, rd.epoch, rd.x, rd.xs, rd.F, rd.batch, rd.bits,
                        rd.dmap, rd.edge_check == 1, rd.max_nodes);
    if (rd.dmap && role.idx == 0) dg_prep_dense_plan((int)threadIdx.x, RD_THREADS, rd.B, rd.graph_ptr, rd.dmap);
    return;
  }
#ifdef RD_TIMING
  if (dbg && threadIdx.x == 0) dbg[1024 + 4 * role.idx + 2] = wall_clock64();
#endif
  dg_tail_bwd_body<BIG, false, HEAD>(role.idx, B, C, w, graph_ptr, perm, dinv, x4, a5g, a6g, a1dg, logp, glogp, y, loss_scale, training,
                                     dlogit, gz1g, gz6g, gz5g, gp1, gp2, gp3, gas4, gb4p, lossv, ptail, pooled, dbg, TbExt{}, gpsel);
#ifdef RD_TIMING
  __syncthreads();
  if (dbg && threadIdx.x == 0) dbg[1024 + 4 * role.idx + 3] = wall_clock64();
#endif
}

// Large batches behind the batched classifier: a workgroup WALKS `per` consecutive graphs and keeps the conv5 / conv6
// weight-gradient partials of all of them in ONE LDS row, stored once at the end -- B / per partial rows instead of B (at 2048
// graphs the per-graph rows were 37 MB written here and read back by k_wgrad), and B / per workgroup launches.  (A first
// persistent form of this kernel spilled 100 registers at the 64-register budget of two workgroups per CU: every address
// piece of the body is loop-invariant and was hoisted out of the walk.  The per-iteration opaque graph index below keeps
// the body's arithmetic inside the iteration -- the fix found on the chain kernels.)
#ifndef DG_TAIL_WALK
#define DG_TAIL_WALK 4
#endif
__global__ void __launch_bounds__(RD_THREADS) __attribute__((amdgpu_waves_per_eu(8)))
k_tail_bwd_walk(int B, int per, int C, TailW w, const int* __restrict__ graph_ptr, const int* __restrict__ perm,
                const float* __restrict__ dinv, const float* __restrict__ x4, const float* __restrict__ a5g,
                const float* __restrict__ a6g, float* __restrict__ gz6g, float* __restrict__ gz5g, float* __restrict__ gp1,
                float* __restrict__ gp2, float* __restrict__ gp3, float* __restrict__ gas4, float* __restrict__ gb4p,
                float* __restrict__ ptail, const float* __restrict__ pooled, int* gpsel) {
  __shared__ float pacc[DG_PT_WF2];
  const int b0 = (int)blockIdx.x * per;
  for (int it = 0; it < per; ++it) {
    int b = b0 + it, tl = (int)threadIdx.x;
    DG_OPAQUE_S(b);
    DG_OPAQUE_V(tl);
    if (b >= B) break;
    dg_tail_bwd_body<true, false, false>(b, B, C, w, graph_ptr, perm, dinv, x4, a5g, a6g, nullptr, nullptr, nullptr, nullptr, 0.f, 0,
                                         nullptr, nullptr, gz6g, gz5g, gp1, gp2, gp3, gas4, gb4p, nullptr, ptail, pooled, nullptr,
                                         TbExt{}, gpsel, pacc, it == 0, tl);
    __syncthreads();
  }
  float* row = ptail + (size_t)blockIdx.x * DG_PTAIL(C);
  for (int t = threadIdx.x; t < DG_PT_WF2; t += RD_THREADS) row[t] = pacc[t];
}

// ---------------------------------------------------------------------------------------------
// Training steps with labels: readout forward and readout backward of a graph need nothing of any other graph (the
// NLL-mean scale 1/B is a constant), so ONE launch runs both per graph -- one dispatch and one cold-read chain fewer
// per step (the backward's operands were written by this very workgroup, on this CU).  Rider range: phase A of the next
// batch's graph preparation, as on k_readout_fwd; phase B then rides on the step's last launch, k_wgrad (conv4's backward when the
// weight gradients take their two-stage form).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RD_THREADS) __attribute__((amdgpu_waves_per_eu(4)))
k_readout_tail(int C, TailW w, const int* __restrict__ graph_ptr, const float* __restrict__ x1,
               const float* __restrict__ x2, const float* __restrict__ x3, const float* __restrict__ x4,
               float* __restrict__ pooled, int* __restrict__ perm, float* __restrict__ a5g, float* __restrict__ a6g,
               float* __restrict__ a1dg, uint8_t* __restrict__ maskg, float* __restrict__ logp, int training, uint64_t seed,
               const float* __restrict__ dinv, const int64_t* __restrict__ y, float loss_scale,
               float* __restrict__ dlogit, float* __restrict__ gz1g, float* __restrict__ gz6g, float* __restrict__ gz5g,
               float* __restrict__ gp1, float* __restrict__ gp2, float* __restrict__ gp3, float* __restrict__ gas4,
               float* __restrict__ gb4p, float* __restrict__ lossv, float* __restrict__ ptail, unsigned long long* dbg,
               int B, DgPrepRider rd, int* gpsel) {
  if ((int)blockIdx.x >= B) {
    dg_rider_phase_a(((int)blockIdx.x - B) * RD_THREADS + (int)threadIdx.x, rd);
    return;
  }
  TbExt ext{};
  {
    __shared__ __attribute__((aligned(16))) unsigned long long region0[RD_REGION0_BYTES / 8];
    __shared__ __attribute__((aligned(16))) char small[RD_SMALL_BYTES];
    const RdSmem M = dg_rd_carve(region0, small);
    {   // (layout of region0 after the sort: dg_readout_fwd_body)
      const float* sp = reinterpret_cast<const float*>(M.region0);
      ext.sp = sp; ext.W5s = sp + 2912; ext.W6s = sp + 2912 + NW5; ext.lg = M.lg;
      ext.flat = M.flat; ext.a5s = M.a5s; ext.a1s = M.a1s; ext.sel = M.sel;
    }
    ext.yb = (threadIdx.x < 64) ? (int)y[blockIdx.x] : 0;      // (wave 0 needs it after the forward half: no cold load there)
    const int b = blockIdx.x;
    const int n0 = graph_ptr[b], n = graph_ptr[b + 1] - n0;
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[14] = clock64();
    dg_readout_fwd_body(M, b, n0, n, C, w, x4, n0, x1, x2, x3, x4, pooled, perm, a5g, a6g, a1dg, maskg, logp, training, seed,
                        dbg);
  }
  __syncthreads();        // (full barrier, vmcnt(0): this graph's activations / perm are written)
  dg_tail_bwd_body<false, true>((int)blockIdx.x, B, C, w, graph_ptr, perm, dinv, x4, a5g, a6g, a1dg, logp, nullptr, y, loss_scale, training,
                                dlogit, gz1g, gz6g, gz5g, gp1, gp2, gp3, gas4, gb4p, lossv, ptail, pooled, dbg, ext, gpsel);
}

int dg_launch_readout_tail(int N, int B, int C, const float* params, const DgParams* pl, const int32_t* graph_ptr,
                           const float* x1, const float* x2, const float* x3, const float* x4, float* pooled, int32_t* perm,
                           float* a5, float* a6, float* a1d, uint8_t* drop_mask, float* logp, int training, uint64_t seed,
                           const float* dinv, const int64_t* y, float loss_scale, float* dlogit, float* gz1, float* gz6,
                           float* gz5, float* gp1, float* gp2, float* gp3, float* gas4, float* gb4p, float* lossv,
                           float* ptail, hipStream_t s, const DgPrepRider* rider, int32_t* gpsel) {
  if (B <= 0 || B >= DG_TAIL_BIG_MIN_B || N <= 0 || C < 1 || C > DGCNN_MAX_C || !y) return DGCNN_EINVAL;
  DgPrepRider rd{};
  if (rider) rd = *rider;
  hipLaunchKernelGGL(k_readout_tail, dim3(B + rd.nblk), dim3(RD_THREADS), 0, s, C, dg_tail_w(params, pl), graph_ptr, x1, x2, x3,
                     x4, pooled, perm, a5, a6, a1d, drop_mask, logp, training, seed, dinv, y, loss_scale, dlogit, gz1, gz6, gz5,
                     gp1, gp2, gp3, gas4, gb4p, lossv, ptail, dg_debug_buffer(), B, rd, gpsel);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}
int dg_readout_tail_max_b() { return DG_TAIL_BIG_MIN_B - 1; }
// batches from which classifier_1 / classifier_2 leave the per-graph readout kernels for the batched form (classifier.hip)
bool dg_classifier_batched(int B) {
  static const bool off = dg_knob("DG_NO_BATCHED_CLASSIFIER");      // A/B switch (DG_DEBUG_KNOBS builds only)
  return B >= DG_TAIL_BIG_MIN_B && !off;
}

int dg_launch_tail_bwd(int N, int B, int C, const float* params, const DgParams* pl, const int32_t* graph_ptr,
                       const int32_t* perm, const float* dinv, const float* x4, const float* a5, const float* a6,
                       const float* a1d, const float* logp, const float* glogp, const int64_t* y,
                       float loss_scale, int training, float* dlogit, float* gz1, float* gz6, float* gz5,
                       float* gp1, float* gp2, float* gp3, float* gas4, float* gb4p, float* lossv, float* ptail,
                       const float* pooled, hipStream_t s, const DgPrepRider* rider, bool head, int32_t* gpsel) {
  if (B <= 0 || N <= 0 || C < 1 || C > DGCNN_MAX_C) return DGCNN_EINVAL;
  if ((glogp == nullptr) == (y == nullptr)) return DGCNN_EINVAL;
  DgPrepRider rd{};
  if (rider) rd = *rider;
  static const bool nobig = dg_knob("DG_NO_BIG_TAIL");      // A/B switch (DG_DEBUG_KNOBS builds only)
  if (!head)       // the classifier's backward ran batched over graphs: gz6 holds the gradient of conv6's output
    hipLaunchKernelGGL((k_tail_bwd<true, false>), dim3(B + rd.nblk_b), dim3(RD_THREADS), 0, s, B, C, dg_tail_w(params, pl), graph_ptr,
                       perm, dinv, x4, a5, a6, a1d, logp, glogp, y, loss_scale, training, dlogit, gz1, gz6, gz5, gp1, gp2,
                       gp3, gas4, gb4p, lossv, ptail, pooled, dg_debug_buffer(), rd, gpsel);
  else if (B >= DG_TAIL_BIG_MIN_B && !nobig)
    hipLaunchKernelGGL(k_tail_bwd<true>, dim3(B + rd.nblk_b), dim3(RD_THREADS), 0, s, B, C, dg_tail_w(params, pl), graph_ptr,
                       perm, dinv, x4, a5, a6, a1d, logp, glogp, y, loss_scale, training, dlogit, gz1, gz6, gz5, gp1, gp2,
                       gp3, gas4, gb4p, lossv, ptail, pooled, dg_debug_buffer(), rd, gpsel);
  else
    hipLaunchKernelGGL(k_tail_bwd<false>, dim3(B + rd.nblk_b), dim3(RD_THREADS), 0, s, B, C, dg_tail_w(params, pl), graph_ptr,
                       perm, dinv, x4, a5, a6, a1d, logp, glogp, y, loss_scale, training, dlogit, gz1, gz6, gz5, gp1, gp2,
                       gp3, gas4, gb4p, lossv, ptail, pooled, dg_debug_buffer(), rd, gpsel);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

int dg_tail_walk_rows(int B) { return dg_cdiv(B, DG_TAIL_WALK); }
int dg_launch_tail_bwd_walk(int N, int B, int C, const float* params, const DgParams* pl, const int32_t* graph_ptr,
                            const int32_t* perm, const float* dinv, const float* x4, const float* a5, const float* a6,
                            float* gz6, float* gz5, float* gp1, float* gp2, float* gp3, float* gas4, float* gb4p, float* ptail,
                            const float* pooled, hipStream_t s, int32_t* gpsel) {
  if (B <= 0 || N <= 0 || C < 1 || C > DGCNN_MAX_C) return DGCNN_EINVAL;
  hipLaunchKernelGGL(k_tail_bwd_walk, dim3(dg_tail_walk_rows(B)), dim3(RD_THREADS), 0, s, B, DG_TAIL_WALK, C, dg_tail_w(params, pl),
                     graph_ptr, perm, dinv, x4, a5, a6, gz6, gz5, gp1, gp2, gp3, gas4, gb4p, ptail, pooled, gpsel);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// weight gradients: ONE launch whose workgroups are partitioned into segments (WgSeg), one per parameter tensor.
//   WG_REDUCE_COL       out[c] = sum of R partial rows (per-workgroup partials of the GCN backward kernels, per-graph
//                       partials of k_tail_bwd), read column-coalesced
//   WG_FC1W_MFMA        classifier_1's weight gradient, a [128 x B].[B x 352] GEMM on the fp32 matrix cores
//   WG_FC1B / WG_SUMB / WG_METRIC   short sums with 64 lanes per output (strided over r, fixed xor butterfly)
//   *_CHUNK             stage 1 of the two-stage form used for batches of more than DG_WG_TWO_STAGE_B graphs
// No floating-point atomics anywhere: fixed reduction orders, bit-reproducible.  The lane that owns an output also
// applies the Adam update when the optimizer is fused.
// ---------------------------------------------------------------------------------------------
enum { WG_FC1B = 0, WG_SUMB, WG_METRIC, WG_FC1W_MFMA, WG_REDUCE_CHUNK, WG_FC1W_MFMA_CHUNK, WG_REDUCE_COL };
#define WG_MAX_SEG 20
struct WgSeg {
  int type;
  int count;         // number of outputs
  int lpo;           // lanes per output
  int R;             // reduction length
  int block0;        // first block of this segment
  int stride;        // WG_REDUCE_COL: floats between partial rows ; WG_METRIC: 2 ; *_CHUNK: row width
  int aux;           // *_CHUNK: total number of rows (graphs) being reduced, R = rows per chunk
  int width;         // WG_REDUCE_CHUNK: columns summed per row (a window of `width` columns of rows `stride` floats apart)
  const float* src;  // partial rows / per-graph values being summed
  float* out;
};
struct WgArgs {
  int nseg, B, C;
  // first block of every segment, contiguous (INT_MAX beyond nseg): a workgroup finds its segment from ONE batch of scalar loads.
  // (Walking seg[k].block0 -- one 48-byte record per step, the kernel arguments cold in the scalar cache at every launch -- was a
  //  chain of ~9 dependent scalar-cache misses in front of every workgroup's first useful load: ~1.5 us of the 7.6 us launch.)
  int blk0[WG_MAX_SEG];
  const float *gz1, *a6;               // classifier_1 operands: d(loss)/d(pre-activation) [B,128], input [B,352]
  // optional fused Adam (torch.optim.Adam defaults semantics): applied by the lane that owns the output
  float *adam_p, *adam_m, *adam_v;     // flat buffers (same layout as grads); null = no optimizer step here
  const float* grads_base;             // to turn an output pointer into a flat index
  float lr, b1, b2, eps, bc1, bc2_sqrt;
  WgSeg seg[WG_MAX_SEG];
};

__device__ __forceinline__ float dg_wg_term(const WgArgs& A, const WgSeg& sg, int i, int r) {
  switch (sg.type) {
    case WG_REDUCE_CHUNK: {   // output i = chunk * width + column: partial sum of rows [chunk*R, chunk*R + R)
      const int ch = i / sg.width, col = i - ch * sg.width, row = ch * sg.R + r;
      return row < sg.aux ? sg.src[(size_t)row * sg.stride + col] : 0.f; }
    case WG_SUMB:   return sg.src[r];
    case WG_METRIC: return sg.src[(size_t)r * 2 + i];
    case WG_FC1B:   return A.gz1[(size_t)r * DGCNN_HID1 + i];
  }
  return 0.f;
}

__device__ __forceinline__ void dg_wg_store(const WgArgs& A, const WgSeg& sg, int i, float acc) {
  sg.out[i] = acc;
  if (A.adam_p) {       // optimizer.step() for this element (train.py:41), same formula as k_adam
    const size_t k = (size_t)(sg.out - A.grads_base) + i;
    const float mi = A.b1 * A.adam_m[k] + (1.f - A.b1) * acc;
    const float vi = A.b2 * A.adam_v[k] + (1.f - A.b2) * acc * acc;
    A.adam_m[k] = mi; A.adam_v[k] = vi;
    const float denom = sqrtf(vi) / A.bc2_sqrt + A.eps;
    A.adam_p[k] = A.adam_p[k] - (A.lr / A.bc1) * (mi / denom);
  }
}

// classifier_1 weight gradient as a small GEMM on the fp32 matrix cores:
//   dW[j][m] = sum_b gz1[b][j] * a6[b][m]  = (gz1^T [128 x B]) . (a6 [B x 352]),  K = B (zero-padded to 4)
// one wave per 16x16 output tile (8 x 22 tiles), v_mfma_f32_16x16x4_f32: a k-ordered fma chain over b.
// Large batches: the K range is cut into chunks (one wave per (chunk, tile)); chunk partials are then summed by a
// WG_REDUCE segment of the second launch.
__device__ __forceinline__ void dg_wg_fc1w_mfma(const WgArgs& A, const WgSeg& sg, int tile, int lane, int kbeg, int B,
                                                float* pout) {
  const int rb = tile / 22, cb = tile - rb * 22;
  const int jr = rb * 16 + (lane & 15), mc = cb * 16 + (lane & 15), kq = lane >> 4;
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  // optimizer state of the 4 elements this lane owns: same memory round trip as the operands
  float pm[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f}, pp[4] = {0.f, 0.f, 0.f, 0.f};
  const bool adam = !pout && A.adam_p;
  const size_t kbase_ = (size_t)(sg.out - A.grads_base);
  if (adam) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t k = kbase_ + (size_t)(rb * 16 + kq * 4 + r) * DGCNN_FLAT + cb * 16 + (lane & 15);
      pm[r] = A.adam_m[k]; pv[r] = A.adam_v[k]; pp[r] = A.adam_p[k];
    }
  }
  for (int k0 = kbeg; k0 < B; k0 += 64) {       // 16 MFMAs (64 graphs) per round: 32 loads in flight per lane, so the
    float av[16], bv[16];                        // reference's batch of 50 is ONE memory round trip
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int bb = k0 + 4 * u + kq;
      av[u] = bb < B ? A.gz1[(size_t)bb * DGCNN_HID1 + jr] : 0.f;
      bv[u] = bb < B ? A.a6[(size_t)bb * DGCNN_FLAT + mc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + 4 * u < B) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = rb * 16 + kq * 4 + r, m = cb * 16 + (lane & 15);
    if (pout) pout[j * DGCNN_FLAT + m] = d[r];
    else if (adam) {
      const size_t k = kbase_ + (size_t)j * DGCNN_FLAT + m;
      const float g = d[r];
      sg.out[j * DGCNN_FLAT + m] = g;
      const float mi = A.b1 * pm[r] + (1.f - A.b1) * g;
      const float vi = A.b2 * pv[r] + (1.f - A.b2) * g * g;
      A.adam_m[k] = mi; A.adam_v[k] = vi;
      const float denom = sqrtf(vi) / A.bc2_sqrt + A.eps;
      A.adam_p[k] = pp[r] - (A.lr / A.bc1) * (mi / denom);
    } else dg_wg_store(A, sg, j * DGCNN_FLAT + m, d[r]);
  }
}

// S1: the stage-1 launch of large batches (chunk sums + split-K classifier_1 only).  Its own instantiation because the
// column-reduction path's 64 loads in flight set the kernel's register count (212: two waves per SIMD) for every segment
// type compiled into it; stage 1's ~1300 workgroups then ran in 2.5 rounds.
template <bool S1>
__global__ void __launch_bounds__(256)
k_wgrad(WgArgs A, DgPrepRider rd, int nb_host) {
  if (!S1 && (int)blockIdx.x >= nb_host) {   // rider range: phase B of the NEXT batch's graph preparation.  This launch is the
                                      // step's last and longest short kernel (7.4 us at batch 50): phase B (5 us alone)
                                      // disappears under it, whereas it stretched k_gcn_bwd1 from 4.8 to 6.1 us
    dg_prep_fast_b_body<256>(((int)blockIdx.x - nb_host) * 256 + (int)threadIdx.x, rd.ei, rd.E, rd.N, rd.B, rd.rowptr, rd.colidx,
                        rd.graph_ptr, rd.graph_eptr, rd.dinv, rd.err, rd.epoch, rd.x, rd.xs, rd.F, rd.batch, rd.bits, rd.dmap, rd.edge_check == 1, rd.max_nodes);
    if (rd.dmap && (int)blockIdx.x == nb_host) dg_prep_dense_plan((int)threadIdx.x, 256, rd.B, rd.graph_ptr, rd.dmap);
    return;
  }
  int si = 0;
#pragma unroll
  for (int k = 1; k < WG_MAX_SEG; ++k) si = (int)blockIdx.x >= A.blk0[k] ? k : si;      // (block0 ascends; INT_MAX beyond nseg)
  const WgSeg sg = A.seg[si];
  if (!S1 && sg.type == WG_FC1W_MFMA) {      // block-uniform branch: 4 waves = 4 tiles per workgroup
    const int tile = ((int)blockIdx.x - sg.block0) * 4 + (threadIdx.x >> 6);
    if (tile < 8 * 22) dg_wg_fc1w_mfma(A, sg, tile, threadIdx.x & 63, 0, A.B, nullptr);
    return;
  }
  if (sg.type == WG_FC1W_MFMA_CHUNK) {   // task = chunk * 176 + tile ; K range of sg.R graphs per chunk
    const int task = ((int)blockIdx.x - sg.block0) * 4 + (threadIdx.x >> 6);
    if (task < sg.count) {
      const int ch = task / (8 * 22), tile = task - ch * (8 * 22);
      const int kbeg = ch * sg.R, kend = min(A.B, kbeg + sg.R);
      dg_wg_fc1w_mfma(A, sg, tile, threadIdx.x & 63, kbeg, kend, sg.out + (size_t)ch * (DGCNN_HID1 * DGCNN_FLAT));
    }
    return;
  }
  if (sg.type == WG_REDUCE_CHUNK && sg.R == 32 && !A.adam_p && sg.count > sg.width) {
    // stage 1 of the large-batch form: output i = (chunk, column) sums 32 consecutive rows; all 32 loads of a lane in flight
    // (the generic path below issues them 16 at a time); same order of additions: (t[u] + t[u+16]) first, then the tree
    const int i = ((int)blockIdx.x - sg.block0) * 256 + (int)threadIdx.x;
    if (i < sg.count) {
      const int ch = i / sg.width, col = i - ch * sg.width;
      const float* sp = sg.src + (size_t)ch * 32 * sg.stride + col;
      const int nrow = min(32, sg.aux - ch * 32);
      float a[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) a[u] = u < nrow ? sp[(size_t)u * sg.stride] : 0.f;
#pragma unroll
      for (int st = 16; st >= 1; st >>= 1)
#pragma unroll
        for (int u = 0; u < st; ++u) a[u] += a[u + st];
      sg.out[i] = a[0];
    }
    return;
  }
  if (!S1 && sg.type == WG_REDUCE_COL) {
    // out[c] = sum_r src[r*stride + c], lanes along the COLUMNS: a workgroup takes 32 columns, its 4 waves x 2 half-waves are
    // EIGHT row groups (every load instruction reads 2 x 128 contiguous bytes of two partial rows), rows dealt round-robin to
    // the groups, up to 32 loads in flight per lane, then a fixed-order combine (half-waves by one shuffle, waves through LDS).
    // Eight groups, not four: the 240 partial rows of the reference's batch of 50 are ONE round of 30 loads per lane.
    __shared__ float red[4][32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane >> 5, cl = lane & 31, grp = 2 * w + half;
    const int c = ((int)blockIdx.x - sg.block0) * 32 + cl;
    const bool live = c < sg.count;
    const bool owner = live && w == 0 && half == 0;
    float pm = 0.f, pv = 0.f, pp = 0.f;
    const size_t k = (size_t)(sg.out - A.grads_base) + (live ? c : 0);
    if (owner && A.adam_p) { pm = A.adam_m[k]; pv = A.adam_v[k]; pp = A.adam_p[k]; }   // optimizer state: same round trip
    const float* sp = sg.src + (live ? c : 0);
    const int R = sg.R, stride = sg.stride;
    float acc = 0.f;
    // DG_WG_COL_DEPTH loads in flight per lane.  32, not 64: the deeper form set the kernel's register count to 212 (two waves
    // per SIMD) for every segment AND for the graph-preparation rider blocks of the same launch; at the reference's batch of
    // 50: k_wgrad 9.0 -> 8.15 us (depth 16: 8.6), step 51.2 -> 50.4 us
#ifndef DG_WG_COL_DEPTH
#define DG_WG_COL_DEPTH 32
#endif
    for (int rb = grp; rb < R; rb += 8 * DG_WG_COL_DEPTH) {
      float a[DG_WG_COL_DEPTH];
#pragma unroll
      for (int u = 0; u < DG_WG_COL_DEPTH; ++u) {
        const int r = rb + 8 * u;
        a[u] = (live && r < R) ? sp[(size_t)r * stride] : 0.f;
      }
#pragma unroll
      for (int st = DG_WG_COL_DEPTH / 2; st >= 1; st >>= 1)
#pragma unroll
        for (int u = 0; u < st; ++u) a[u] += a[u + st];
      acc += a[0];
    }
    acc += __shfl_xor(acc, 32);
    if (half == 0) red[w][cl] = acc;
    __syncthreads();
    if (owner) {
      const float g = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
      sg.out[c] = g;
      if (A.adam_p) {
        const float mi = A.b1 * pm + (1.f - A.b1) * g;
        const float vi = A.b2 * pv + (1.f - A.b2) * g * g;
        A.adam_m[k] = mi; A.adam_v[k] = vi;
        const float denom = sqrtf(vi) / A.bc2_sqrt + A.eps;
        A.adam_p[k] = pp - (A.lr / A.bc1) * (mi / denom);
      }
    }
    return;
  }
  const int gid = ((int)blockIdx.x - sg.block0) * 256 + threadIdx.x;
  const int lpo = sg.lpo;
  const int i = gid / lpo, r0 = gid - i * lpo;
  const bool live = i < sg.count;
  // 16 independent accumulators -> 16 loads in flight per lane (one memory round trip for the usual
  // reduction lengths); combined in a fixed order
  float a[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) a[u] = 0.f;
  {
    // term r of output i = base[r * rs] for r < Rl, whatever the segment type (block-uniform): ONE load site, every load
    // UNCONDITIONAL on a clamped row and selected.  (dg_wg_term's switch put each of the 16 loads of a trip into branch blocks
    // of its own, where it was waited for on the spot -- 16 dependent round trips per trip in the ISA: at 256 graphs the 64-lane
    // sums of classifier_1's bias gradient, db4 and the metrics were FOUR round trips deep instead of one.)  Same order of
    // additions as before: a[u] takes rows r0 + (16 k + u) lpo in k order.
    const float* base = sg.src ? sg.src : A.gz1;
    int rs = 1, Rl = 0;
    if (live) {
      Rl = sg.R;
      switch (sg.type) {
        case WG_REDUCE_CHUNK: {
          const int ch = i / sg.width, col = i - ch * sg.width;
          base = sg.src + (size_t)ch * sg.R * sg.stride + col; rs = sg.stride; Rl = max(0, min(sg.R, sg.aux - ch * sg.R)); break; }
        case WG_SUMB:   base = sg.src; rs = 1; break;
        case WG_METRIC: base = sg.src + i; rs = 2; break;
        case WG_FC1B:   base = A.gz1 + i; rs = DGCNN_HID1; break;
        default: Rl = 0;
      }
    }
    const int rc = max(Rl - 1, 0);
    for (int r = r0; r - r0 < sg.R; r += 16 * lpo) {       // (block-uniform trip count)
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = base[(size_t)min(r + u * lpo, rc) * rs];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] += (r + u * lpo < Rl) ? t[u] : 0.f;
    }
  }
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
    for (int u = 0; u < w; ++u) a[u] += a[u + w];
  float acc = a[0];
  for (int o = 1; o < lpo; o <<= 1) acc += __shfl_xor(acc, o);
  if (live && r0 == 0) {
    if (sg.type == WG_METRIC) sg.out[i] += acc;     // running loss / #correct accumulators
    else dg_wg_store(A, sg, i, acc);
  }
}

static void dg_wg_finish(WgArgs& A) {      // the segment table is complete: fill the contiguous first-block table
  for (int k = 0; k < WG_MAX_SEG; ++k) A.blk0[k] = k < A.nseg ? A.seg[k].block0 : 0x7fffffff;
}
// which: bit 0 = tail parameters (depend on k_tail_bwd only), bit 1 = GCN parameters (depend on the GCN
// backward kernels).  The two halves can run on different streams.
int dg_wgrad_takes_rider(int B) { return B <= dg_wg_two_stage_b() ? 1 : 0; }      // single-launch form only
int dg_launch_wgrad(int which, int N, int B, int F, int C, const DgParams* pl, const DgWs* wl, const void* ws,
                    float* grads, float* metrics, const DgAdam* adam, hipStream_t s, const DgPrepRider* rider, int tail_rows,
                    int gcn_rows) {
  // tail_rows: rows of `ptail` that hold conv5 / conv6 partials (k_tail_bwd_walk: one per workgroup); 0 = one per graph
  // gcn_rows: rows of pa4 / pb3 / pb2 / pb1 that hold GCN partials (the one-launch training kernel: one per graph); 0 = P1 / P32
  const int R1 = gcn_rows > 0 ? gcn_rows : wl->P1, R32 = gcn_rows > 0 ? gcn_rows : wl->P32;
  WgArgs A;
  memset(&A, 0, sizeof(A));
  A.B = B; A.C = C;
  A.gz1 = dg_cptr<float>(ws, wl->gz1); A.a6 = dg_cptr<float>(ws, wl->a6);
  A.grads_base = grads;
  if (adam && adam->params) {
    A.adam_p = adam->params; A.adam_m = adam->exp_avg; A.adam_v = adam->exp_avg_sq;
    A.lr = adam->lr; A.b1 = adam->beta1; A.b2 = adam->beta2; A.eps = adam->eps;
    A.bc1 = (float)(1.0 - pow((double)adam->beta1, (double)adam->step));
    A.bc2_sqrt = (float)sqrt(1.0 - pow((double)adam->beta2, (double)adam->step));
  }
  int nb = 0, ns = 0;
  auto add = [&](int type, int count, int lpo, int R, float* out, const float* src, int stride) {
    WgSeg& g = A.seg[ns++];
    g.type = type; g.count = count; g.lpo = lpo; g.R = R; g.block0 = nb; g.stride = stride; g.src = src; g.out = out;
    nb += dg_cdiv(count * lpo, 256);
  };
  auto add_col = [&](int count, int R, float* out, const float* src, int stride) {   // column-coalesced reduction
    WgSeg& g = A.seg[ns++];
    g.type = WG_REDUCE_COL; g.count = count; g.lpo = 1; g.R = R; g.block0 = nb; g.stride = stride; g.aux = 0;
    g.src = src; g.out = out;
    nb += dg_cdiv(count, 32);
  };
  auto add_tiles = [&](int type, int tiles, float* out) {     // one wave per tile, 4 tiles per workgroup
    WgSeg& g = A.seg[ns++];
    g.type = type; g.count = tiles; g.lpo = 64; g.R = B; g.block0 = nb; g.stride = 0; g.src = nullptr; g.out = out;
    nb += dg_cdiv(tiles, 4);
  };
  // large batches: per-graph vectors are first summed per chunk of DG_WG_ROWS_PER_CHUNK graphs (stage 1, below); the three
  // short per-graph sums -- classifier_1's bias gradient, the metrics, db4 -- go through it too: as 64-lane sums over 2048
  // graphs they were the longest segments of the final launch (15 / 13 / 7.5 us at 2048 graphs, each alone)
  const bool big1 = (which & 1) && B > dg_wg_two_stage_b();
  const int nch_ = dg_cdiv(B, DG_WG_ROWS_PER_CHUNK);
  float* t1_ = const_cast<float*>(dg_cptr<float>(ws, wl->wg_t1));
  float* t3_ = t1_ + (size_t)nch_ * DG_PTAIL(C);            // [nch][128]
  float* t4_ = t3_ + (size_t)nch_ * DGCNN_HID1;             // [nch][2]
  float* t5_ = t4_ + (size_t)nch_ * 2;                      // [nch]
  // segments with the longest dependent latency first: their workgroups are dispatched first
  if (which & 2) {
    const float* pb1 = dg_cptr<float>(ws, wl->pb1);
    const float* pb2 = dg_cptr<float>(ws, wl->pb2);
    const float* pb3 = dg_cptr<float>(ws, wl->pb3);
    const float* pa4 = dg_cptr<float>(ws, wl->pa4);
    add_col(32, R1, grads + pl->off[5], pa4 + 32, 64);                              // db3 (from conv4 backward)
    add_col(32, R1, grads + pl->off[6], pa4, 64);                                   // dW4
    add_col(1024, R32, grads + pl->off[2], pb2, 1056);                              // dW2
    add_col(1024, R32, grads + pl->off[4], pb3, 1056);                              // dW3
    add_col(32 * F, R32, grads + pl->off[0], pb1, 32 * F);                          // dW1
    add_col(32, R32, grads + pl->off[1], pb2 + 1024, 1056);                         // db1 (from layer-2 backward)
    add_col(32, R32, grads + pl->off[3], pb3 + 1024, 1056);                         // db2 (from layer-3 backward)
    if (big1) add(WG_SUMB, 1, 64, nch_, grads + pl->off[7], t5_, 0);                // db4 (chunk partials of stage 1)
    else add(WG_SUMB, 1, 64, B, grads + pl->off[7], dg_cptr<float>(ws, wl->gb4p), 0);    // db4
  }
  if (which & 1) {
    const bool small = B <= dg_wg_two_stage_b();
    const int st = DG_PTAIL(C);
    const float* pt = dg_cptr<float>(ws, wl->ptail);
    int Rt = B;                      // rows the final reduction of the per-graph tail partials runs over
    const int TR = tail_rows > 0 ? tail_rows : B;      // rows holding the conv5 / conv6 columns [0, DG_PT_WF2)
    int Rc = TR, stc = st;           // ... their final reduction length and row stride
    const float* ptc = pt;
    if (!small) {
      // ---- large batches, stage 1 (own launch): per-graph partials -> per-chunk partials, fully coalesced and
      // with (columns x chunks) parallelism; classifier_1's GEMM split over K.  Fixed chunking -> deterministic.
      WgArgs S = A;
      S.adam_p = nullptr; S.adam_m = nullptr; S.adam_v = nullptr;
      int nb1 = 0;
      const int nch = dg_cdiv(B, DG_WG_ROWS_PER_CHUNK), nk = dg_cdiv(B, DG_WG_FC1_KCHUNK);
      float* t1 = const_cast<float*>(dg_cptr<float>(ws, wl->wg_t1));
      float* t2 = const_cast<float*>(dg_cptr<float>(ws, wl->wg_t2));
      WgSeg& g0 = S.seg[0];
      g0.type = WG_FC1W_MFMA_CHUNK; g0.count = nk * 8 * 22; g0.lpo = 64; g0.R = DG_WG_FC1_KCHUNK; g0.block0 = 0;
      g0.stride = 0; g0.aux = B; g0.src = nullptr; g0.out = t2;
      nb1 += dg_cdiv(g0.count, 4);
      // the partial rows in two column windows: conv5 / conv6 (TR rows: per graph, or per walking workgroup) and classifier_2
      // (always per graph, written by the classifier kernel); chunk sums laid out [chunks][window]
      const int nchc = dg_cdiv(TR, DG_WG_ROWS_PER_CHUNK), wf2w = st - DG_PT_WF2;
      float* t1c = t1;                                      // [nchc][DG_PT_WF2]
      float* t1f = t1 + (size_t)nchc * DG_PT_WF2;           // [nch][wf2w]
      WgSeg& g1 = S.seg[1];
      g1.type = WG_REDUCE_CHUNK; g1.count = nchc * DG_PT_WF2; g1.lpo = 1; g1.R = DG_WG_ROWS_PER_CHUNK; g1.block0 = nb1;
      g1.stride = st; g1.width = DG_PT_WF2; g1.aux = TR; g1.src = pt; g1.out = t1c;
      nb1 += dg_cdiv(g1.count, 256);
      auto chunked = [&](int k, const float* src, int width, float* out, int stride = 0) {
        WgSeg& g = S.seg[k];
        g.type = WG_REDUCE_CHUNK; g.count = nch * width; g.lpo = 1; g.R = DG_WG_ROWS_PER_CHUNK; g.block0 = nb1;
        g.stride = stride ? stride : width; g.width = width; g.aux = B; g.src = src; g.out = out;
        nb1 += dg_cdiv(g.count, 256);
      };
      chunked(2, A.gz1, DGCNN_HID1, t3_);
      chunked(3, dg_cptr<float>(ws, wl->lossv), 2, t4_);
      chunked(4, dg_cptr<float>(ws, wl->gb4p), 1, t5_);
      chunked(5, pt + DG_PT_WF2, wf2w, t1f, st);
      S.nseg = 6;
      static const bool s1a = dg_knob("DG_WG_S1_ONLY_MFMA"), s1b = dg_knob("DG_WG_S1_ONLY_REDUCE");      // (timing A/B, debug builds)
      if (s1a) { nb1 = g1.block0; S.nseg = 1; }
      if (s1b) { S.seg[0] = S.seg[1]; S.seg[0].block0 = 0; nb1 -= g1.block0; S.nseg = 1; }
      dg_wg_finish(S);
      hipLaunchKernelGGL(k_wgrad<true>, dim3(nb1), dim3(256), 0, s, S, DgPrepRider{}, nb1);
      DG_CHECK_LAUNCH();
      ptc = t1c; Rc = nchc; stc = DG_PT_WF2;
      pt = t1f - DG_PT_WF2; Rt = nch;      // (so that pt + DG_PT_WF2 is the classifier_2 window of the chunk sums)
      if (nk <= 32) {
        // few chunk rows (<= 4096 graphs): one lane per element, all nk loads in flight, the lane applies Adam -- 176 workgroups
        // for the 45 k elements instead of 704 four-wave workgroups of which one wave owns the outputs (8.8 us alone)
        WgSeg& g = A.seg[ns++];
        g.type = WG_REDUCE_CHUNK; g.count = DGCNN_HID1 * DGCNN_FLAT; g.lpo = 1; g.R = nk; g.block0 = nb;
        g.stride = DGCNN_HID1 * DGCNN_FLAT; g.width = DGCNN_HID1 * DGCNN_FLAT; g.aux = nk; g.src = t2; g.out = grads + pl->off[12];
        nb += dg_cdiv(g.count, 256);
      } else {
        add_col(DGCNN_HID1 * DGCNN_FLAT, nk, grads + pl->off[12], t2, DGCNN_HID1 * DGCNN_FLAT);
      }
    } else {
      add_tiles(WG_FC1W_MFMA, 8 * 22, grads + pl->off[12]);                         // classifier_1 weight: MFMA GEMM
    }
    // conv5 / conv6 / classifier_2: k_tail_bwd left one partial per graph; sum the partials per element
    const int stf = small ? st : st - DG_PT_WF2;      // row stride of the classifier_2 window
    add_col(DGCNN_C5 * DGCNN_CAT, Rc, grads + pl->off[8], ptc + DG_PT_W5, stc);
    add_col(DGCNN_C5, Rc, grads + pl->off[9], ptc + DG_PT_B5, stc);
    add_col(DGCNN_C6 * DGCNN_C5 * DGCNN_KW6, Rc, grads + pl->off[10], ptc + DG_PT_W6, stc);
    add_col(DGCNN_C6, Rc, grads + pl->off[11], ptc + DG_PT_B6, stc);
    add_col(C * DGCNN_HID1, Rt, grads + pl->off[14], pt + DG_PT_WF2, stf);
    add_col(C, Rt, grads + pl->off[15], pt + DG_PT_WF2 + C * DGCNN_HID1, stf);
    if (big1) add_col(DGCNN_HID1, nch_, grads + pl->off[13], t3_, DGCNN_HID1);
    else add(WG_FC1B, DGCNN_HID1, 64, B, grads + pl->off[13], nullptr, 0);
    if (metrics) {                                                                        // train.py:44-45 bookkeeping
      if (big1) add(WG_METRIC, 2, 64, nch_, metrics, t4_, 2);
      else add(WG_METRIC, 2, 64, B, metrics, dg_cptr<float>(ws, wl->lossv), 2);
    }
  }
  A.nseg = ns;
  dg_wg_finish(A);
  if (nb == 0) return DGCNN_OK;
  static const bool split = dg_knob("DG_WGRAD_SPLIT");     // diagnostic (DG_DEBUG_KNOBS builds only): one launch per segment
  if (split) {
    for (int k = 0; k < ns; ++k) {
      WgArgs One = A;
      One.nseg = 1; One.seg[0] = A.seg[k]; One.seg[0].block0 = 0;
      dg_wg_finish(One);
      const int g1 = A.seg[k].type == WG_FC1W_MFMA ? dg_cdiv(A.seg[k].count, 4)
                     : (A.seg[k].type == WG_REDUCE_COL ? dg_cdiv(A.seg[k].count, 32) : dg_cdiv(A.seg[k].count * A.seg[k].lpo, 256));
      hipLaunchKernelGGL(k_wgrad<false>, dim3(g1), dim3(256), 0, s, One, DgPrepRider{}, g1);
    }
    if (rider && rider->nblk_b > 0)       // (diagnostic mode: the rider as a launch of its own)
      hipLaunchKernelGGL(k_wgrad<false>, dim3(4 * rider->nblk_b), dim3(256), 0, s, A, *rider, 0);
    DG_CHECK_LAUNCH();
    return DGCNN_OK;
  }
  DgPrepRider rd{};
  if (rider) rd = *rider;
  hipLaunchKernelGGL(k_wgrad<false>, dim3(nb + 4 * rd.nblk_b), dim3(256), 0, s, A, rd, nb);      // (nblk_b counts 1024-thread blocks)
  DG_CHECK_LAUNCH();
  (void)N;
  return DGCNN_OK;
}

// fixed-order column sums of partial rows through the production reduction (k_wgrad, WG_REDUCE_COL segments):
// used by the stand-alone dgcnn_gcn_bwd
int dg_launch_reduce_cols(int nseg, const DgRedSeg* segs, hipStream_t s) {
  if (nseg < 1 || nseg > WG_MAX_SEG || !segs) return DGCNN_EINVAL;
  WgArgs A;
  memset(&A, 0, sizeof(A));
  int nb = 0;
  for (int k = 0; k < nseg; ++k) {
    WgSeg& g = A.seg[k];
    g.type = WG_REDUCE_COL; g.count = segs[k].count; g.lpo = 1; g.R = segs[k].R; g.block0 = nb; g.stride = segs[k].stride;
    g.aux = 0; g.src = segs[k].src; g.out = segs[k].out;
    nb += dg_cdiv(segs[k].count, 32);
  }
  A.nseg = nseg;
  dg_wg_finish(A);
  A.grads_base = segs[0].out;
  if (nb == 0) return DGCNN_OK;
  hipLaunchKernelGGL(k_wgrad<false>, dim3(nb), dim3(256), 0, s, A, DgPrepRider{}, nb);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults semantics) + fused zero_grad; metrics accumulation
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
       float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, int zero_grads) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  // exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2); p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
  float mi, vi, pi;
  dg_adam_elem(gi, m[i], v[i], p[i], lr / bc1, b1, b2, eps, bc2_sqrt, mi, vi, pi);
  m[i] = mi; v[i] = vi; p[i] = pi;
  if (zero_grads) g[i] = 0.f;
}

int dg_launch_adam(float* p, float* g, float* m, float* v, int64_t n, int64_t step, float lr, float b1,
                   float b2, float eps, int zero_grads, hipStream_t s) {
  if (n <= 0 || step < 1) return DGCNN_EINVAL;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps,
                     (float)bc1, (float)sqrt(bc2), zero_grads);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

__global__ void k_metrics(int B, const float* __restrict__ lossv, float* __restrict__ metrics) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float l = 0.f, c = 0.f;
    for (int b = 0; b < B; ++b) { l += lossv[2 * b]; c += lossv[2 * b + 1]; }
    metrics[0] += l;
    metrics[1] += c;
  }
}

// evaluation bookkeeping (the body of /root/reference/train.py:59-64 after `pred = model(data)`):
//   metrics[0] += mean_b( -logp[b][y_b] )      (nn.NLLLoss() mean, train.py:62,98)
//   metrics[1] += #{ b : argmax_c logp[b][c] == y_b }   (first maximal index, like torch.argmax; train.py:64)
// one workgroup, fixed-order reduction
__global__ void __launch_bounds__(256)
k_eval_metrics(int B, int C, const float* __restrict__ logp, const int64_t* __restrict__ y, float* __restrict__ metrics,
               float scale) {
  __shared__ float sl[256], sc[256];
  float l = 0.f, c = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* row = logp + (size_t)b * C;
    const int yr = (int)y[b];
    const bool ybad = (unsigned)yr >= (unsigned)C;       // out-of-range label: NaN loss (sticky), no out-of-bounds read
    const int yb = ybad ? 0 : yr;
    int am = 0;
    float mx = row[0];
    for (int k = 1; k < C; ++k) { const float v = row[k]; if (v > mx) { mx = v; am = k; } }
    l -= ybad ? __builtin_nanf("") : row[yb];
    c += (am == yb) ? 1.f : 0.f;
  }
  sl[threadIdx.x] = l; sc[threadIdx.x] = c;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) { sl[threadIdx.x] += sl[threadIdx.x + st]; sc[threadIdx.x] += sc[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { metrics[0] += sl[0] * scale; metrics[1] += sc[0]; }
}

int dg_launch_eval_metrics(int B, int C, const float* logp, const int64_t* y, float* metrics, float loss_scale, hipStream_t s) {
  if (B <= 0 || C < 1) return DGCNN_EINVAL;
  hipLaunchKernelGGL(k_eval_metrics, dim3(1), dim3(256), 0, s, B, C, logp, y, metrics,
                     loss_scale != 0.f ? loss_scale : 1.0f / (float)B);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

int dg_launch_metrics(int B, const float* lossv, float* metrics, hipStream_t s) {
  if (B <= 0) return DGCNN_EINVAL;
  hipLaunchKernelGGL(k_metrics, dim3(1), dim3(64), 0, s, B, lossv, metrics);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}
