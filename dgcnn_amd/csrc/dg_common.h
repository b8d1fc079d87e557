// dg_common.h -- shared host/device helpers for libdgcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/dgcnn_hip.h"
// ---- the few places where the sources speak ISA, behind macros: the CPU SIMT emulation build (-DDG_EMU: emu/, test infrastructure)
// compiles the SAME sources as plain C++ ----------------------------------------------------------------------------------------
#ifdef DG_EMU
#define DG_DYN_SMEM(T, name) T* const name = reinterpret_cast<T*>(dg_emu::dyn_smem())
#define DG_WAIT_LGKM() __builtin_amdgcn_wave_barrier()      // (stands where lanes of one wave exchange data through LDS in program order: see DG_LOCKSTEP)
#define DG_OPAQUE_V(x) asm volatile("" : "+m"(x))
#define DG_OPAQUE_S(x) asm volatile("" : "+m"(x))
// The emulation runs a wave's lanes ONE AFTER THE OTHER between wave-level operations; the hardware runs them in lockstep.  Where
// the code relies on lockstep order between lanes of a wave for plain memory accesses ("every lane clears the slot, then lane 0
// fills it"; a lane reading what another lane of its wave just stored), this marker makes the lanes meet.  Nothing on the GPU.
#define DG_LOCKSTEP() __builtin_amdgcn_wave_barrier()
#else
#define DG_LOCKSTEP() do { } while (0)
#define DG_DYN_SMEM(T, name) extern __shared__ __attribute__((aligned(16))) T name[]      // the launch's dynamic LDS, under the kernel's name and element type
#define DG_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define DG_OPAQUE_V(x) asm volatile("" : "+v"(x))      // the value becomes opaque to the optimizer (vector / scalar register)
#define DG_OPAQUE_S(x) asm volatile("" : "+s"(x))
#endif

#define DG_WAVE 64
#define DG_TILE 16            // destination nodes per workgroup tile in the F=32 GCN kernels
#define DG_TILE_THREADS 1024  // 16 waves: one wave per destination node
#ifndef DG_MAX_PART
#define DG_MAX_PART 512       // cap on per-workgroup partial-gradient slots = grid of the backward kernels: two 1024-thread
                              // workgroups per CU.  (1024 measured: the weight-gradient kernel reads twice the partial rows --
                              // step 131 -> 120 us at 256 graphs, 135 -> 125 DD batch 50, 429 -> 411 at 2048; 256: 124 / 129)
#endif
// per-graph partial weight gradients written by k_tail_bwd: conv5 W|b, conv6 W|b, classifier_2 W|b
#define DG_PT_W5 0
#define DG_PT_B5 (DGCNN_C5 * DGCNN_CAT)
#define DG_PT_W6 (DG_PT_B5 + DGCNN_C5)
#define DG_PT_B6 (DG_PT_W6 + DGCNN_C6 * DGCNN_C5 * DGCNN_KW6)
#define DG_PT_WF2 (DG_PT_B6 + DGCNN_C6)
#define DG_PTAIL(C) (DG_PT_WF2 + (C) * DGCNN_HID1 + (C))
#define DG_LDS_PAD 36         // row stride (floats) of 16x32 LDS tiles: 16-B aligned rows

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DG_CHECK_LAUNCH()                                   \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return DGCNN_ELAUNCH; \
  } while (0)

static inline int64_t dg_align(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int dg_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------
// Flat parameter layout (see include/dgcnn_hip.h)
// ---------------------------------------------------------------------------------------
struct DgParams {
  int64_t off[DGCNN_NUM_PARAM_SEGMENTS];
  int64_t total;
};
static inline int dg_param_layout(int F, int C, DgParams* p) {
  if (F < 1 || F > DGCNN_MAX_F || C < 1 || C > DGCNN_MAX_C) return DGCNN_EINVAL;
  const int64_t sizes[DGCNN_NUM_PARAM_SEGMENTS] = {
      (int64_t)DGCNN_HID * F, DGCNN_HID, DGCNN_HID * DGCNN_HID, DGCNN_HID,
      DGCNN_HID * DGCNN_HID, DGCNN_HID, DGCNN_HID, 1,
      DGCNN_C5 * DGCNN_CAT, DGCNN_C5, DGCNN_C6 * DGCNN_C5 * DGCNN_KW6, DGCNN_C6,
      (int64_t)DGCNN_HID1 * DGCNN_FLAT, DGCNN_HID1, (int64_t)C * DGCNN_HID1, C};
  int64_t o = 0;
  for (int i = 0; i < DGCNN_NUM_PARAM_SEGMENTS; ++i) {
    o = dg_align(o, 4);
    p->off[i] = o;
    o += sizes[i];
  }
  p->total = dg_align(o, 4);
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------
// Workspace layout.  Every region is 256-B aligned.
// ---------------------------------------------------------------------------------------
#define DG_WS_REGIONS(X) \
  X(err) X(cnt_in) X(cnt_out) X(rowptr) X(rowptr_t) X(colidx) X(colidx_t) X(dinv) X(graph_ptr) X(graph_eptr) \
  X(hsA) X(hsB) X(h4s) X(x1) X(x2) X(x3) X(x4) X(perm) X(pooled) X(a5) X(a6) X(a1d) X(drop_mask) \
  X(dlogit) X(gz1) X(gz6) X(gz5) X(gp1) X(gp2) X(gp3) X(gas4) X(gasA) X(gasB) X(lossv) X(gb4p) \
  X(pa4) X(pb3) X(pb2) X(pb1) X(ptail) X(ax) X(wg_t1) X(wg_t2) X(adjbits) X(dmap)

// conv1 is evaluated aggregate-first, (A_hat X) W1^T instead of A_hat (X W1^T), whenever the raw feature width is
// <= 32: the gather then moves F floats per edge instead of 32, conv1 needs no stand-alone linear (so graph prep
// depends on the batch only, never on the weights), and its weight gradient dW1 = ga1^T (A_hat X) needs NO
// gather at all in backward -- it rides on conv2's backward kernel from the saved A_hat X.
#define DG_AF_MAX_F 32
#ifndef DG_WG_TWO_STAGE_B
#define DG_WG_TWO_STAGE_B 1024      // batches above this reduce the per-graph weight-gradient partials in two stages
                                     // (measured crossover ~1500 graphs: one launch is faster below, two above)
#endif
// Measurement A/B switches (environment variables) exist only in -DDG_DEBUG_KNOBS builds (make EXTRA=-DDG_DEBUG_KNOBS);
// the shipped library reads exactly ONE environment variable, DG_WG_TWO_STAGE_B=<n> (once per process): it lets
// tests/test_gpu_model.py drive the two-stage weight-gradient form with batches small enough for the CPU oracle.
#include <cstdlib>
#ifdef DG_DEBUG_KNOBS
static inline bool dg_knob(const char* name) { return getenv(name) != nullptr; }
#else
static inline constexpr bool dg_knob(const char*) { return false; }
#endif
static inline int dg_wg_two_stage_b() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("DG_WG_TWO_STAGE_B"); v = (e && atoi(e) > 0) ? atoi(e) : DG_WG_TWO_STAGE_B; }
  return v;
}
#ifndef DG_DENSE_EDGE_COST
#define DG_DENSE_EDGE_COST 12      // dense block form when N * K_estimate <= this * (E + N)   (see dg_use_dense, api.hip)
#endif
#ifndef DG_DENSE_MIN_NODES
#define DG_DENSE_MIN_NODES 28672   // ... and the batch has at least this many nodes (~375 COLLAB-shaped graphs; measured step,
                                   // gather vs dense: 256 graphs 120 vs 138 us, 400: 165 vs 154, 600: 231 vs 188)
#endif
#ifndef DG_WG_ROWS_PER_CHUNK
#define DG_WG_ROWS_PER_CHUNK 32
#endif
#define DG_WG_T1_EXTRA (DGCNN_HID1 + 2 + 1 + 5)      // per chunk, behind the tail partials: classifier_1's bias gradient, {loss, #correct}, db4
#ifndef DG_WG_FC1_KCHUNK
#define DG_WG_FC1_KCHUNK 128
#endif
static inline int dg_af_lfp(int F) { int l = 0; while ((1 << l) < F) ++l; return l; }   // log2 of lanes per neighbour row

struct DgWs {
#define X(n) int64_t n;
  DG_WS_REGIONS(X)
#undef X
  int64_t total;
  int P32;   // partial slots (= grid) of the F=32 backward kernels
  int P1;    // partial slots (= grid) of the F=1 backward kernel
};

static inline int dg_grid32(int N) {
  int tiles = dg_cdiv(N, DG_TILE);
  return tiles < 1 ? 1 : (tiles > DG_MAX_PART ? DG_MAX_PART : tiles);
}
static inline int dg_grid1(int N) {   // conv4 backward: 16 waves (1024 threads) per workgroup, wave per node
  int b = dg_cdiv(N, 16);
  return b < 1 ? 1 : (b > DG_MAX_PART ? DG_MAX_PART : b);
}

static inline int dg_ws_layout(int N, int E, int B, int F, int C, DgWs* w) {
  if (N < 0 || E < 0 || B < 0 || F < 1 || F > DGCNN_MAX_F || C < 1 || C > DGCNN_MAX_C) return DGCNN_EINVAL;
  int64_t o = 0;
  const int64_t n = N, e = E, b = B;
  w->P32 = dg_grid32(N);
  w->P1 = dg_grid1(N);
#define R(name, bytes) do { w->name = o; o = dg_align(o + (int64_t)(bytes), 256); } while (0)
  R(err, 32);        // 8 epoch-tagged words (include/dgcnn_hip.h: err[0..7])
  R(cnt_in, 4 * (n + 1));
  R(cnt_out, 4 * (n + 1));
  R(rowptr, 4 * (n + 1));
  R(rowptr_t, 4 * (n + 1));
  R(colidx, 4 * e);
  R(colidx_t, 4 * e);
  R(dinv, 4 * n);
  R(graph_ptr, 4 * (b + 1));
  R(graph_eptr, 4 * (b + 1));
  R(hsA, 4 * n * 32);   // pre-scaled linear outputs, ping; before conv1 (F <= 32) it holds xs = dinv*x [N,F] from graph prep
  R(hsB, 4 * n * 32);
  R(h4s, 4 * n);
  R(x1, 4 * n * 32);
  R(x2, 4 * n * 32);
  R(x3, 4 * n * 32);
  R(x4, 4 * n);
  R(perm, 4 * b * DGCNN_K);
  R(pooled, 4 * b * DGCNN_K * DGCNN_CAT);
  R(a5, 4 * b * DGCNN_C5 * DGCNN_K);
  R(a6, 4 * b * DGCNN_FLAT);
  R(a1d, 4 * b * DGCNN_HID1);
  R(drop_mask, b * DGCNN_HID1);
  R(dlogit, 4 * b * C);
  R(gz1, 4 * b * DGCNN_HID1);
  R(gz6, 4 * b * DGCNN_FLAT);
  R(gz5, 4 * b * DGCNN_C5 * DGCNN_K);
  R(gp1, 4 * n * 32);
  R(gp2, 4 * n * 32);
  R(gp3, 4 * n * 32);
  R(gas4, 4 * n);
  R(gasA, 4 * n * 32);
  R(gasB, 4 * n * 32);
  R(lossv, 4 * 2 * b);
  R(gb4p, 4 * b);
  R(pa4, 4 * (int64_t)w->P1 * 64);
  R(pb3, 4 * (int64_t)w->P32 * 1056);
  R(pb2, 4 * (int64_t)w->P32 * 1056);
  R(pb1, 4 * (int64_t)w->P32 * 32 * F);
  R(ptail, 4 * b * (int64_t)DG_PTAIL(C));
  R(ax, F <= DG_AF_MAX_F ? 4 * n * F : 0);      // aggregated raw input (aggregate-first conv1), saved for dW1
  // large batches only: stage-1 buffers of the two-stage weight-gradient reduction (tail.hip, dg_launch_wgrad)
  R(wg_t1, B > dg_wg_two_stage_b() ? 4 * (int64_t)dg_cdiv(B, DG_WG_ROWS_PER_CHUNK) * (DG_PTAIL(C) + DG_WG_T1_EXTRA) : 0);
  R(wg_t2, B > dg_wg_two_stage_b() ? 4 * (int64_t)dg_cdiv(B, DG_WG_FC1_KCHUNK) * DGCNN_HID1 * DGCNN_FLAT : 0);
  // dense per-graph block structures (dg_prep.h): adjacency bitmap (five stride classes) + work-item map
  R(adjbits, 4 * 31 * n);
#if defined(DGD_ROWS) && DGD_ROWS == 64
  R(dmap, 4 * (3096 + 3 * (n / 64 + b + 1) + 2 * b + 4));
#else
  R(dmap, 4 * (3096 + 3 * (n / 128 + b + 1) + 2 * b + 4));
#endif      // item table: shares + records (dg_prep.h: dgd_table_ints)
#undef R
  w->total = o;
  return DGCNN_OK;
}

template <typename T>
static inline T* dg_ptr(void* ws, int64_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }
template <typename T>
static inline const T* dg_cptr(const void* ws, int64_t off) {
  return reinterpret_cast<const T*>(static_cast<const char*>(ws) + off);
}

// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------
#ifdef __HIPCC__
// Wave-wide sum on the DPP data path (no LDS crossbar round trips): quad swaps, row rotations,
// then row broadcasts; fixed order -> deterministic.  The total is returned in EVERY lane (read back
// from lane 63 through an SGPR).  Same sequence as rocPRIM's wave64 DPP reduce.
#define DG_DPP(v, ctrl, rmask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, true))
__device__ __forceinline__ float dg_wave_sum(float v) {
  v += DG_DPP(v, 0xB1, 0xf);    // quad_perm:[1,0,3,2]
  v += DG_DPP(v, 0x4E, 0xf);    // quad_perm:[2,3,0,1]
  v += DG_DPP(v, 0x124, 0xf);   // row_ror:4
  v += DG_DPP(v, 0x128, 0xf);   // row_ror:8     -> every lane holds its 16-lane row total
  v += DG_DPP(v, 0x142, 0xa);   // row_bcast:15  -> rows 1,3 += previous row
  v += DG_DPP(v, 0x143, 0xc);   // row_bcast:31  -> rows 2,3 += lanes 0..31 total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over each 32-lane half separately (lanes 0..31 / 32..63); result in every lane of the half
__device__ __forceinline__ float dg_half_sum(float v) {
  v += DG_DPP(v, 0xB1, 0xf);
  v += DG_DPP(v, 0x4E, 0xf);
  v += DG_DPP(v, 0x124, 0xf);
  v += DG_DPP(v, 0x128, 0xf);
  v += DG_DPP(v, 0x142, 0xa);   // lanes 16..31 (and 48..63) now hold the half total
  const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
  const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return (threadIdx.x & 32) ? hi : lo;
}
// XCD-aware tile order (MI355X: 8 XCDs, private 4 MiB L2 each, workgroup b is observed to run on XCD
// b % 8): give each XCD a CONTIGUOUS range of node tiles so the rows a tile gathers (its own graph's,
// i.e. nearby tiles') are served by that XCD's L2 instead of being fetched into all eight.  Bijective
// for any tile count; placement only affects speed, never results.
__device__ __forceinline__ int dg_xcd_tile(int t, int numTiles) {
  const int q = numTiles >> 3, r = numTiles & 7;
  const int xcd = t & 7, k = t >> 3;
  // XCD x owns q+1 tiles if x < r else q tiles; tiles with k beyond the owner's share wrap to the tail
  const int share = xcd < r ? q + 1 : q;
  if (k < share) return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  return t;   // unreachable for t < numTiles when iterated as below; keeps the map total
}
// One 16x16 output tile of C = A.B on the fp32 matrix cores, one wavefront, v_mfma_f32_16x16x4_f32:
//   fa(m, k) -> A[m][k],  fb(k, n) -> B[k][n]   (lane-callable, return 0 outside the real extents = padding)
//   st(m, n, v) receives the 4 elements this lane owns.   K is rounded up to a multiple of 4 by the padding.
// The sum over k is a k-ordered fma chain (deterministic).  Operands typically come from LDS: a scalar
// VALU formulation of these small products is LDS-read-bound (2 reads per FMA), the matrix unit needs 2
// reads per 64x4x... block of FMAs.
template <typename FA, typename FB, typename ST>
__device__ __forceinline__ void dg_mfma_tile16(int m0, int n0, int K, int lane, FA fa, FB fb, ST st) {
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  const int mi = lane & 15, kq = lane >> 4;
  // operands of 8 k-steps are fetched together (16 LDS reads in flight), then the 8 dependent MFMAs issue
  // back to back: one LDS round trip per 8 MFMAs instead of one per MFMA
  for (int k0 = 0; k0 < K; k0 += 32) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 4 * u + kq;
      av[u] = (k0 + 4 * u < K) ? fa(m0 + mi, k) : 0.f;
      bv[u] = (k0 + 4 * u < K) ? fb(k, n0 + mi) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + 4 * u < K) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) st(m0 + kq * 4 + r, n0 + mi, d[r]);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE: a process-wide `static bool` guard raised the LDS limit on the
// first GPU a process drove and on no other (launches above 64 KB then fail there).  One bit per device ordinal, atomic.
#include <atomic>
struct DgPerDeviceOnce {
  std::atomic<unsigned long long> mask{0ull};
  static int current() { int d = 0; if (hipGetDevice(&d) != hipSuccess) d = 0; return d & 63; }
  bool needed() const {                             // true: the calling thread's device has not been set up yet (set-ups are idempotent)
    return ((mask.load(std::memory_order_acquire) >> current()) & 1ull) == 0ull;
  }
  void done() { mask.fetch_or(1ull << current(), std::memory_order_release); }
};

// the same tile with a COMPILE-TIME K (the dense tail's products: K = 40, 12, 32, 16): exactly ceil(K/4) steps, no run-time
// trip counts or per-step bounds tests on k0 -- every operand fetched up front (<= 10 steps) or 8 steps at a time
template <int K, typename FA, typename FB, typename ST>
__device__ __forceinline__ void dg_mfma_tile16_k(int m0, int n0, int lane, FA fa, FB fb, ST st) {
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  const int mi = lane & 15, kq = lane >> 4;
  constexpr int STEPS = (K + 3) / 4, CH = STEPS <= 10 ? STEPS : 8;
#pragma unroll
  for (int s0 = 0; s0 < STEPS; s0 += CH) {
    float av[CH], bv[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u)
      if (s0 + u < STEPS) { const int k = 4 * (s0 + u) + kq; av[u] = fa(m0 + mi, k); bv[u] = fb(k, n0 + mi); }
#pragma unroll
    for (int u = 0; u < CH; ++u)
      if (s0 + u < STEPS) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) st(m0 + kq * 4 + r, n0 + mi, d[r]);
}

// One element of torch.optim.Adam (defaults semantics; /root/reference/train.py:41,99) with the operation sequence PINNED: the
// two moment updates as separately rounded products and one add, the parameter update as one fused multiply-add.  That is
// what the compiler made of `b1 * m + (1 - b1) * g` etc. in k_adam; left to the contraction heuristic, the same source
// expression in the exchange kernel's four-elements-per-thread form became v_pk_fma -- and the one-shot route stopped being
// bit-identical to the all_reduce + k_adam route (tests/test_dist_gloo.py compares them bit for bit).
__device__ __forceinline__ void dg_adam_elem(float g, float m, float v, float p, float lr_over_bc1, float b1, float b2, float eps,
                                             float bc2_sqrt, float& mo, float& vo, float& po) {
#pragma clang fp contract(off)      // (HIP's __fmul_rn / __fadd_rn are plain operators in this toolchain and contract like them)
  const float mi = b1 * m + (1.f - b1) * g;
  const float vi = b2 * v + ((1.f - b2) * g) * g;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  mo = mi; vo = vi;
  po = __builtin_fmaf(-lr_over_bc1, mi / denom, p);
}
// workgroup barrier that orders LDS traffic only: outstanding GLOBAL stores/loads are NOT drained
// (a plain __syncthreads() waits vmcnt(0), i.e. a full HBM write round trip per barrier)
#ifdef DG_EMU
__device__ __forceinline__ void dg_lds_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void dg_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#endif
__device__ __forceinline__ float4 dg_add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 dg_shfl_xor4(float4 a, int m) {
  return make_float4(__shfl_xor(a.x, m), __shfl_xor(a.y, m), __shfl_xor(a.z, m), __shfl_xor(a.w, m));
}
// tanh on the hardware transcendental units: 1 - 2/(2^(2x*log2 e) + 1)  (v_exp_f32 + v_rcp_f32, both
// <= 1 ulp).  Absolute error <= ~1.2e-7 over the whole range (saturates cleanly to +-1, exact 0 at 0),
// ~5 instructions instead of ~40 for the libm routine; the SAME function is used by every kernel so
// the tiled and fused paths stay bit-identical.  Backward uses 1 - y^2 of the stored y.
__device__ __forceinline__ float dg_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // e^(2x)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}
// counter-based dropout bit: splitmix64 of (seed, index); keep with probability 1/2
__device__ __forceinline__ bool dg_keep(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (z >> 63) != 0;
}
// all-ascending bitonic sorting network over data[0..n) executed by one whole workgroup;
// the tail up to the next power of two is virtual +inf padding (compare-exchanges that
// would touch it are no-ops because every comparison sorts ascending).
template <typename T, typename PTR>
__device__ void dg_block_bitonic(PTR data, int n) {
  int P = 1;
  while (P < n) P <<= 1;
  const int half = P >> 1;
  for (int k = 2; k <= P; k <<= 1) {
    const int hk = k >> 1;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {   // flip stage
      const int blk = i / hk, within = i - blk * hk;
      const int a = blk * k + within, b = blk * k + k - 1 - within;
      if (b < n) {
        const T va = data[a], vb = data[b];
        if (va > vb) { data[a] = vb; data[b] = va; }
      }
    }
    __syncthreads();
    for (int j = k >> 2; j > 0; j >>= 1) {                  // disperse stages
      for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const int blk = i / j, within = i - blk * j;
        const int a = blk * 2 * j + within, b = a + j;
        if (b < n) {
          const T va = data[a], vb = data[b];
          if (va > vb) { data[a] = vb; data[b] = va; }
        }
      }
      __syncthreads();
    }
  }
}
// Cooperative global -> LDS staging of N floats by THREADS threads, split in two steps so that EVERY load is in
// flight before the first LDS store waits for one: a plain `for (t = tid; t < N; t += THREADS) lds[t] = g[t]` is
// compiled as load / s_waitcnt vmcnt(0) / ds_write per iteration, i.e. one full memory round trip per iteration.
template <int N, int THREADS>
struct DgStage {
  static constexpr int IT = (N + THREADS - 1) / THREADS;
  float v[IT];
  __device__ __forceinline__ void load(const float* __restrict__ src, int tid) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int t = tid + i * THREADS;
      v[i] = t < N ? src[t] : 0.f;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ dst, int tid) const {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int t = tid + i * THREADS;
      if (t < N) dst[t] = v[i];
    }
  }
};

// 16 bytes per lane (N a multiple of 4, source and destination 16-byte aligned): a quarter of the vector-memory instructions --
// every wave-level load costs the CU's address path ~14 cycles whatever its width, and a prologue that stages a few KB with
// dword loads from all 16 waves is ~100 of them in front of the first barrier
template <int N, int THREADS>
struct DgStage4 {
  static_assert(N % 4 == 0, "whole float4 pieces");
  static constexpr int N4 = N / 4, IT = (N4 + THREADS - 1) / THREADS;
  float4 v[IT];
  __device__ __forceinline__ void load(const float* __restrict__ src, int tid) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int t = tid + i * THREADS;
      v[i] = *reinterpret_cast<const float4*>(src + 4 * (t < N4 ? t : 0));      // (unconditional on a clamped piece)
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ dst, int tid) const {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int t = tid + i * THREADS;
      if (t < N4) *reinterpret_cast<float4*>(dst + 4 * t) = v[i];
    }
  }
};

// ---- aggregate-first conv1 (F <= DG_AF_MAX_F) -------------------------------------------------------
// One wavefront, one destination row.  Lane = (g = lane >> lfp neighbour group, q = lane & (2^lfp - 1) feature):
// 64 >> lfp neighbours in flight per wave-instruction.  Returns, in every lane with q < F,
//     sum_{e in [start,end)} dinv[col[e]] * x[col[e]][q]  +  dinv[self] * x[self][q]        (self term last, group 0)
// combined over the groups by a fixed xor butterfly.  PRESCALED: xsrc already holds dinv*x (LDS copy, stride F).
// The product is rounded on its own (never contracted into the add) so both variants are bit-identical.
__device__ __forceinline__ int dg_af_lfp_dev(int F) { return F <= 1 ? 0 : 32 - __builtin_clz(F - 1); }
template <bool PRESCALED>
__device__ __forceinline__ float dg_af_gather(const float* __restrict__ xsrc, const float* __restrict__ dsrc, int F,
                                              int lfp, const int* __restrict__ col, int start, int end, int self,
                                              int lane) {
  const int q = lane & ((1 << lfp) - 1), g = lane >> lfp, sh = 6 - lfp;
  const bool qa = q < F;
  float acc = 0.f, vself = 0.f;
  if (g == 0 && qa) {
    if (PRESCALED) vself = xsrc[self * F + q];
    else { vself = dsrc[self] * xsrc[(size_t)self * F + q]; DG_OPAQUE_V(vself); }
  }
  for (int base = start; base < end; base += 64) {
    const int cnt = min(64, end - base);
    const int cj = lane < cnt ? col[base + lane] : 0;
    const int iters = (cnt + (1 << sh) - 1) >> sh;
    for (int it0 = 0; it0 < iters; it0 += 8) {      // 8 neighbour rounds in flight, summed in round order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = ((it0 + u) << sh) + g;
        const int j = __shfl(cj, idx & 63);
        v[u] = 0.f;
        if (idx < cnt && qa) {
          if (PRESCALED) v[u] = xsrc[j * F + q];
          else { v[u] = dsrc[j] * xsrc[(size_t)j * F + q]; DG_OPAQUE_V(v[u]); }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if ((((it0 + u) << sh) + g) < cnt && qa) acc += v[u];
    }
  }
  if (g == 0 && qa) acc += vself;
  for (int off = 1 << lfp; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
  return acc;
}
// conv1's dense step on the aggregated row: lane c = lane & 31 returns sum_f ax[f] * W1[c][f] (k-ordered fma chain);
// ax[f] lives in lane f of the wave (group 0), Wt = W1 transposed [F][32] in LDS.
__device__ __forceinline__ float dg_af_transform(float ax, int F, const float* __restrict__ Wt, int lane) {
  const int c = lane & 31;
  float pre = 0.f;
  for (int f = 0; f < F; ++f) {
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ax), f));
    pre = fmaf(a, Wt[f * 32 + c], pre);
  }
  return pre;
}
#endif

struct DgDense;
// dense per-graph block forms of the aggregation kernels (gcn_dense.hip)
int dg_launch_gcn_fwd32d(int mode, int bf16_in, int bf16_out, const DgDense* G, const float* dinv, const void* hs,
                         const float* bias, float* xout, const float* Wnext, void* hs_next, hipStream_t s,
                         hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
int dg_launch_gcn_fwd_af_d(int bf16_out, const DgDense* G, int F, const float* dinv, const float* xs, const float* W1,
                           const float* bias, float* ax, float* xout, const float* Wnext, void* hs_next, hipStream_t s,
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
int dg_launch_gcn_fwd1d(const DgDense* G, const float* dinv, const float* h4s, const float* bias, float* x4, hipStream_t s);
int dg_launch_gcn_bwd32d(const DgDense* G, const float* dinv, const float* gas, const float* Wl, const float* xprev,
                         const float* gpprev, float* gas_prev, float* part, int P32, hipStream_t s,
                         const float* ax = nullptr, int Fa = 0, float* part1 = nullptr);
int dg_launch_gcn_bwd1d(const DgDense* G, const float* dinv, const float* gas4, const float* W4, const float* x3,
                        const float* gp3, float* gas3, float* pa4, int P1, hipStream_t s);
// graph-chain kernels (gcn_chain.hip): conv1..conv4 of a graph in one workgroup, hs resident in LDS
int dg_chain_max_nodes();
int dg_chain_small_rows(int B);
int dg_chain_needs_schedule(int B);
int dg_chain_bwd_max_nodes();
int dg_launch_chain_bwd_a(int N, int B, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv, const float* gas4,
                          const float* W4, const float* W3, const float* x3, const float* gp3, const float* x2, const float* gp2,
                          float* gas2, float* pa4, int P1, float* pb3, int P32, int32_t* dmap, hipStream_t s, const int32_t* gpsel = nullptr);
int dg_launch_chain_bwd_b(int N, int B, int Fa, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv, const float* gas2,
                          const float* W2, const float* x1, const float* gp1, const float* ax, float* pb2, float* pb1, int P32,
                          int32_t* dmap, hipStream_t s, const int32_t* gpsel = nullptr);
int dg_chain_train_max_b();
int dg_chain_train_max_nodes();
int dg_chain_eval_max_nodes();       // largest graph of the one-launch EVALUATION kernel (512)
int dg_launch_chain_readout_tail(int N, int B, int F, int C, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv,
                                 const float* xs, const float* params, const struct DgParams* pl, float* ax, float* x1, float* x2,
                                 float* x3, float* x4, float* pooled, int32_t* perm, float* a5, float* a6, float* a1d,
                                 uint8_t* drop_mask, float* logp, int training, uint64_t seed, const int64_t* y, float loss_scale,
                                 float* dlogit, float* gz1, float* gz6, float* gz5, float* gp1, float* gp2, float* gp3, float* gas4,
                                 float* gb4p, float* lossv, float* ptail, int32_t* err, uint32_t epoch, float* gas3, float* pa4, int P1,
                                 hipStream_t s, const struct DgPrepRider* rider = nullptr,
                                 hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                                 float* pb3 = nullptr, float* pb2 = nullptr, float* pb1 = nullptr, int bf16 = 0,
                                 int* fused_b_out = nullptr,       // != nullptr: phase B of the rider may join this launch; says if it did
                                 int insym = 0);                   // 1: the kernel verifies the bitmap's symmetry (reverse edges) itself
int dg_launch_chain_readout_eval(int N, int B, int F, int C, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv,
                                 const float* xs, const float* params, const struct DgParams* pl, float* ax, float* x1, float* x2,
                                 float* x3, float* x4, float* pooled, int32_t* perm, float* a5, float* a6, float* a1d,
                                 uint8_t* drop_mask, float* logp, int training, uint64_t seed, const int64_t* y, float loss_scale,
                                 float* evl, unsigned int* ev_ctr, unsigned int* ev_host, float* metrics, int32_t* err, uint32_t epoch,
                                 hipStream_t s, const struct DgPrepRider* rider = nullptr, hipEvent_t ev_start = nullptr,
                                 hipEvent_t ev_stop = nullptr, int bf16 = 0, int* fused_b_out = nullptr, int insym = 0,
                                 int max_nodes = 256);      // (max_nodes in 257..512: the two-tiles-per-wave instantiation)
int dg_launch_chain_fwd(int N, int B, int F, int max_nodes, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv,
                        const float* xs, const float* params, const struct DgParams* pl, float* ax, float* x1, float* x2,
                        float* x3, float* x4, int32_t* dmap, int bf16, hipStream_t s, hipEvent_t ev_start = nullptr,
                        hipEvent_t ev_stop = nullptr);
// graph-per-workgroup fused forward in the dense block form (gcn_dense.hip): conv1..conv4 + readout, one launch
struct DgParams;
int dg_fused_d_max_nodes();
int dg_launch_fused_fwd_d(int N, int B, int F, int C, const float* params, const DgParams* pl, const float* x,
                          const int32_t* rowptr, const int32_t* colidx, const float* dinv, const int32_t* graph_ptr, float* ax,
                          float* x1, float* x2, float* x3, float* x4, float* pooled, int32_t* perm, float* a5, float* a6,
                          float* a1d, uint8_t* drop_mask, float* logp, int training, uint64_t seed, int32_t* err, uint32_t epoch,
                          hipStream_t s, const struct DgPrepRider* rider = nullptr, hipEvent_t ev_start = nullptr,
                          hipEvent_t ev_stop = nullptr);
#ifndef DG_FUSED_D_MAX_B
#define DG_FUSED_D_MAX_B 0          // fused dense forward (one workgroup per graph) chosen automatically up to this many graphs.
                                    // 0 = never: measured at the reference's batch of 50 COLLAB-shaped graphs it takes 26.6 us
                                    // against 29.5 us for the five launches it replaces (step 56.9 vs 57.6 us) -- the longest
                                    // graph's serial chain (n = 126: 50 k cycles) sets its duration; FORCE_FUSED + AGG_DENSE selects it
#endif
struct DgLinFirst { const float* x; const float* W; float* hs; int F; };   // optional conv1 linear riding on the prep launch
// kernel launchers implemented in the .hip files (host side, internal linkage across TUs)
int dg_launch_prep(const int64_t* edge_index, int E, const int64_t* batch, int N, int B,
                   int32_t* rowptr, int32_t* colidx, int32_t* rowptr_t, int32_t* colidx_t,
                   float* dinv, int32_t* graph_ptr, int32_t* graph_eptr, int32_t* cnt_in, int32_t* cnt_out,
                   int32_t* err, int flags, uint32_t epoch, hipStream_t s, const DgLinFirst* lf = nullptr,
                   int* lin_done = nullptr, uint32_t* bits = nullptr, int32_t* dmap = nullptr, int edge_check = 0,
                   int max_nodes = 0);
int dg_launch_prep_phase_b(const struct DgPrepRider* rd, hipStream_t s);
int dg_launch_prep_sym(const int64_t* edge_index, int E, int N, int B, const int64_t* batch, const int32_t* graph_ptr,
                       const uint32_t* bits, int32_t* err, uint32_t epoch, hipStream_t s);
int dg_launch_lin_first(int N, int F, const float* x, const float* W, const float* dinv, float* hs,
                        int Fout, hipStream_t s, int bf16_out = 0);
struct DgRedSeg { int count, R, stride; const float* src; float* out; };     // out[c] = sum_{r<R} src[r*stride + c]
int dg_launch_reduce_cols(int nseg, const DgRedSeg* segs, hipStream_t s);
// mode: 0 = fused next 32x32 linear (MFMA), 1 = fused next 32->1 dot, 2 = no post-step
int dg_launch_gcn_fwd32(int mode, int N, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                        const float* hs, const float* bias, float* xout, const float* Wnext, float* hs_next,
                        hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int E = -1);
int dg_narrow_gather_enable(int on);      // run-time A/B switch of the eight-lanes-per-node forms (gcn.hip); returns the previous setting
int dg_launch_gcn_fwd1(int N, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                       const float* h4s, const float* bias, float* x4, hipStream_t s, int E = -1);      // E: directed edges without self loops (< 0: unknown)
int dg_launch_gcn_bwd1(int N, const int32_t* rowptr_t, const int32_t* colidx_t, const float* dinv,
                       const float* gas4, const float* W4, const float* x3, const float* gp3,
                       float* gas3, float* pa4, int P1, hipStream_t s, const struct DgPrepRider* rider = nullptr,
                       const int32_t* gpsel = nullptr, int E = -1);
// readout forward + readout backward of a training step (labels) as ONE launch (tail.hip)
int dg_launch_readout_tail(int N, int B, int C, const float* params, const DgParams* pl, const int32_t* graph_ptr,
                           const float* x1, const float* x2, const float* x3, const float* x4, float* pooled, int32_t* perm,
                           float* a5, float* a6, float* a1d, uint8_t* drop_mask, float* logp, int training, uint64_t seed,
                           const float* dinv, const int64_t* y, float loss_scale, float* dlogit, float* gz1, float* gz6,
                           float* gz5, float* gp1, float* gp2, float* gp3, float* gas4, float* gb4p, float* lossv,
                           float* ptail, hipStream_t s, const struct DgPrepRider* rider = nullptr, int32_t* gpsel = nullptr);
int dg_readout_tail_max_b();
// which: 3 or 2 -> MFMA gx + partial gW(32x32) ; 1 -> first layer (partial gW1 [32,F] only)
int dg_launch_gcn_bwd32(int first, int N, int F, const int32_t* rowptr_t, const int32_t* colidx_t,
                        const float* dinv, const float* gas, const float* Wl, const float* xprev,
                        const float* gpprev, float* gas_prev, float* part, int P32, hipStream_t s,
                        const float* ax = nullptr, int Fa = 0, float* part1 = nullptr, int E = -1,
                        const int32_t* gpsel = nullptr);      // gpsel: flag word per node of SPARSE SortPooling-gradient slabs
int dg_narrow_applies(int N, int E);      // the narrow forms take a batch of N nodes / E directed edges (gcn.hip)
// conv1 aggregate-first forward (F <= DG_AF_MAX_F): ax = A_hat x saved, x1 = tanh(ax W1^T + b1), hs_next = dinv*(x1 Wnext^T)
int dg_launch_gcn_fwd_af(int N, int F, const int32_t* rowptr, const int32_t* colidx, const float* dinv, const float* x,
                         const float* W1, const float* bias, float* ax, float* xout, const float* Wnext,
                         float* hs_next, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
int dg_launch_sortpool_fwd(int N, int B, const int32_t* graph_ptr, const float* x1, const float* x2,
                           const float* x3, const float* x4, float* pooled, int32_t* perm, hipStream_t s);
int dg_launch_sortpool_bwd(int N, int B, const int32_t* graph_ptr, const int32_t* perm, const float* gpooled,
                           float* g1, float* g2, float* g3, float* g4, hipStream_t s);
int dg_launch_readout_fwd(int N, int B, int C, const float* params, const DgParams* pl, const int32_t* graph_ptr,
                          const float* x1, const float* x2, const float* x3, const float* x4, float* pooled,
                          int32_t* perm, float* a5, float* a6, float* a1d, uint8_t* drop_mask, float* logp,
                          int training, uint64_t seed, hipStream_t s, const struct DgPrepRider* rider = nullptr, bool head = true);
// classifier_1 / classifier_2 batched over graphs (classifier.hip); y != nullptr: with the backward from labels
int dg_launch_classifier(int B, int C, const float* params, const DgParams* pl, const float* a6, float* a1d, uint8_t* drop_mask,
                         float* logp, int training, uint64_t seed, const int64_t* y, float loss_scale, float* dlogit,
                         float* gz1, float* gz6, float* lossv, float* ptail, hipStream_t s);
bool dg_classifier_batched(int B);
int dg_launch_fused_fwd(int N, int B, int F, int C, int nmax, int emax, const float* params, const DgParams* pl,
                        const float* x, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                        const int32_t* graph_ptr, const int32_t* graph_eptr, float* ax, float* x1, float* x2, float* x3,
                        float* x4, float* pooled, int32_t* perm, float* a5, float* a6, float* a1d, uint8_t* drop_mask,
                        float* logp, int training, uint64_t seed, int32_t* err, uint32_t epoch, hipStream_t s,
                        hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
int dg_fused_max_nodes(int F);
int dg_fused_fits(int nmax, int emax, int F);
void dg_fused_set_debug(unsigned long long* p);
unsigned long long* dg_debug_buffer();
#define DG_GATHER_UNROLL 8
int dg_launch_tail_bwd(int N, int B, int C, const float* params, const DgParams* pl, const int32_t* graph_ptr,
                       const int32_t* perm, const float* dinv, const float* x4, const float* a5, const float* a6,
                       const float* a1d, const float* logp, const float* glogp, const int64_t* y,
                       float loss_scale, int training, float* dlogit, float* gz1, float* gz6, float* gz5,
                       float* gp1, float* gp2, float* gp3, float* gas4, float* gb4p, float* lossv, float* ptail,
                       const float* pooled, hipStream_t s, const struct DgPrepRider* rider = nullptr, bool head = true,
                       int32_t* gpsel = nullptr);
// large batches behind the batched classifier: a workgroup walks several graphs, ONE partial row per workgroup (tail.hip)
int dg_launch_tail_bwd_walk(int N, int B, int C, const float* params, const DgParams* pl, const int32_t* graph_ptr,
                            const int32_t* perm, const float* dinv, const float* x4, const float* a5, const float* a6,
                            float* gz6, float* gz5, float* gp1, float* gp2, float* gp3, float* gas4, float* gb4p, float* ptail,
                            const float* pooled, hipStream_t s, int32_t* gpsel = nullptr);
int dg_tail_walk_rows(int B);
struct DgAdam {          // optional optimizer step fused into the weight-gradient kernel
  float *params, *exp_avg, *exp_avg_sq;
  float lr, beta1, beta2, eps;
  int64_t step;
};
int dg_launch_wgrad(int which, int N, int B, int F, int C, const DgParams* pl, const DgWs* wl, const void* ws,
                    float* grads, float* metrics, const DgAdam* adam, hipStream_t s, const struct DgPrepRider* rider = nullptr,
                    int tail_rows = 0, int gcn_rows = 0);
int dg_wgrad_takes_rider(int B);
int dg_launch_adam(float* p, float* g, float* m, float* v, int64_t n, int64_t step, float lr, float b1,
                   float b2, float eps, int zero_grads, hipStream_t s);
int dg_launch_metrics(int B, const float* lossv, float* metrics, hipStream_t s);
int dg_launch_eval_metrics(int B, int C, const float* logp, const int64_t* y, float* metrics, float loss_scale, hipStream_t s);
int dg_launch_collate_scan(int B, int F, int64_t N, int64_t E, int64_t Etot, const int64_t* ids_dev, const float* x_all,
                           const int64_t* ei_all, const int64_t* node_ptr, const int64_t* edge_ptr, const int64_t* y_all,
                           float* x, int64_t* ei, int64_t* batch, int64_t* y, hipStream_t s);
int dg_launch_collate(int B, int F, int64_t N, int64_t E, int64_t Etot, const int64_t* ids, const int64_t* onode,
                      const int64_t* oedge, const float* x_all, const int64_t* ei_all, const int64_t* node_ptr,
                      const int64_t* edge_ptr, const int64_t* y_all, float* x, int64_t* ei, int64_t* batch, int64_t* y,
                      hipStream_t s);
