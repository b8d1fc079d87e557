// dg_wgrad.h -- weight-gradient reductions (+ optional fused Adam): device body shared by the stand-alone
// k_wgrad launch (tail.hip) and the "rider" block ranges of the GCN backward kernels (gcn.hip), and the
// host-side segment builder.
#pragma once
#include "dg_common.h"
#include <math.h>
#include <stdlib.h>
#define KCAT_W (DGCNN_K * DGCNN_CAT)

// ---------------------------------------------------------------------------------------------
// weight gradients.  out[i] = sum_{r<R} term(i, r): every output is owned by a group of LPO
// consecutive lanes (LPO in {1,8,16,64}) that stride over the reduction index r and then combine
// with a fixed xor-butterfly -> no floating-point atomics, bit-reproducible, and the long
// reductions (bias sums over B*30 terms, per-workgroup partials) are no longer serial chains.
// ---------------------------------------------------------------------------------------------
enum { WG_REDUCE = 0, WG_FC2W, WG_FC2B, WG_FC1W, WG_FC1B, WG_C6W, WG_C6B, WG_C5W, WG_C5B, WG_SUMB, WG_METRIC, WG_FC1W_MFMA };
#define WG_MAX_SEG 20
struct WgSeg {
  int type;
  int count;         // number of outputs
  int lpo;           // lanes per output
  int R;             // reduction length
  int block0;        // first block of this segment
  int stride;        // WG_REDUCE: floats between partial slots ; WG_METRIC: 2
  const float* src;  // WG_REDUCE / WG_SUMB / WG_METRIC source
  float* out;
};
struct WgArgs {
  int nseg, B, C;
  const float *dlogit, *a1d, *gz1, *a6, *gz6, *a5, *gz5, *pooled;
  // optional fused Adam (torch.optim.Adam defaults semantics): applied by the lane that owns the output
  float *adam_p, *adam_m, *adam_v;     // flat buffers (same layout as grads); null = no optimizer step here
  const float* grads_base;             // to turn an output pointer into a flat index
  float lr, b1, b2, eps, bc1, bc2_sqrt;
  WgSeg seg[WG_MAX_SEG];
};

__device__ __forceinline__ float dg_wg_term(const WgArgs& A, const WgSeg& sg, int i, int r) {
  const int C = A.C;
  switch (sg.type) {
    case WG_REDUCE: return sg.src[(size_t)r * sg.stride + i];
    case WG_SUMB:   return sg.src[r];
    case WG_METRIC: return sg.src[(size_t)r * 2 + i];
    case WG_FC2W: { const int c = i / DGCNN_HID1, j = i - c * DGCNN_HID1;
                    return A.dlogit[(size_t)r * C + c] * A.a1d[(size_t)r * DGCNN_HID1 + j]; }
    case WG_FC2B:   return A.dlogit[(size_t)r * C + i];
    case WG_FC1W: { const int j = i / DGCNN_FLAT, m = i - j * DGCNN_FLAT;
                    return A.gz1[(size_t)r * DGCNN_HID1 + j] * A.a6[(size_t)r * DGCNN_FLAT + m]; }
    case WG_FC1B:   return A.gz1[(size_t)r * DGCNN_HID1 + i];
    case WG_C6W: {  // i = (oc*16 + c)*5 + d ; r = b*11 + t
      const int d = i % DGCNN_KW6, c = (i / DGCNN_KW6) % DGCNN_C5, oc = i / (DGCNN_KW6 * DGCNN_C5);
      const int b = r / DGCNN_T6, t = r - b * DGCNN_T6;
      const float* a5b = A.a5 + (size_t)b * (DGCNN_C5 * DGCNN_K) + c * DGCNN_K + 2 * (t + d);
      return A.gz6[(size_t)b * DGCNN_FLAT + oc * DGCNN_T6 + t] * fmaxf(a5b[0], a5b[1]); }
    case WG_C6B: {  const int b = r / DGCNN_T6, t = r - b * DGCNN_T6;
                    return A.gz6[(size_t)b * DGCNN_FLAT + i * DGCNN_T6 + t]; }
    case WG_C5W: {  // i = o*97 + m ; r = b*30 + s
      const int o = i / DGCNN_CAT, m = i - o * DGCNN_CAT;
      const int b = r / DGCNN_K, s = r - b * DGCNN_K;
      return A.gz5[(size_t)b * (DGCNN_C5 * DGCNN_K) + o * DGCNN_K + s] * A.pooled[(size_t)b * KCAT_W + s * DGCNN_CAT + m]; }
    case WG_C5B: {  const int b = r / DGCNN_K, s = r - b * DGCNN_K;
                    return A.gz5[(size_t)b * (DGCNN_C5 * DGCNN_K) + i * DGCNN_K + s]; }
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
__device__ __forceinline__ void dg_wg_fc1w_mfma(const WgArgs& A, const WgSeg& sg, int tile, int lane) {
  const int rb = tile / 22, cb = tile - rb * 22;
  const int B = A.B;
  const int jr = rb * 16 + (lane & 15), mc = cb * 16 + (lane & 15), kq = lane >> 4;
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < B; k0 += 32) {          // 8 MFMAs (32 graphs) per round: 16 loads in flight per lane
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int bb = k0 + 4 * u + kq;
      av[u] = bb < B ? A.gz1[(size_t)bb * DGCNN_HID1 + jr] : 0.f;
      bv[u] = bb < B ? A.a6[(size_t)bb * DGCNN_FLAT + mc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + 4 * u < B) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = rb * 16 + kq * 4 + r, m = cb * 16 + (lane & 15);
    dg_wg_store(A, sg, j * DGCNN_FLAT + m, d[r]);
  }
}

// body of the weight-gradient kernel for one 256-thread (virtual) block; callable from other kernels ("riders")
__device__ __forceinline__ void dg_wgrad_body(const WgArgs& A, int vblock, int vtid) {
  int si = 0;
  for (int k = 1; k < A.nseg; ++k) if (vblock >= A.seg[k].block0) si = k;
  const WgSeg sg = A.seg[si];
  if (sg.type == WG_FC1W_MFMA) {      // block-uniform branch: 4 waves = 4 tiles per workgroup
    const int tile = (vblock - sg.block0) * 4 + (vtid >> 6);
    if (tile < 8 * 22) dg_wg_fc1w_mfma(A, sg, tile, vtid & 63);
    return;
  }
  const int gid = (vblock - sg.block0) * 256 + vtid;
  const int lpo = sg.lpo;
  const int i = gid / lpo, r0 = gid - i * lpo;
  const bool live = i < sg.count;
  // 16 independent accumulators -> 16 loads in flight per lane (one memory round trip for the usual
  // reduction lengths); combined in a fixed order
  float a[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) a[u] = 0.f;
  if (live) {
    const int R = sg.R;
    int r = r0;
    for (; r + 15 * lpo < R; r += 16 * lpo) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] += dg_wg_term(A, sg, i, r + u * lpo);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (r + u * lpo < R) a[u] += dg_wg_term(A, sg, i, r + u * lpo);
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


// ---------------------------------------------------------------------------------------------
// host: build the segment list for a set of stages.  Stage bits (each stage's inputs are complete after
// the kernel named in brackets, so the stage may ride on any LATER launch):
//   1  tail parameters + db4 + loss/accuracy bookkeeping   [k_tail_bwd]
//   2  dW4, db3        (partials pa4)                        [k_gcn_bwd1]
//   4  dW3, db2        (partials pb3)                        [k_gcn_bwd32, layer 3]
//   8  dW2, db1        (partials pb2)                        [k_gcn_bwd32, layer 2]
//  16  dW1             (partials pb1)                        [k_gcn_bwd32, layer 1]
// Returns the number of 256-thread (virtual) blocks.
// ---------------------------------------------------------------------------------------------
static inline int dg_wgrad_build(WgArgs* out, int which, int B, int F, int C, const DgParams* pl, const DgWs* wl,
                                 const void* ws, float* grads, float* metrics, const DgAdam* adam) {
  WgArgs& A = *out;
  memset(&A, 0, sizeof(A));
  A.B = B; A.C = C;
  A.dlogit = dg_cptr<float>(ws, wl->dlogit); A.a1d = dg_cptr<float>(ws, wl->a1d);
  A.gz1 = dg_cptr<float>(ws, wl->gz1); A.a6 = dg_cptr<float>(ws, wl->a6);
  A.gz6 = dg_cptr<float>(ws, wl->gz6); A.a5 = dg_cptr<float>(ws, wl->a5);
  A.gz5 = dg_cptr<float>(ws, wl->gz5); A.pooled = dg_cptr<float>(ws, wl->pooled);
  A.grads_base = grads;
  if (adam && adam->params) {
    A.adam_p = adam->params; A.adam_m = adam->exp_avg; A.adam_v = adam->exp_avg_sq;
    A.lr = adam->lr; A.b1 = adam->beta1; A.b2 = adam->beta2; A.eps = adam->eps;
    A.bc1 = (float)(1.0 - pow((double)adam->beta1, (double)adam->step));
    A.bc2_sqrt = (float)sqrt(1.0 - pow((double)adam->beta2, (double)adam->step));
  }
  int nb = 0, ns = 0;
  auto add = [&](int type, int count, int lpo, int R, float* o, const float* src, int stride) {
    WgSeg& g = A.seg[ns++];
    g.type = type; g.count = count; g.lpo = lpo; g.R = R; g.block0 = nb; g.stride = stride; g.src = src; g.out = o;
    nb += dg_cdiv(count * lpo, 256);
  };
  auto add_tiles = [&](int type, int tiles, float* o) {     // one wave per tile, 4 tiles per (virtual) block
    WgSeg& g = A.seg[ns++];
    g.type = type; g.count = tiles; g.lpo = 64; g.R = B; g.block0 = nb; g.stride = 0; g.src = nullptr; g.out = o;
    nb += dg_cdiv(tiles, 4);
  };
  const int lp = 16;
  if (which & 2) {
    const float* pa4 = dg_cptr<float>(ws, wl->pa4);
    add(WG_REDUCE, 32, 64, wl->P1, grads + pl->off[5], pa4 + 32, 64);               // db3 (from conv4 backward)
    add(WG_REDUCE, 32, 64, wl->P1, grads + pl->off[6], pa4, 64);                    // dW4
  }
  if (which & 4) {
    const float* pb3 = dg_cptr<float>(ws, wl->pb3);
    add(WG_REDUCE, 1024, lp, wl->P32, grads + pl->off[4], pb3, 1056);               // dW3
    add(WG_REDUCE, 32, 64, wl->P32, grads + pl->off[3], pb3 + 1024, 1056);          // db2 (from layer-3 backward)
  }
  if (which & 8) {
    const float* pb2 = dg_cptr<float>(ws, wl->pb2);
    add(WG_REDUCE, 1024, lp, wl->P32, grads + pl->off[2], pb2, 1056);               // dW2
    add(WG_REDUCE, 32, 64, wl->P32, grads + pl->off[1], pb2 + 1024, 1056);          // db1 (from layer-2 backward)
  }
  if (which & 16) {
    const float* pb1 = dg_cptr<float>(ws, wl->pb1);
    add(WG_REDUCE, 32 * F, lp, wl->P32, grads + pl->off[0], pb1, 32 * F);           // dW1
  }
  if (which & 1) {
    const bool small = B <= 128;
    add_tiles(WG_FC1W_MFMA, 8 * 22, grads + pl->off[12]);                           // classifier_1 weight: MFMA GEMM
    // conv5 / conv6 / classifier_2: k_tail_bwd left one partial per graph; sum B contiguous partials per element
    const float* pt = dg_cptr<float>(ws, wl->ptail);
    const int st = DG_PTAIL(C);
    const int lpr = small ? 8 : 1;
    add(WG_REDUCE, DGCNN_C5 * DGCNN_CAT, lpr, B, grads + pl->off[8], pt + DG_PT_W5, st);
    add(WG_REDUCE, DGCNN_C5, 64, B, grads + pl->off[9], pt + DG_PT_B5, st);
    add(WG_REDUCE, DGCNN_C6 * DGCNN_C5 * DGCNN_KW6, lpr, B, grads + pl->off[10], pt + DG_PT_W6, st);
    add(WG_REDUCE, DGCNN_C6, 64, B, grads + pl->off[11], pt + DG_PT_B6, st);
    add(WG_REDUCE, C * DGCNN_HID1, lpr, B, grads + pl->off[14], pt + DG_PT_WF2, st);
    add(WG_REDUCE, C, 64, B, grads + pl->off[15], pt + DG_PT_WF2 + C * DGCNN_HID1, st);
    add(WG_FC1B, DGCNN_HID1, 64, B, grads + pl->off[13], nullptr, 0);
    add(WG_SUMB, 1, 64, B, grads + pl->off[7], dg_cptr<float>(ws, wl->gb4p), 0);    // db4
    if (metrics) add(WG_METRIC, 2, 64, B, metrics, dg_cptr<float>(ws, wl->lossv), 2);   // train.py:44-45 bookkeeping
  }
  A.nseg = ns;
  return nb;
}
