// gcn.hip -- graph-convolution layer kernels, forward and backward (gfx950 / CDNA4).
//
// Replaces PyG GCNConv as the reference calls it (/root/reference/model.py:13-16,30-33) plus the
// torch.tanh around it and the torch.cat of model.py:34 (each layer writes its own [N,32] slab):
//     h      = x W^T                                   (Linear, no bias)      "dense step" -> MFMA
//     out[i] = dinv[i] * ( sum_{j in N_in(i)} dinv[j] h[j] + dinv[i] h[i] ) + b   (gather, no atomics)
//     x'     = tanh(out)
// Data layout in HBM: every per-node activation slab is [N,32] fp32 row-major (128-B rows);
// the linear output is stored PRE-SCALED by the source factor, hs[j] = dinv[j]*h[j], so the
// gather loop needs exactly one index load and one 128-B row load per edge.
//
// Forward kernel shape (F=32): one wavefront per destination node, 16 nodes (16 waves) per
// workgroup.  Lane l = (g = l>>3, q = l&7): neighbour group g (8 neighbours in flight per
// wave-instruction), float4 column chunk q (channels 4q..4q+3) -> every row read is 8 lanes x 16 B
// = one coalesced 128-B line.  The 8 partial sums are combined by a fixed xor-butterfly
// (wavefront segmented reduction, deterministic).  Epilogue: dst scale, bias, tanh, store the
// row, and keep the 16x32 tile in LDS; the NEXT layer's X.W^T is then done on the tile with
// v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain) and stored pre-scaled.
//
// Backward is the same gather on the transposed graph (CSR by source) applied to
// gas[i] = dinv[i] * dL/d(out)[i], followed on the LDS tile by
//     dL/dx_prev = gh . W       (MFMA)            dL/dW += gh^T . x_prev   (MFMA, K = nodes)
// with per-workgroup partial weight gradients reduced later in a fixed order (no fp atomics).
#include "dg_common.h"
#include <hip/hip_ext.h>

// ---------------------------------------------------------------------------------------------
// first linear: hs[i][c] = dinv[i] * sum_k x[i][k] W[c][k]   (x is the raw [N,F] input, F arbitrary)
// FOUT = 32: 8 rows x 32 channels per 256-thread pass.  FOUT = 1: one wave per row.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_lin_first32(int N, int F, const float* __restrict__ x, const float* __restrict__ W,
              const float* __restrict__ dinv, float* __restrict__ hs) {
  extern __shared__ __attribute__((aligned(16))) float Wt[];   // [F][32] (transposed: conflict-free)
  for (int t = threadIdx.x; t < 32 * F; t += blockDim.x) {
    const int c = t / F, k = t - c * F;
    Wt[k * 32 + c] = W[t];
  }
  __syncthreads();
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  for (int i = blockIdx.x * 8 + r; i < N; i += gridDim.x * 8) {
    const float* xr = x + (size_t)i * F;
    float acc = 0.f;
    for (int k = 0; k < F; ++k) acc = fmaf(xr[k], Wt[k * 32 + c], acc);
    hs[(size_t)i * 32 + c] = dinv[i] * acc;
  }
}

__global__ void __launch_bounds__(256)
k_lin_first1(int N, int F, const float* __restrict__ x, const float* __restrict__ W,
             const float* __restrict__ dinv, float* __restrict__ hs) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + w; i < N; i += gridDim.x * 4) {
    const float* xr = x + (size_t)i * F;
    float acc = 0.f;
    for (int k = lane; k < F; k += 64) acc = fmaf(xr[k], W[k], acc);
    acc = dg_wave_sum(acc);
    if (lane == 0) hs[i] = dinv[i] * acc;
  }
}

int dg_launch_lin_first(int N, int F, const float* x, const float* W, const float* dinv, float* hs,
                        int Fout, hipStream_t s) {
  if (N <= 0 || F < 1 || F > DGCNN_MAX_F) return DGCNN_EINVAL;
  if (Fout == 32) {
    int grid = dg_cdiv(N, 8);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_lin_first32, dim3(grid), dim3(256), sizeof(float) * 32 * F, s, N, F, x, W, dinv, hs);
  } else if (Fout == 1) {
    int grid = dg_cdiv(N, 4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_lin_first1, dim3(grid), dim3(256), 0, s, N, F, x, W, dinv, hs);
  } else {
    return DGCNN_EUNSUPPORTED;
  }
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// CANONICAL SUMMATION ORDER (shared by the tiled kernels here and the fused kernels in fused.hip, so
// both paths are bit-identical): every (node, channel) sum is accumulated SEQUENTIALLY over the row's
// neighbours in ascending index order, the self-loop term last -- the order in which the reference's
// CPU scatter_add visits a coalesced edge list with the self loops appended at the end
// (/root/reference/model.py:30-33 via PyG gcn_norm/propagate).
//
// Mapping: half-wave per destination node, lane = channel (32 lanes x 4 B = one 128-B row per
// neighbour, fully coalesced); loads are issued 8 neighbours at a time, adds applied in order.
// No cross-lane reduction is needed at all.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dg_gather_seq32(const float* __restrict__ src, const int* __restrict__ col,
                                                 int start, int end, int self, int c, bool upper) {
  const float acc = dg_coop_gather32<false>(
      start, end, self, c, upper, [&](int e) { return col[e]; },
      [&](int j) { return src[(size_t)j * 32 + c]; });
  return acc + src[(size_t)self * 32 + c];
}

// ---------------------------------------------------------------------------------------------
// forward, F = 32: 32 destination nodes per workgroup (16 waves x 2 half-waves).
//   MODE 0: fused next 32x32 linear on MFMA -> hs_next [N,32]
//   MODE 1: fused next 32->1 linear (dot)   -> hs_next [N]
//   MODE 2: no post-step (stand-alone layer)
// ---------------------------------------------------------------------------------------------
#define DG_NODES_PER_WG 32
template <int MODE>
__global__ void __launch_bounds__(DG_TILE_THREADS)
k_gcn_fwd32(int N, int numTiles, const int* __restrict__ rowptr, const int* __restrict__ colidx,
            const float* __restrict__ dinv, const float* __restrict__ hs, const float* __restrict__ bias,
            float* __restrict__ xout, const float* __restrict__ Wn, float* __restrict__ hs_next) {
  __shared__ __attribute__((aligned(16))) float xt[DG_NODES_PER_WG][DG_LDS_PAD];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  const int slot = wave * 2 + half;              // node slot inside the tile, 0..31

  float wreg[8];
  if (MODE == 0 && wave < 4) {   // B operand: block (rb = wave>>1, nb = wave&1): B[k][n] = Wn[nb*16+n][k]
    const int cc = (wave & 1) * 16 + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wreg[kk] = Wn[cc * 32 + 4 * kk + (lane >> 4)];
  }
  const float wc = (MODE == 1) ? Wn[c] : 0.f;
  const float bc = bias[c];

  for (int tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
    const int i = tile * DG_NODES_PER_WG + slot;
    const bool act = i < N;                    // the cooperative gather needs every lane: no divergence here
    const int ii = act ? i : 0;
    const int start = act ? rowptr[ii] : 0, end = act ? rowptr[ii + 1] : 0;
    const float acc = dg_gather_seq32(hs, colidx, start, end, ii, c, half != 0);
    float val = 0.f;
    if (act) {
      val = tanhf(fmaf(dinv[i], acc, bc));
      xout[(size_t)i * 32 + c] = val;
    }
    if (MODE == 1) {     // conv4's linear (32 -> 1): per-channel products, fixed-order half-wave sum
      const float pacc = dg_half_sum(val * wc);
      if (c == 0 && i < N) hs_next[i] = dinv[i] * pacc;
    }
    if (MODE == 0) xt[slot][c] = val;
    if (MODE == 0) __syncthreads();
    if (MODE == 0 && wave < 4) {   // [32 nodes x 32] . Wn^T as four 16x16 blocks
      const int rb = wave >> 1, nb = wave & 1;
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float a = xt[rb * 16 + (lane & 15)][4 * kk + (lane >> 4)];
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[kk], d, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = tile * DG_NODES_PER_WG + rb * 16 + (lane >> 4) * 4 + r;
        if (node < N) hs_next[(size_t)node * 32 + nb * 16 + (lane & 15)] = dinv[node] * d[r];
      }
    }
    if (MODE == 0) __syncthreads();
  }
}

int dg_launch_gcn_fwd32(int mode, int N, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                        const float* hs, const float* bias, float* xout, const float* Wnext, float* hs_next,
                        hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (N <= 0) return DGCNN_EINVAL;
  const int tiles = dg_cdiv(N, DG_NODES_PER_WG);
  const int grid = tiles > 8192 ? 8192 : tiles;
  // hipExtLaunchKernelGGL attaches the events to THIS dispatch (its own start/end timestamps, the
  // same ones rocprofv3 reports); with null events it is a plain launch.
  if (mode == 0)
    hipExtLaunchKernelGGL(k_gcn_fwd32<0>, dim3(grid), dim3(DG_TILE_THREADS), 0, s, ev_start, ev_stop, 0, N, tiles,
                          rowptr, colidx, dinv, hs, bias, xout, Wnext, hs_next);
  else if (mode == 1)
    hipExtLaunchKernelGGL(k_gcn_fwd32<1>, dim3(grid), dim3(DG_TILE_THREADS), 0, s, ev_start, ev_stop, 0, N, tiles,
                          rowptr, colidx, dinv, hs, bias, xout, Wnext, hs_next);
  else
    hipExtLaunchKernelGGL(k_gcn_fwd32<2>, dim3(grid), dim3(DG_TILE_THREADS), 0, s, ev_start, ev_stop, 0, N, tiles,
                          rowptr, colidx, dinv, hs, bias, xout, Wnext, hs_next);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// forward, F = 1 (conv4): wave per node, lanes across neighbours.
// ---------------------------------------------------------------------------------------------
// F = 1 (conv4): thread per node, sequential sum over ascending neighbours, self last (canonical order)
// half-wave per node; every lane of the half returns the node's sum
__device__ __forceinline__ float dg_gather_seq1(const float* __restrict__ src, const int* __restrict__ col,
                                                int start, int end, int self, int c, bool upper) {
  const float s = dg_coop_gather1(start, end, c, upper, [&](int e) { return col[e]; },
                                  [&](int j) { return src[j]; });
  return s + src[self];
}

__global__ void __launch_bounds__(256)
k_gcn_fwd1(int N, const int* __restrict__ rowptr, const int* __restrict__ colidx,
           const float* __restrict__ dinv, const float* __restrict__ h4s, const float* __restrict__ bias,
           float* __restrict__ x4) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & 31, slot = w * 2 + (lane >> 5);
  const float b = bias[0];
  for (int base = blockIdx.x * 8; base < N; base += gridDim.x * 8) {     // uniform trip count
    const int i = base + slot;
    const bool act = i < N;
    const int ii = act ? i : 0;
    const float s = dg_gather_seq1(h4s, colidx, act ? rowptr[ii] : 0, act ? rowptr[ii + 1] : 0, ii, c, lane >= 32);
    if (act && c == 0) x4[i] = tanhf(fmaf(dinv[i], s, b));
  }
}

int dg_launch_gcn_fwd1(int N, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                       const float* h4s, const float* bias, float* x4, hipStream_t s) {
  if (N <= 0) return DGCNN_EINVAL;
  int grid = dg_cdiv(N, 8);
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(k_gcn_fwd1, dim3(grid), dim3(256), 0, s, N, rowptr, colidx, dinv, h4s, bias, x4);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// backward of conv4 (F_out = 1) fused with the start of conv3's backward:
//   gh4[j]  = dinv[j] * ( sum_{i in N_out(j)} gas4[i] + gas4[j] )            (scalar per node)
//   gx3[j]  = gh4[j] * W4 + gp3[j]          (gp3 = SortPooling gradient slab of layer 3)
//   ga3[j]  = gx3[j] * (1 - x3[j]^2)        gas3[j] = dinv[j] * ga3[j]
//   partials: dW4 += gh4[j] * x3[j]   (32)  ,  db3 += ga3[j]   (32)
// wave per node; lanes 0..31 = channel.  pa4[P1][64] = per-workgroup partial {dW4, db3}.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gcn_bwd1(int N, const int* __restrict__ rowptr_t, const int* __restrict__ colidx_t,
           const float* __restrict__ dinv, const float* __restrict__ gas4, const float* __restrict__ W4,
           const float* __restrict__ x3, const float* __restrict__ gp3, float* __restrict__ gas3,
           float* __restrict__ pa4) {
  __shared__ float red[8][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & 31, slot = w * 2 + (lane >> 5);      // half-wave per node, lane = channel
  const float w4c = W4[c];
  float pW = 0.f, pb = 0.f;
  for (int base = blockIdx.x * 8; base < N; base += gridDim.x * 8) {     // uniform trip count
    const int j = base + slot;
    const bool act = j < N;
    const int jj = act ? j : 0;
    const float s = dg_gather_seq1(gas4, colidx_t, act ? rowptr_t[jj] : 0, act ? rowptr_t[jj + 1] : 0, jj, c,
                                   lane >= 32);
    if (act) {
      const float dj = dinv[j];
      const float gh = dj * s;
      const float xv = x3[(size_t)j * 32 + c];
      const float gx = fmaf(gh, w4c, gp3[(size_t)j * 32 + c]);
      const float ga = gx * (1.f - xv * xv);
      gas3[(size_t)j * 32 + c] = dj * ga;
      pW = fmaf(gh, xv, pW);
      pb += ga;
    }
  }
  red[slot][c] = pW; red[slot][32 + c] = pb;
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[k][threadIdx.x];     // fixed order
    pa4[(size_t)blockIdx.x * 64 + threadIdx.x] = v;
  }
}

int dg_launch_gcn_bwd1(int N, const int32_t* rowptr_t, const int32_t* colidx_t, const float* dinv,
                       const float* gas4, const float* W4, const float* x3, const float* gp3,
                       float* gas3, float* pa4, int P1, hipStream_t s) {
  if (N <= 0 || P1 <= 0) return DGCNN_EINVAL;
  hipLaunchKernelGGL(k_gcn_bwd1, dim3(P1), dim3(256), 0, s, N, rowptr_t, colidx_t, dinv, gas4, W4, x3, gp3, gas3,
                     pa4);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// backward of a 32-wide layer l (l = 3, 2): input gas_l [N,32] (= dinv * dL/d pre-activation)
//   gh[j]      = dinv[j] * ( sum_{i in N_out(j)} gas_l[i] + gas_l[j] )        -> LDS tile [32][32]
//   dW_l      += gh^T . x_{l-1}          (MFMA 16x16x4, K = 32 nodes of the tile; waves 4..7)
//   gx_{l-1}   = gh . W_l + gp_{l-1}     (MFMA; waves 0..3)
//   ga_{l-1}   = gx_{l-1} * (1 - x_{l-1}^2) ; gas_{l-1} = dinv * ga_{l-1} ; db_{l-1} += ga_{l-1}
// part[P][1056] = per-workgroup {dW_l [32x32], db_{l-1} [32]}.
//
// FIRST = true (layer 1): x_{l-1} is the raw input x [N,F]; only dW_1 [32,F] is produced
// (data.x needs no gradient, /root/reference/train.py:36-40).  part[P][32*F].
// ---------------------------------------------------------------------------------------------
template <bool FIRST>
__global__ void __launch_bounds__(DG_TILE_THREADS)
k_gcn_bwd32(int N, int F, int numTiles, const int* __restrict__ rowptr_t, const int* __restrict__ colidx_t,
            const float* __restrict__ dinv, const float* __restrict__ gas, const float* __restrict__ Wl,
            const float* __restrict__ xprev, const float* __restrict__ gpprev, float* __restrict__ gas_prev,
            float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float ght[DG_NODES_PER_WG][DG_LDS_PAD];
  __shared__ __attribute__((aligned(16))) float xt[DG_NODES_PER_WG][DG_LDS_PAD];
  extern __shared__ __attribute__((aligned(16))) float xs[];   // FIRST: [32][F] raw-input tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, half = lane >> 5;
  const int slot = wave * 2 + half;

  // persistent accumulators
  float wreg[8];
  f32x4 accW = {0.f, 0.f, 0.f, 0.f};
  float pb = 0.f;
  float acc1[(32 * DGCNN_MAX_F) / DG_TILE_THREADS];   // FIRST: 16 outputs per thread max
  if (FIRST) {
#pragma unroll
    for (int u = 0; u < (32 * DGCNN_MAX_F) / DG_TILE_THREADS; ++u) acc1[u] = 0.f;
  } else if (wave < 4) {   // B operand of gx = gh . W_l : block (rb = wave>>1, nb = wave&1): B[k][n] = W_l[k][nb*16+n]
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wreg[kk] = Wl[(4 * kk + (lane >> 4)) * 32 + (wave & 1) * 16 + (lane & 15)];
  }

  for (int tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
    const int j = tile * DG_NODES_PER_WG + slot;
    float gh = 0.f, xv = 0.f;
    {
      const bool act = j < N;                  // cooperative gather: every lane takes part
      const int jj = act ? j : 0;
      const float gsum = dg_gather_seq32(gas, colidx_t, act ? rowptr_t[jj] : 0, act ? rowptr_t[jj + 1] : 0, jj, c,
                                         half != 0);
      if (act) {
        gh = dinv[j] * gsum;
        if (!FIRST) xv = xprev[(size_t)j * 32 + c];
      }
    }
    ght[slot][c] = gh;
    if (!FIRST) xt[slot][c] = xv;
    if (FIRST) {
      for (int k = c; k < F; k += 32) xs[slot * F + k] = j < N ? xprev[(size_t)j * F + k] : 0.f;
    }
    __syncthreads();
    if (FIRST) {
      // dW1[cc][k] += sum_node ght[node][cc] * xs[node][k] ; output o = k*32 + cc (cc fastest -> conflict-free)
      const int total = 32 * F;
#pragma unroll
      for (int u = 0; u < (32 * DGCNN_MAX_F) / DG_TILE_THREADS; ++u) {
        const int o = u * DG_TILE_THREADS + threadIdx.x;
        if (o < total) {
          const int k = o >> 5, cc = o & 31;
          float a = acc1[u];
#pragma unroll 8
          for (int nd = 0; nd < DG_NODES_PER_WG; ++nd) a = fmaf(ght[nd][cc], xs[nd * F + k], a);
          acc1[u] = a;
        }
      }
    } else {
      if (wave < 4) {
        const int rb = wave >> 1, nb = wave & 1;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float a = ght[rb * 16 + (lane & 15)][4 * kk + (lane >> 4)];
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[kk], d, 0, 0, 0);
        }
        const int cc = nb * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rb * 16 + (lane >> 4) * 4 + r;
          const int node = tile * DG_NODES_PER_WG + row;
          if (node < N) {
            const float x_ = xt[row][cc];
            const float gx = d[r] + gpprev[(size_t)node * 32 + cc];
            const float ga = gx * (1.f - x_ * x_);
            gas_prev[(size_t)node * 32 + cc] = dinv[node] * ga;
            pb += ga;
          }
        }
      } else if (wave < 8) {   // dW block (mb, nb): A[m][k] = ght[k][mb*16+m], B[k][n] = xt[k][nb*16+n], K = 32 nodes
        const int mb = (wave - 4) >> 1, nb = (wave - 4) & 1;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int k = 4 * kk + (lane >> 4);
          const float a = ght[k][mb * 16 + (lane & 15)];
          const float bq = xt[k][nb * 16 + (lane & 15)];
          accW = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, accW, 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // write this workgroup's partials
  if (FIRST) {
    const int total = 32 * F;
    float* dst = part + (size_t)blockIdx.x * total;
#pragma unroll
    for (int u = 0; u < (32 * DGCNN_MAX_F) / DG_TILE_THREADS; ++u) {
      const int o = u * DG_TILE_THREADS + threadIdx.x;
      if (o < total) {
        const int k = o >> 5, cc = o & 31;
        dst[cc * F + k] = acc1[u];     // stored in W1's own [32,F] layout
      }
    }
  } else {
    // db partial: waves 0..3 hold, per lane, the column sum over their rows; (rb=0,nb) and (rb=1,nb) share
    // columns -> combine through LDS in a fixed order
    __shared__ float pbs[4][16];
    if (wave < 4) {
      pb += __shfl_xor(pb, 16);
      pb += __shfl_xor(pb, 32);
      if (lane < 16) pbs[wave][lane] = pb;
    }
    __syncthreads();
    float* dst = part + (size_t)blockIdx.x * 1056;
    if (threadIdx.x < 32) {
      const int nb = threadIdx.x >> 4, l = threadIdx.x & 15;
      dst[1024 + threadIdx.x] = pbs[nb][l] + pbs[2 + nb][l];     // rb = 0 block + rb = 1 block
    }
    if (wave >= 4 && wave < 8) {
      const int mb = (wave - 4) >> 1, nb = (wave - 4) & 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mb * 16 + (lane >> 4) * 4 + r;     // output channel of W_l
        const int col = nb * 16 + (lane & 15);             // input channel
        dst[row * 32 + col] = accW[r];
      }
    }
  }
}

int dg_launch_gcn_bwd32(int first, int N, int F, const int32_t* rowptr_t, const int32_t* colidx_t,
                        const float* dinv, const float* gas, const float* Wl, const float* xprev,
                        const float* gpprev, float* gas_prev, float* part, int P32, hipStream_t s) {
  if (N <= 0 || P32 <= 0) return DGCNN_EINVAL;
  const int tiles = dg_cdiv(N, DG_NODES_PER_WG);
  if (first) {
    if (F < 1 || F > DGCNN_MAX_F) return DGCNN_EINVAL;
    hipLaunchKernelGGL(k_gcn_bwd32<true>, dim3(P32), dim3(DG_TILE_THREADS), sizeof(float) * DG_NODES_PER_WG * F, s, N, F,
                       tiles, rowptr_t, colidx_t, dinv, gas, Wl, xprev, gpprev, gas_prev, part);
  } else {
    hipLaunchKernelGGL(k_gcn_bwd32<false>, dim3(P32), dim3(DG_TILE_THREADS), 0, s, N, 32, tiles, rowptr_t, colidx_t,
                       dinv, gas, Wl, xprev, gpprev, gas_prev, part);
  }
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}
