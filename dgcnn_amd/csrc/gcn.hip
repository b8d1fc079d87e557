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
#include "dg_prep.h"
#include <hip/hip_ext.h>
#include <cstdlib>
// Gather depth = row loads in flight per wavefront.  Few tiles (the reference's batch of 50: one workgroup per CU,
// latency-bound): deepest batching, 8 -> one memory round trip per 64 neighbours.  Many tiles (large batches,
// throughput-bound): shallower batching keeps the VGPR count low enough for two 1024-thread workgroups per CU,
// whose gathers then hide each other's epilogues (measured: profiles/r01_sweep.txt).
#define DG_DEPTH_SMALL 8
#ifndef DG_DEPTH_FWD_BIG
#define DG_DEPTH_FWD_BIG 4
#endif
#ifndef DG_DEPTH_BWD_BIG
#define DG_DEPTH_BWD_BIG 2
#endif
#ifndef DG_SMALL_GRID_TILES
#define DG_SMALL_GRID_TILES 256       // = number of CUs: up to one workgroup per CU the deep form wins, beyond it the
#endif                                // two-workgroups-per-CU forms do (measured at 232 / 300 / 470 tiles)
#define DG_PERSIST_WGS 512           // persistent large-grid kernels: 2 workgroups per CU
#ifndef DG_PERSIST_MIN_TILES
#define DG_PERSIST_MIN_TILES (4 * DG_PERSIST_WGS)
#endif

// ---------------------------------------------------------------------------------------------
// first linear: hs[i][c] = dinv[i] * sum_k x[i][k] W[c][k]   (x is the raw [N,F] input, F arbitrary)
// FOUT = 32: 8 rows x 32 channels per 256-thread pass.  FOUT = 1: one wave per row.
// ---------------------------------------------------------------------------------------------
// FOUT = 32 runs on the fp32 matrix cores: one wave per 16-row tile, two 16x16 output blocks, K = F in steps of 4
// (v_mfma_f32_16x16x4_f32, operands of 8 k-steps fetched together); W^T staged in LDS.  The fused kernel's conv1
// linear issues the same MFMA sequence on the same operands, so both paths stay bit-identical.
template <bool BF>       // BF: the pre-scaled output is stored in bf16 (round to nearest even): the bf16 leg's storage format
__global__ void __launch_bounds__(256)
k_lin_first32(int N, int F, const float* __restrict__ x, const float* __restrict__ W,
              const float* __restrict__ dinv, void* __restrict__ hsv) {
  float* hs = reinterpret_cast<float*>(hsv);
  unsigned short* hb = reinterpret_cast<unsigned short*>(hsv);
  DG_DYN_SMEM(float, Wt);   // [F][32] (transposed: conflict-free)
  for (int t = threadIdx.x; t < 32 * F; t += blockDim.x) {
    const int c = t / F, k = t - c * F;
    Wt[k * 32 + c] = W[t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tiles = (N + 15) >> 4;
  for (int tile = blockIdx.x * 4 + w; tile < tiles; tile += gridDim.x * 4) {
    const int r0 = tile * 16;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
      dg_mfma_tile16(
          r0, nb * 16, F, lane,
          [&](int m, int k) { return (m < N && k < F) ? x[(size_t)m * F + k] : 0.f; },
          [&](int k, int n) { return k < F ? Wt[k * 32 + n] : 0.f; },
          [&](int m, int n, float v) {
            if (m < N) {
              const float o = dinv[m] * v;
              if (BF) { unsigned u = __float_as_uint(o); u += 0x7fffu + ((u >> 16) & 1u); hb[(size_t)m * 32 + n] = (unsigned short)(u >> 16); }
              else hs[(size_t)m * 32 + n] = o;
            }
          });
  }
}

// Staged form (F <= DG_LIN_STAGE_MAX_F, 16-byte aligned x and W): a 16-row tile of the raw input is ONE contiguous run of
// 16 F floats whose start is 64-byte aligned (tile row = multiple of 16), so a wave brings it in with 16-byte loads that
// cover whole cache lines and keeps it in a wave-private LDS tile; W stays in its own [32][F] layout in LDS (a flat copy:
// the transposing copy above writes one bank 64 times per instruction).  The form above reads the A operand straight from
// global memory -- one dword per lane, 16 different lines per instruction, and again for the second output block: at DD's
// 14.5 k x 90 input 10.9 us.  Row stride F in both tiles: the 16 rows of an operand read fall on 16 different banks for every
// F that is 2 mod 4 or odd (for the others the reads are 2- to 4-way conflicted: still cheaper than the global form).
// SAME matrix-instruction sequence on the SAME operand values as the form above (and as the fused kernels): bit-identical.
#define DG_LIN_STAGE_MAX_F 128
template <bool BF>
__global__ void __launch_bounds__(256)
k_lin_first32s(int N, int F, const float* __restrict__ x, const float* __restrict__ W,
               const float* __restrict__ dinv, void* __restrict__ hsv) {
  float* hs = reinterpret_cast<float*>(hsv);
  unsigned short* hb = reinterpret_cast<unsigned short*>(hsv);
  DG_DYN_SMEM(float, lsm);     // [32][F] weights | 4 x [16][F] tiles | 4 floats of slack
  float* Ws = lsm;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* xs = lsm + 32 * F + w * 16 * F;                 // this wave's tile (32 F and 16 F floats: multiples of 16 bytes)
  const int tiles = (N + 15) >> 4;
  const int mi = lane & 15, kq = lane >> 4;
  constexpr int XQ = DG_LIN_STAGE_MAX_F * 16 / 4 / 64;   // 16-byte pieces of a tile per lane (8 at the largest F)
  constexpr int WQ = DG_LIN_STAGE_MAX_F * 32 / 4 / 256;  // ... of the weights per thread (4)
  const int nq = 4 * F;                                  // 16-byte pieces of a full tile
  int tile = blockIdx.x * 4 + w;
  float4 v[XQ], wv[WQ];
  float dv[4];
  // (a macro, not a lambda: captured by reference the register arrays stay in scratch)
#define DG_LS_LOAD_TILE()                                                                                               \
  do {                                                                                                                  \
    const int r0_ = tile * 16;                                                                                          \
    const int nfl = tile < tiles ? min(16, N - r0_) * F : 0;      /* floats of this tile that exist */                  \
    const float* xt = x + (size_t)r0_ * F;                                                                              \
    /* nfl is wave-uniform.  A multiple of 4 (every full tile): no 16-byte piece straddles the end, pieces beyond it load      \
       nothing.  Otherwise (the batch's last tile): dword loads, every one UNCONDITIONAL on a clamped index and selected --     \
       under the lane-divergent test of the first version the straddling piece was waited for on the spot, one exposed round   \
       trip on the one wave whose end is the kernel's end */                                                                   \
    if ((nfl & 3) == 0) {                                                                                                      \
      _Pragma("unroll") for (int i = 0; i < XQ; ++i) {                                                                         \
        const int p = lane + 64 * i;                                                                                           \
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                                \
        if (p < nq && 4 * p + 3 < nfl) v[i] = *reinterpret_cast<const float4*>(xt + 4 * p);                                    \
      }                                                                                                                        \
    } else {                                                                                                                   \
      _Pragma("unroll") for (int i = 0; i < XQ; ++i) {                                                                         \
        const int e = 4 * (lane + 64 * i);                                                                                     \
        const float t0 = xt[min(e, nfl - 1)], t1 = xt[min(e + 1, nfl - 1)], t2 = xt[min(e + 2, nfl - 1)],                      \
                    t3 = xt[min(e + 3, nfl - 1)];                                                                              \
        v[i] = make_float4(e < nfl ? t0 : 0.f, e + 1 < nfl ? t1 : 0.f, e + 2 < nfl ? t2 : 0.f, e + 3 < nfl ? t3 : 0.f);        \
      }                                                                                                                        \
    }                                                                                                                          \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) { const int m = r0_ + kq * 4 + r; dv[r] = m < N ? dinv[m] : 0.f; }    \
  } while (0)
  // every load of the set-up in flight before the first LDS store
#pragma unroll
  for (int i = 0; i < WQ; ++i) {
    const int p = threadIdx.x + 256 * i;
    wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < 8 * F) wv[i] = *reinterpret_cast<const float4*>(W + 4 * p);
  }
  DG_LS_LOAD_TILE();
#pragma unroll
  for (int i = 0; i < WQ; ++i) {
    const int p = threadIdx.x + 256 * i;
    if (p < 8 * F) *reinterpret_cast<float4*>(Ws + 4 * p) = wv[i];
  }
  bool first = true;
  while (true) {
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
      const int p = lane + 64 * i;
      if (p < nq) *reinterpret_cast<float4*>(xs + 4 * p) = v[i];
    }
    if (first) { __syncthreads(); first = false; }       // (outside any wave-dependent condition: the weights are staged)
    if (tile >= tiles) break;
    const int r0 = tile * 16;
    // the two 16-column output blocks advance together on one read of the A operand; per accumulator the sequence is
    // dg_mfma_tile16's: 8 k-steps per round, operands beyond F are zeros, steps beyond F are not issued
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < F; k0 += 32) {
      float av[8], b0[8], b1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 4 * u + kq;
        const float a_ = xs[mi * F + k], p_ = Ws[mi * F + k], q_ = Ws[(16 + mi) * F + k];     // (k up to F + 34: inside the LDS block)
        av[u] = k < F ? a_ : 0.f; b0[u] = k < F ? p_ : 0.f; b1[u] = k < F ? q_ : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + 4 * u < F) {
          d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b0[u], d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b1[u], d1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + kq * 4 + r;
      if (row < N) {
        const float o0 = dv[r] * d0[r], o1 = dv[r] * d1[r];
        if (BF) {
          unsigned u0 = __float_as_uint(o0), u1 = __float_as_uint(o1);
          u0 += 0x7fffu + ((u0 >> 16) & 1u); u1 += 0x7fffu + ((u1 >> 16) & 1u);
          hb[(size_t)row * 32 + mi] = (unsigned short)(u0 >> 16); hb[(size_t)row * 32 + 16 + mi] = (unsigned short)(u1 >> 16);
        } else { hs[(size_t)row * 32 + mi] = o0; hs[(size_t)row * 32 + 16 + mi] = o1; }
      }
    }
    tile += gridDim.x * 4;
    if (tile >= tiles) break;
    DG_LS_LOAD_TILE();
  }
#undef DG_LS_LOAD_TILE
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
                        int Fout, hipStream_t s, int bf16_out) {
  if (N <= 0 || F < 1 || F > DGCNN_MAX_F) return DGCNN_EINVAL;
  if (Fout == 32) {
    int grid = dg_cdiv(dg_cdiv(N, 16), 4);
    if (grid > 4096) grid = 4096;
    const bool staged = F <= DG_LIN_STAGE_MAX_F && (((uintptr_t)x | (uintptr_t)W) & 15) == 0;
    const size_t lds_s = sizeof(float) * (96 * F + 32);       // weights + four tiles + the slack the last k-round reads into
    if (staged) {
      if (bf16_out) hipLaunchKernelGGL(k_lin_first32s<true>, dim3(grid), dim3(256), lds_s, s, N, F, x, W, dinv, (void*)hs);
      else hipLaunchKernelGGL(k_lin_first32s<false>, dim3(grid), dim3(256), lds_s, s, N, F, x, W, dinv, (void*)hs);
    } else if (bf16_out) hipLaunchKernelGGL(k_lin_first32<true>, dim3(grid), dim3(256), sizeof(float) * 32 * F, s, N, F, x, W, dinv, (void*)hs);
    else hipLaunchKernelGGL(k_lin_first32<false>, dim3(grid), dim3(256), sizeof(float) * 32 * F, s, N, F, x, W, dinv, (void*)hs);
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
// gather of one destination row: returns (in every lane with g==0, and in fact all lanes) the
// float4 chunk q of   sum_{e in [start,end)} src[col[e]]  +  src[self]
// ---------------------------------------------------------------------------------------------
// HAVE0: the first batch of neighbour ids (col[start + lane], 0 beyond the row) was loaded by the caller (cj0).
template <int DG_GATHER_DEPTH, bool HAVE0 = false>
__device__ __forceinline__ float4 dg_gather_row32(const float* __restrict__ src, const int* __restrict__ col,
                                                  int start, int end, int self, int lane, int cj0 = 0) {
  const int g = lane >> 3, q = lane & 7;
  const float* sq = src + 4 * q;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 vself = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g == 0) vself = *reinterpret_cast<const float4*>(sq + (size_t)self * 32);   // issued first, added last
  for (int base = start; base < end; base += 64) {
    const int cnt = min(64, end - base);
    const int cj = (HAVE0 && base == start) ? cj0 : (lane < cnt ? col[base + lane] : 0);
    // DG_GATHER_DEPTH row loads are issued back to back and only then summed, in the same order as a
    // one-at-a-time loop would: one memory round trip per DEPTH*8 neighbours instead of one per 8
#pragma unroll
    for (int u0 = 0; u0 < 8; u0 += DG_GATHER_DEPTH) {
      if (u0 * 8 >= cnt) break;
      float4 v[DG_GATHER_DEPTH];
      int j[DG_GATHER_DEPTH];
#pragma unroll
      for (int u = 0; u < DG_GATHER_DEPTH; ++u) j[u] = __shfl(cj, (u0 + u) * 8 + g);   // index broadcasts first
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < DG_GATHER_DEPTH; ++u) {
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((u0 + u) * 8 + g < cnt) v[u] = *reinterpret_cast<const float4*>(sq + (size_t)j[u] * 32);
      }
#pragma unroll
      for (int u = 0; u < DG_GATHER_DEPTH; ++u)
        if ((u0 + u) * 8 + g < cnt) acc = dg_add4(acc, v[u]);
    }
  }
  if (g == 0) acc = dg_add4(acc, vself);   // self loop term, added last in group 0 (PyG appends self loops at the end)
  acc = dg_add4(acc, dg_shfl_xor4(acc, 8));
  acc = dg_add4(acc, dg_shfl_xor4(acc, 16));
  acc = dg_add4(acc, dg_shfl_xor4(acc, 32));
  return acc;
}

// ---------------------------------------------------------------------------------------------
// forward, F = 32.  MODE 0: fused next 32x32 linear on MFMA -> hs_next [N,32]
//                   MODE 1: fused next 32->1 linear (dot)   -> hs_next [N]
//                   MODE 2: no post-step (stand-alone layer)
// ---------------------------------------------------------------------------------------------
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(DG_TILE_THREADS)
k_gcn_fwd32(int N, int numTiles, const int* __restrict__ rowptr, const int* __restrict__ colidx,
            const float* __restrict__ dinv, const float* __restrict__ hs, const float* __restrict__ bias,
            float* __restrict__ xout, const float* __restrict__ Wn, float* __restrict__ hs_next) {
  __shared__ __attribute__((aligned(16))) float xt[DG_TILE][DG_LDS_PAD];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 3, q = lane & 7;

  float wreg[8];
  if (MODE == 0 && wave < 2) {   // B operand of the post-step: B[k][n] = Wn[n][k]
    const int c = wave * 16 + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wreg[kk] = Wn[c * 32 + 4 * kk + (lane >> 4)];
  }
  float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 1) w4 = *reinterpret_cast<const float4*>(Wn + 4 * q);
  const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * q);

  for (int tl = blockIdx.x; tl < numTiles; tl += gridDim.x) {
    const int tile = (gridDim.x == (unsigned)numTiles) ? dg_xcd_tile(tl, numTiles) : tl;
    const int i = tile * DG_TILE + wave;
    if (i < N) {
      const int start = __builtin_amdgcn_readfirstlane(rowptr[i]);
      const int end = __builtin_amdgcn_readfirstlane(rowptr[i + 1]);
      const float di = dinv[i];                       // issued before the gather, consumed after it
      const float4 acc = dg_gather_row32<DEPTH>(hs, colidx, start, end, i, lane);
      float4 val;
      val.x = dg_tanh(fmaf(di, acc.x, b4.x));
      val.y = dg_tanh(fmaf(di, acc.y, b4.y));
      val.z = dg_tanh(fmaf(di, acc.z, b4.z));
      val.w = dg_tanh(fmaf(di, acc.w, b4.w));
      if (g == 0) {
        *reinterpret_cast<float4*>(xout + (size_t)i * 32 + 4 * q) = val;
        if (MODE == 0) *reinterpret_cast<float4*>(&xt[wave][4 * q]) = val;
      }
      if (MODE == 1) {
        float p = val.x * w4.x;
        p = fmaf(val.y, w4.y, p);
        p = fmaf(val.z, w4.z, p);
        p = fmaf(val.w, w4.w, p);
        p += __shfl_xor(p, 1);
        p += __shfl_xor(p, 2);
        p += __shfl_xor(p, 4);
        if (lane == 0) hs_next[i] = di * p;
      }
    } else if (MODE == 0 && g == 0) {
      *reinterpret_cast<float4*>(&xt[wave][4 * q]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float dpre[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0 && wave < 2) {      // dst scales of the MFMA epilogue rows: issue the loads before the barrier
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = tile * DG_TILE + (lane >> 4) * 4 + r;
        dpre[r] = node < N ? dinv[node] : 0.f;
      }
    }
    if (MODE == 0) {
      __syncthreads();
      if (wave < 2) {   // [16 nodes x 32] . W^T -> 16x16 block `wave` of the [16 x 32] result
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float a = xt[lane & 15][4 * kk + (lane >> 4)];
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[kk], d, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int node = tile * DG_TILE + (lane >> 4) * 4 + r;
          if (node < N) hs_next[(size_t)node * 32 + wave * 16 + (lane & 15)] = dpre[r] * d[r];
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Large grids: PERSISTENT, software-pipelined form of k_gcn_fwd32.  With one tile per workgroup a wave's life is a
// chain of three dependent round trips (row pointers -> neighbour ids -> neighbour rows) and then it retires; at
// thousands of tiles the chip is latency-bound on that chain.  Here 2 workgroups per CU each walk a contiguous,
// XCD-contiguous chunk of tiles and every load a tile needs except the rows themselves is issued one or two tiles
// AHEAD (row pointers of tile t+2, first 64 ids of tile t+1) -- per tile only the row round trip is exposed.
// Same lane mapping and summation order as k_gcn_fwd32: bit-identical results.
// ---------------------------------------------------------------------------------------------
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(DG_TILE_THREADS)
k_gcn_fwd32p(int N, int numTiles, const int* __restrict__ rowptr, const int* __restrict__ colidx,
             const float* __restrict__ dinv, const float* __restrict__ hs, const float* __restrict__ bias,
             float* __restrict__ xout, const float* __restrict__ Wn, float* __restrict__ hs_next) {
  __shared__ __attribute__((aligned(16))) float xt[DG_TILE][DG_LDS_PAD];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 3, q = lane & 7;

  float wreg[8];
  if (MODE == 0 && wave < 2) {
    const int c = wave * 16 + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wreg[kk] = Wn[c * 32 + 4 * kk + (lane >> 4)];
  }
  float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 1) w4 = *reinterpret_cast<const float4*>(Wn + 4 * q);
  const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * q);

  const int chunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int wg = dg_xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int t0 = wg * chunk, t1 = min(numTiles, t0 + chunk);
  if (t0 >= t1) return;
  // pipeline state: (rs0,re0,cj0) = this tile; (rs1,re1) = next tile's row pointers (vector registers until used)
  int rs0 = 0, re0 = 0, cj0 = 0, rs1v = 0, re1v = 0;
  {
    const int i0 = t0 * DG_TILE + wave, i1 = i0 + DG_TILE;
    if (i0 < N) { rs0 = rowptr[i0]; re0 = rowptr[i0 + 1]; }
    if (t0 + 1 < t1 && i1 < N) { rs1v = rowptr[i1]; re1v = rowptr[i1 + 1]; }
    rs0 = __builtin_amdgcn_readfirstlane(rs0); re0 = __builtin_amdgcn_readfirstlane(re0);
    cj0 = lane < min(64, re0 - rs0) ? colidx[rs0 + lane] : 0;
  }
  for (int tile = t0; tile < t1; ++tile) {
    const int i = tile * DG_TILE + wave;
    // stage A: row pointers of tile + 2
    int rs2v = 0, re2v = 0;
    {
      const int i2 = i + 2 * DG_TILE;
      if (tile + 2 < t1 && i2 < N) { rs2v = rowptr[i2]; re2v = rowptr[i2 + 1]; }
    }
    // stage B: first neighbour ids of tile + 1 (its row pointers were requested one tile ago)
    const int rs1 = __builtin_amdgcn_readfirstlane(rs1v), re1 = __builtin_amdgcn_readfirstlane(re1v);
    const int cj1 = lane < min(64, re1 - rs1) ? colidx[rs1 + lane] : 0;
    // stage C: this tile
    if (i < N) {
      const float di = dinv[i];
      const float4 acc = dg_gather_row32<DEPTH, true>(hs, colidx, rs0, re0, i, lane, cj0);
      float4 val;
      val.x = dg_tanh(fmaf(di, acc.x, b4.x));
      val.y = dg_tanh(fmaf(di, acc.y, b4.y));
      val.z = dg_tanh(fmaf(di, acc.z, b4.z));
      val.w = dg_tanh(fmaf(di, acc.w, b4.w));
      if (g == 0) {
        *reinterpret_cast<float4*>(xout + (size_t)i * 32 + 4 * q) = val;
        if (MODE == 0) *reinterpret_cast<float4*>(&xt[wave][4 * q]) = val;
      }
      if (MODE == 1) {
        float p = val.x * w4.x;
        p = fmaf(val.y, w4.y, p);
        p = fmaf(val.z, w4.z, p);
        p = fmaf(val.w, w4.w, p);
        p += __shfl_xor(p, 1);
        p += __shfl_xor(p, 2);
        p += __shfl_xor(p, 4);
        if (lane == 0) hs_next[i] = di * p;
      }
    } else if (MODE == 0 && g == 0) {
      *reinterpret_cast<float4*>(&xt[wave][4 * q]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float dpre[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0 && wave < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = tile * DG_TILE + (lane >> 4) * 4 + r;
        dpre[r] = node < N ? dinv[node] : 0.f;
      }
    }
    if (MODE == 0) {
      __syncthreads();
      if (wave < 2) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float a = xt[lane & 15][4 * kk + (lane >> 4)];
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[kk], d, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int node = tile * DG_TILE + (lane >> 4) * 4 + r;
          if (node < N) hs_next[(size_t)node * 32 + wave * 16 + (lane & 15)] = dpre[r] * d[r];
        }
      }
      __syncthreads();
    }
    rs0 = rs1; re0 = re1; cj0 = cj1; rs1v = rs2v; re1v = re2v;
  }
}

// ---------------------------------------------------------------------------------------------
// NARROW forms for sparse batches of many nodes (mean in-degree <= DG_NARROW_MAX_DEG: the DD / PROTEINS / ENZYMES / MUTAG
// shapes; DD at the reference's batch of 50 is 14.5 k nodes of degree 5).  A wave per destination node leaves 7 of its 8
// neighbour groups idle at degree 5, and 910 sixteen-wave workgroups are two rounds of a four-round-trip latency chain
// (row pointers -> neighbour ids -> rows -> tile product): 8.9 us per layer.  Here EIGHT LANES own a destination node (lane =
// 8 g + q: node g of the wave's eight, 16-byte column chunk q of its 128-byte row), a wave owns 8 nodes and a 256-thread
// workgroup 32 nodes = two 16-row tiles of the matrix-core epilogue: an eighth of the waves, one round.
// The eight accumulators a[u] (u = position of the neighbour modulo 8) and their combine
//     ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7))
// are dg_gather_row32's eight lane groups and its xor-8 / 16 / 32 butterfly: the SAME fp32 additions in the same order --
// results are bit-identical to the wave-per-node kernels (tests/test_gpu_kernels.py compares the two forms exactly).
// ---------------------------------------------------------------------------------------------
#define DG_NARROW_MAX_DEG 8
static int g_narrow = -1;      // (-1: not yet read; DGCNN_NARROW_GATHER=0|1|2 in the environment sets the initial value, default 2)
static inline int dg_narrow_level() {
  if (g_narrow < 0) { const char* e = getenv("DGCNN_NARROW_GATHER"); g_narrow = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }
  return g_narrow;
}
// 0: wave per node everywhere; 1: the 32-wide narrow forms only (the round-4 / round-5 default); 2 (default since round 6): also the
// scalar narrow forms of conv4's two gathers (k_gcn_fwd1n / k_gcn_bwd1n) -- verified on the CPU emulation, not yet timed on a GPU
int dg_narrow_gather_enable(int on) { const int prev = dg_narrow_level(); g_narrow = on <= 0 ? 0 : (on >= 2 ? 2 : 1); return prev; }
// E: directed edges of the batch without the self loops (< 0: unknown -> the wave-per-node forms)
static inline bool dg_use_narrow(int N, int E) {
  return dg_narrow_level() && E >= 0 && dg_cdiv(N, DG_TILE) > DG_SMALL_GRID_TILES && (int64_t)E <= (int64_t)DG_NARROW_MAX_DEG * N;
}
int dg_narrow_applies(int N, int E) { return dg_use_narrow(N, E) ? 1 : 0; }
__device__ __forceinline__ float4 dg_gather_row32_n(const float* __restrict__ src, const int* __restrict__ col, int start,
                                                    int cnt, int self, bool valid, int lane) {
  const int q = lane & 7, gb = lane & ~7;
  const float* sq = src + 4 * q;
  float4 a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 vself = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) vself = *reinterpret_cast<const float4*>(sq + (size_t)self * 32);      // issued first, added last
  int mx = cnt;                                          // trip count: the longest row among the wave's eight nodes
  mx = max(mx, __shfl_xor(mx, 8)); mx = max(mx, __shfl_xor(mx, 16)); mx = max(mx, __shfl_xor(mx, 32));
  mx = __builtin_amdgcn_readfirstlane(mx);
  for (int base = 0; base < mx; base += 8) {
    const int cj = base + q < cnt ? col[start + base + q] : 0;      // the node's next eight neighbour ids: 32 contiguous bytes
    int j[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) j[u] = __shfl(cj, gb + u);
    __builtin_amdgcn_sched_barrier(0);
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                          // eight row loads in flight per lane
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (base + u < cnt) v[u] = *reinterpret_cast<const float4*>(sq + (size_t)j[u] * 32);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + u < cnt) a[u] = dg_add4(a[u], v[u]);
  }
  if (valid) a[0] = dg_add4(a[0], vself);                // self loop term, added last in group 0 (PyG appends self loops at the end)
  return dg_add4(dg_add4(dg_add4(a[0], a[1]), dg_add4(a[2], a[3])), dg_add4(dg_add4(a[4], a[5]), dg_add4(a[6], a[7])));
}

#define DG_NB 32                   // destination nodes per 256-thread workgroup of the narrow forms
template <int MODE>
__global__ void __launch_bounds__(256)
k_gcn_fwd32n(int N, int numBlocks, const int* __restrict__ rowptr, const int* __restrict__ colidx,
             const float* __restrict__ dinv, const float* __restrict__ hs, const float* __restrict__ bias,
             float* __restrict__ xout, const float* __restrict__ Wn, float* __restrict__ hs_next) {
  __shared__ __attribute__((aligned(16))) float xt[DG_NB][DG_LDS_PAD];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 3, q = lane & 7;
  const int tt = wave >> 1, nb = wave & 1;                 // epilogue: 16-row tile tt of the block, 16-column output block nb
  float wreg[8];
  if (MODE == 0) {   // B operand of the post-step: B[k][n] = Wn[n][k]
    const int c = nb * 16 + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wreg[kk] = Wn[c * 32 + 4 * kk + (lane >> 4)];
  }
  float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 1) w4 = *reinterpret_cast<const float4*>(Wn + 4 * q);
  const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * q);
  for (int bl = blockIdx.x; bl < numBlocks; bl += gridDim.x) {
    const int blk = (gridDim.x == (unsigned)numBlocks) ? dg_xcd_tile(bl, numBlocks) : bl;
    const int i = blk * DG_NB + wave * 8 + g;
    const bool valid = i < N;
    int start = 0, end = 0;
    float di = 0.f;
    if (valid) { start = rowptr[i]; end = rowptr[i + 1]; di = dinv[i]; }
    float dpre[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {       // dst scales of the epilogue rows: issued with the row pointers
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = blk * DG_NB + tt * 16 + (lane >> 4) * 4 + r;
        dpre[r] = node < N ? dinv[node] : 0.f;
      }
    }
    const float4 acc = dg_gather_row32_n(hs, colidx, start, end - start, i, valid, lane);
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      val.x = dg_tanh(fmaf(di, acc.x, b4.x));
      val.y = dg_tanh(fmaf(di, acc.y, b4.y));
      val.z = dg_tanh(fmaf(di, acc.z, b4.z));
      val.w = dg_tanh(fmaf(di, acc.w, b4.w));
      *reinterpret_cast<float4*>(xout + (size_t)i * 32 + 4 * q) = val;
    }
    if (MODE == 0) *reinterpret_cast<float4*>(&xt[wave * 8 + g][4 * q]) = val;
    if (MODE == 1) {
      float p = val.x * w4.x;
      p = fmaf(val.y, w4.y, p);
      p = fmaf(val.z, w4.z, p);
      p = fmaf(val.w, w4.w, p);
      p += __shfl_xor(p, 1);
      p += __shfl_xor(p, 2);
      p += __shfl_xor(p, 4);
      if (valid && q == 0) hs_next[i] = di * p;
    }
    if (MODE == 0) {
      __syncthreads();
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float a = xt[tt * 16 + (lane & 15)][4 * kk + (lane >> 4)];
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[kk], d, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = blk * DG_NB + tt * 16 + (lane >> 4) * 4 + r;
        if (node < N) hs_next[(size_t)node * 32 + nb * 16 + (lane & 15)] = dpre[r] * d[r];
      }
      if (bl + (int)gridDim.x < numBlocks) __syncthreads();
    }
  }
}

int dg_launch_gcn_fwd32(int mode, int N, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                        const float* hs, const float* bias, float* xout, const float* Wnext, float* hs_next,
                        hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, int E) {
  if (N <= 0) return DGCNN_EINVAL;
  const int tiles = dg_cdiv(N, DG_TILE);
  const int grid = tiles;          // one workgroup per tile (XCD-aware order inside the kernel)
  if (dg_use_narrow(N, E)) {       // sparse batch of many nodes: eight lanes per node (bit-identical results)
    const int nblk = dg_cdiv(N, DG_NB);
#define DG_FWD32N_LAUNCH(M) hipExtLaunchKernelGGL((k_gcn_fwd32n<M>), dim3(nblk), dim3(256), 0, s, ev_start, ev_stop, 0, N, nblk, \
                                                  rowptr, colidx, dinv, hs, bias, xout, Wnext, hs_next)
    if (mode == 0) DG_FWD32N_LAUNCH(0); else if (mode == 1) DG_FWD32N_LAUNCH(1); else DG_FWD32N_LAUNCH(2);
#undef DG_FWD32N_LAUNCH
    DG_CHECK_LAUNCH();
    return DGCNN_OK;
  }
  // hipExtLaunchKernelGGL attaches the events to THIS dispatch (its own start/end timestamps, the
  // same ones rocprofv3 reports); with null events it is a plain launch.
#define DG_FWD32_LAUNCH(M, D) hipExtLaunchKernelGGL((k_gcn_fwd32<M, D>), dim3(grid), dim3(DG_TILE_THREADS), 0, s, ev_start, \
                                                    ev_stop, 0, N, tiles, rowptr, colidx, dinv, hs, bias, xout, Wnext, hs_next)
#define DG_FWD32P_LAUNCH(M, D) hipExtLaunchKernelGGL((k_gcn_fwd32p<M, D>), dim3(DG_PERSIST_WGS), dim3(DG_TILE_THREADS), 0, s, \
                                                     ev_start, ev_stop, 0, N, tiles, rowptr, colidx, dinv, hs, bias, xout, Wnext, hs_next)
  const bool small = tiles <= DG_SMALL_GRID_TILES;
  static const bool nopersist = dg_knob("DG_NO_PERSIST");     // A/B switch (DG_DEBUG_KNOBS builds only)
  const bool persist = tiles >= DG_PERSIST_MIN_TILES && !nopersist;   // pays from ~4 tiles per workgroup (measured)
  if (mode == 0) { if (small) DG_FWD32_LAUNCH(0, DG_DEPTH_SMALL); else if (persist) DG_FWD32P_LAUNCH(0, DG_DEPTH_FWD_BIG); else DG_FWD32_LAUNCH(0, DG_DEPTH_FWD_BIG); }
  else if (mode == 1) { if (small) DG_FWD32_LAUNCH(1, DG_DEPTH_SMALL); else if (persist) DG_FWD32P_LAUNCH(1, DG_DEPTH_FWD_BIG); else DG_FWD32_LAUNCH(1, DG_DEPTH_FWD_BIG); }
  else { if (small) DG_FWD32_LAUNCH(2, DG_DEPTH_SMALL); else if (persist) DG_FWD32P_LAUNCH(2, DG_DEPTH_FWD_BIG); else DG_FWD32_LAUNCH(2, DG_DEPTH_FWD_BIG); }
#undef DG_FWD32_LAUNCH
#undef DG_FWD32P_LAUNCH
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// forward of conv1, aggregate-first (raw feature width F <= DG_AF_MAX_F):
//     ax[i]  = dinv[i] * ( sum_j xs[j] + xs[i] ),  xs = dinv*x from graph prep   (F-wide gather, saved for backward)
//     x1[i]  = tanh( ax[i] W1^T + b1 )                               (F fmas per channel)
//     hs2[i] = dinv[i] * ( x1[i] W2^T )                              (MFMA on the LDS tile, as k_gcn_fwd32<0>)
// Same tile shape as k_gcn_fwd32: wave per destination node, 16 nodes per workgroup.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(DG_TILE_THREADS)
k_gcn_fwd_af(int N, int F, int lfp, int numTiles, const int* __restrict__ rowptr, const int* __restrict__ colidx,
             const float* __restrict__ dinv, const float* __restrict__ x, const float* __restrict__ W1,
             const float* __restrict__ bias, float* __restrict__ axout, float* __restrict__ xout,
             const float* __restrict__ Wn, float* __restrict__ hs_next) {
  __shared__ __attribute__((aligned(16))) float xt[DG_TILE][DG_LDS_PAD];
  __shared__ float Wt[DG_AF_MAX_F * 32];     // W1 transposed [F][32]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31;

  float wreg[8];
  if (wave < 2) {
    const int cc = wave * 16 + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wreg[kk] = Wn[cc * 32 + 4 * kk + (lane >> 4)];
  }
  const float bc = bias[c];
  for (int t = threadIdx.x; t < 32 * F; t += DG_TILE_THREADS) {
    const int cc = t / F, k = t - cc * F;
    Wt[k * 32 + cc] = W1[t];
  }
  bool wt_ready = false;

  for (int tl = blockIdx.x; tl < numTiles; tl += gridDim.x) {
    const int tile = (gridDim.x == (unsigned)numTiles) ? dg_xcd_tile(tl, numTiles) : tl;
    const int i = tile * DG_TILE + wave;
    float ax = 0.f, di = 0.f;
    if (i < N) {
      const int start = __builtin_amdgcn_readfirstlane(rowptr[i]);
      const int end = __builtin_amdgcn_readfirstlane(rowptr[i + 1]);
      di = dinv[i];
      const float acc = dg_af_gather<true>(x, nullptr, F, lfp, colidx, start, end, i, lane);   // x = xs = dinv*x
      ax = di * acc;
      if (lane < F) axout[(size_t)i * F + lane] = ax;
    }
    float dpre[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = tile * DG_TILE + (lane >> 4) * 4 + r;
        dpre[r] = node < N ? dinv[node] : 0.f;
      }
    }
    if (!wt_ready) { __syncthreads(); wt_ready = true; }     // W1^T staged (its load overlapped the gather)
    float val = 0.f;
    if (i < N) {
      val = dg_tanh(dg_af_transform(ax, F, Wt, lane) + bc);
      if (lane < 32) xout[(size_t)i * 32 + c] = val;
    }
    if (lane < 32) xt[wave][c] = val;
    __syncthreads();
    if (wave < 2) {
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float a = xt[lane & 15][4 * kk + (lane >> 4)];
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[kk], d, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = tile * DG_TILE + (lane >> 4) * 4 + r;
        if (node < N) hs_next[(size_t)node * 32 + wave * 16 + (lane & 15)] = dpre[r] * d[r];
      }
    }
    __syncthreads();
  }
}

int dg_launch_gcn_fwd_af(int N, int F, const int32_t* rowptr, const int32_t* colidx, const float* dinv, const float* x,
                         const float* W1, const float* bias, float* ax, float* xout, const float* Wnext,
                         float* hs_next, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (N <= 0 || F < 1 || F > DG_AF_MAX_F) return DGCNN_EINVAL;
  const int tiles = dg_cdiv(N, DG_TILE);
  hipExtLaunchKernelGGL(k_gcn_fwd_af, dim3(tiles), dim3(DG_TILE_THREADS), 0, s, ev_start, ev_stop, 0, N, F,
                        dg_af_lfp(F), tiles, rowptr, colidx, dinv, x, W1, bias, ax, xout, Wnext, hs_next);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// forward, F = 1 (conv4): wave per node, lanes across neighbours.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dg_gather_row1(const float* __restrict__ src, const int* __restrict__ col,
                                                int start, int end, int lane) {
  // 128 neighbours per round: both index loads, then both value loads, in flight together; the per-lane order of
  // the additions (e, e+64, e+128, ...) is that of the plain loop
  float s = 0.f;
  for (int base = start + lane; base < end; base += 128) {
    const int e1 = base + 64;
    const bool h1 = e1 < end;
    const int c0 = col[base];
    const int c1 = h1 ? col[e1] : 0;
    const float v0 = src[c0];
    const float v1 = h1 ? src[c1] : 0.f;
    s += v0;
    if (h1) s += v1;
  }
  return dg_wave_sum(s);
}

__global__ void __launch_bounds__(256)
k_gcn_fwd1(int N, const int* __restrict__ rowptr, const int* __restrict__ colidx,
           const float* __restrict__ dinv, const float* __restrict__ h4s, const float* __restrict__ bias,
           float* __restrict__ x4) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float b = bias[0];
  for (int i = blockIdx.x * 4 + w; i < N; i += gridDim.x * 4) {
    const int start = rowptr[i], end = rowptr[i + 1];
    const float hself = h4s[i], di = dinv[i];          // issued before the gather, not after it
    const float s = dg_gather_row1(h4s, colidx, start, end, lane) + hself;
    if (lane == 0) x4[i] = dg_tanh(fmaf(di, s, b));
  }
}

// NARROW form of the scalar gather (round 6): eight lanes per destination node, lane q of the group takes the
// node's neighbours q, q + 8, ...; a wave owns 8 nodes.  At DD's degree of ~5 the wave-per-node form above keeps 5 of 64 lanes
// busy and needs 14.5 k waves for a batch of 50; this one 1.8 k.  The group's sum runs dg_wave_sum's first two steps (quad
// swaps on the DPP path) and then adds the two quads: for rows of <= 8 neighbours -- one value per lane -- these are the SAME
// fp32 additions the wave-per-node form performs on its lanes 0..7 (its other lanes add zeros), so the results are bit-identical
// there; longer rows accumulate q, q + 8, ... per lane first and agree to summation-order rounding.
// The total is returned in every lane of the group.
__device__ __forceinline__ float dg_gather_row1_n(const float* __restrict__ src, const int* __restrict__ col, int start, int cnt,
                                                  int lane) {
  const int q = lane & 7;
  int mx = cnt;                                          // trip count: the longest row among the wave's eight nodes
  mx = max(mx, __shfl_xor(mx, 8)); mx = max(mx, __shfl_xor(mx, 16)); mx = max(mx, __shfl_xor(mx, 32));
  mx = __builtin_amdgcn_readfirstlane(mx);
  float s = 0.f;
  for (int base = 0; base < mx; base += 8) {
    const bool h = base + q < cnt;
    const int cj = h ? col[start + base + q] : 0;        // (node 0 stands in where the row has ended: a valid address, never added)
    const float v = src[cj];
    if (h) s += v;
  }
  s += DG_DPP(s, 0xB1, 0xf);    // quad_perm:[1,0,3,2]
  s += DG_DPP(s, 0x4E, 0xf);    // quad_perm:[2,3,0,1]
  s += __shfl_xor(s, 4);        // the group's two quads
  return s;
}
#define DG_NB1 32                  // destination nodes per 256-thread workgroup of the scalar narrow forward
__global__ void __launch_bounds__(256)
k_gcn_fwd1n(int N, const int* __restrict__ rowptr, const int* __restrict__ colidx, const float* __restrict__ dinv,
            const float* __restrict__ h4s, const float* __restrict__ bias, float* __restrict__ x4) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float b = bias[0];
  const int i = (int)blockIdx.x * DG_NB1 + w * 8 + (lane >> 3);
  const bool valid = i < N;
  const int ic = valid ? i : N - 1;                      // (clamped: every load below is unconditional)
  const int start = rowptr[ic], end = rowptr[ic + 1];
  const float hself = h4s[ic], di = dinv[ic];            // issued before the gather, not after it
  const float s = dg_gather_row1_n(h4s, colidx, start, valid ? end - start : 0, lane) + hself;
  if (valid && (lane & 7) == 0) x4[i] = dg_tanh(fmaf(di, s, b));
}

int dg_launch_gcn_fwd1(int N, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                       const float* h4s, const float* bias, float* x4, hipStream_t s, int E) {
  if (N <= 0) return DGCNN_EINVAL;
  if (dg_narrow_level() >= 2 && dg_use_narrow(N, E)) {
    hipLaunchKernelGGL(k_gcn_fwd1n, dim3(dg_cdiv(N, DG_NB1)), dim3(256), 0, s, N, rowptr, colidx, dinv, h4s, bias, x4);
    DG_CHECK_LAUNCH();
    return DGCNN_OK;
  }
  int grid = dg_cdiv(N, 4);
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
// a row of zeros: where a node has no row in the SPARSE SortPooling-gradient slabs (flag word 0) the consumers below redirect
// their row load to it -- unconditional loads on a selected address (a load under a lane-divergent branch is waited for on
// the spot), all of them hits on one cached line; reading the node's unwritten row instead brought 5.6 MB
// of never-used lines in from HBM per layer at DD's batch of 50 (k_gcn_bwd32n 6.9 -> 9.0 us)
__device__ __attribute__((aligned(128))) float dg_zero_row[32];
__device__ int dg_one_word = 1;      // ... and the flag word read where the slabs are dense (no branch around the flag load either)
__global__ void __launch_bounds__(1024)
k_gcn_bwd1(int N, const int* __restrict__ rowptr_t, const int* __restrict__ colidx_t,
           const float* __restrict__ dinv, const float* __restrict__ gas4, const float* __restrict__ W4,
           const float* __restrict__ x3, const float* __restrict__ gp3, float* __restrict__ gas3,
           float* __restrict__ pa4, int P1, DgPrepRider rd, const int* __restrict__ gpsel) {
  if ((int)blockIdx.x >= P1) {   // rider range: phase B of the NEXT batch's graph preparation, when the step's readout
                                 // forward + backward ran as one launch (k_readout_tail carried phase A)
    dg_prep_fast_b_body(((int)blockIdx.x - P1) * 1024 + (int)threadIdx.x, rd.ei, rd.E, rd.N, rd.B, rd.rowptr, rd.colidx,
                        rd.graph_ptr, rd.graph_eptr, rd.dinv, rd.err, rd.epoch, rd.x, rd.xs, rd.F, rd.batch, rd.bits, rd.dmap, rd.edge_check == 1,
                        rd.max_nodes);
    if (rd.dmap && (int)blockIdx.x == P1) dg_prep_dense_plan((int)threadIdx.x, 1024, rd.B, rd.graph_ptr, rd.dmap);
    return;
  }
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & 31;
  const float w4c = W4[c];
  float pW = 0.f, pb = 0.f;
  for (int j = blockIdx.x * 16 + w; j < N; j += P1 * 16) {
    const int start = rowptr_t[j], end = rowptr_t[j + 1];
    // issue everything that does not depend on the gather before it
    float xv = 0.f, gpv = 0.f;
    const int gsel = *(gpsel ? gpsel + j : &dg_one_word);      // (sparse SortPooling-gradient slabs: a row exists only where the flag word says so)
    const float* gprow = gsel ? gp3 + (size_t)j * 32 : dg_zero_row;
    xv = x3[(size_t)j * 32 + c]; gpv = gprow[c];      // (all 64 lanes, c = lane & 31: no lane-divergent branch around the loads)
    const float dj = dinv[j], gself = gas4[j];
    const float s = dg_gather_row1(gas4, colidx_t, start, end, lane) + gself;
    const float gh = dj * s;
    if (lane < 32) {
      const float gx = fmaf(gh, w4c, gpv);
      const float ga = gx * (1.f - xv * xv);
      gas3[(size_t)j * 32 + c] = dj * ga;
      pW = fmaf(gh, xv, pW);
      pb += ga;
    }
  }
  if (lane < 32) { red[w][c] = pW; red[w][32 + c] = pb; }
  __syncthreads();
  if (threadIdx.x < 64) {     // fixed-order tree over the 16 waves
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = red[u][threadIdx.x];
#pragma unroll
    for (int st = 8; st >= 1; st >>= 1)
#pragma unroll
      for (int u = 0; u < st; ++u) v[u] += v[u + st];
    pa4[(size_t)blockIdx.x * 64 + threadIdx.x] = v[0];
  }
}

// NARROW form of k_gcn_bwd1 (round 6, no rider range): eight lanes per node -- lane q of a group gathers neighbours q, q + 8, ...
// of the scalar gas4 (dg_gather_row1_n) and then owns the 16-byte chunk q of the node's 128-byte rows (x3, gp3, gas3), as in the
// 32-wide narrow kernels.  256 threads = 32 nodes per pass; P1 workgroups, workgroup p takes node blocks p, p + P1, ...; every
// workgroup writes its row of pa4 (zeros where it had no node: k_wgrad sums all P1 rows).  Partial sums: per lane over its passes,
// then over the wave's eight groups (xor 8, 16, 32) and the four waves, all in a fixed order.
__global__ void __launch_bounds__(256)
k_gcn_bwd1n(int N, const int* __restrict__ rowptr_t, const int* __restrict__ colidx_t, const float* __restrict__ dinv,
            const float* __restrict__ gas4, const float* __restrict__ W4, const float* __restrict__ x3,
            const float* __restrict__ gp3, float* __restrict__ gas3, float* __restrict__ pa4, int P1,
            const int* __restrict__ gpsel) {
  __shared__ __attribute__((aligned(16))) float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 3, q = lane & 7;
  const float4 w4 = *reinterpret_cast<const float4*>(W4 + 4 * q);
  float4 pW = make_float4(0.f, 0.f, 0.f, 0.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nblk = (N + DG_NB1 - 1) / DG_NB1;
  for (int blk = blockIdx.x; blk < nblk; blk += P1) {
    const int j = blk * DG_NB1 + w * 8 + g;
    const bool valid = j < N;
    const int jc = valid ? j : N - 1;
    const int start = rowptr_t[jc], end = rowptr_t[jc + 1];
    // everything that does not depend on the gather is requested before it (unconditional loads on clamped / redirected addresses)
    const int gsel = *(gpsel ? gpsel + jc : &dg_one_word);
    const float* gprow = gsel ? gp3 + (size_t)jc * 32 : dg_zero_row;
    const float4 xv = *reinterpret_cast<const float4*>(x3 + (size_t)jc * 32 + 4 * q);
    const float4 gpv = *reinterpret_cast<const float4*>(gprow + 4 * q);
    const float dj = dinv[jc], gself = gas4[jc];
    const float sg = dg_gather_row1_n(gas4, colidx_t, start, valid ? end - start : 0, lane) + gself;
    const float gh = dj * sg;
    if (valid) {
      float4 ga;
      ga.x = fmaf(gh, w4.x, gpv.x) * (1.f - xv.x * xv.x);
      ga.y = fmaf(gh, w4.y, gpv.y) * (1.f - xv.y * xv.y);
      ga.z = fmaf(gh, w4.z, gpv.z) * (1.f - xv.z * xv.z);
      ga.w = fmaf(gh, w4.w, gpv.w) * (1.f - xv.w * xv.w);
      *reinterpret_cast<float4*>(gas3 + (size_t)j * 32 + 4 * q) = make_float4(dj * ga.x, dj * ga.y, dj * ga.z, dj * ga.w);
      pW.x = fmaf(gh, xv.x, pW.x); pW.y = fmaf(gh, xv.y, pW.y); pW.z = fmaf(gh, xv.z, pW.z); pW.w = fmaf(gh, xv.w, pW.w);
      pb.x += ga.x; pb.y += ga.y; pb.z += ga.z; pb.w += ga.w;
    }
  }
  // the wave's eight groups (lanes with the same q), fixed order; then the four waves
  float v[8] = {pW.x, pW.y, pW.z, pW.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    v[u] += __shfl_xor(v[u], 8);
    v[u] += __shfl_xor(v[u], 16);
    v[u] += __shfl_xor(v[u], 32);
  }
  if (g == 0) {      // lane q holds channels 4q .. 4q + 3 of dW4 (v[0..3]) and db3 (v[4..7])
    *reinterpret_cast<float4*>(&red[w][4 * q]) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(&red[w][32 + 4 * q]) = make_float4(v[4], v[5], v[6], v[7]);
  }
  __syncthreads();
  if (threadIdx.x < 64)
    pa4[(size_t)blockIdx.x * 64 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int dg_launch_gcn_bwd1(int N, const int32_t* rowptr_t, const int32_t* colidx_t, const float* dinv,
                       const float* gas4, const float* W4, const float* x3, const float* gp3,
                       float* gas3, float* pa4, int P1, hipStream_t s, const DgPrepRider* rider, const int32_t* gpsel, int E) {
  if (N <= 0 || P1 <= 0) return DGCNN_EINVAL;
  if (dg_narrow_level() >= 2 && !rider && dg_use_narrow(N, E) && (((uintptr_t)x3 | (uintptr_t)gp3 | (uintptr_t)gas3 | (uintptr_t)W4) & 15) == 0) {
    hipLaunchKernelGGL(k_gcn_bwd1n, dim3(P1), dim3(256), 0, s, N, rowptr_t, colidx_t, dinv, gas4, W4, x3, gp3, gas3, pa4, P1, gpsel);
    DG_CHECK_LAUNCH();
    return DGCNN_OK;
  }
  DgPrepRider rd{};
  if (rider) rd = *rider;
  hipLaunchKernelGGL(k_gcn_bwd1, dim3(P1 + rd.nblk_b), dim3(1024), 0, s, N, rowptr_t, colidx_t, dinv, gas4, W4, x3, gp3, gas3,
                     pa4, P1, rd, gpsel);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// ---------------------------------------------------------------------------------------------
// backward of a 32-wide layer l (l = 3, 2): input gas_l [N,32] (= dinv * dL/d pre-activation)
//   gh[j]      = dinv[j] * ( sum_{i in N_out(j)} gas_l[i] + gas_l[j] )        -> LDS tile [16][32]
//   dW_l      += gh^T . x_{l-1}          (MFMA 16x16x4, K = 16 nodes of the tile; waves 2..5)
//   gx_{l-1}   = gh . W_l + gp_{l-1}     (MFMA; waves 0..1)
//   ga_{l-1}   = gx_{l-1} * (1 - x_{l-1}^2) ; gas_{l-1} = dinv * ga_{l-1} ; db_{l-1} += ga_{l-1}
// part[P][1056] = per-workgroup {dW_l [32x32], db_{l-1} [32]}.
//
// FIRST = true (layer 1): x_{l-1} is the raw input x [N,F]; only dW_1 [32,F] is produced
// (data.x needs no gradient, /root/reference/train.py:36-40).  part[P][32*F].
// ---------------------------------------------------------------------------------------------
//
// AF = true (layer 2 when conv1 ran aggregate-first): additionally dW_1 += ga_1^T . ax  with ax = A_hat X [N,Fa]
// saved by k_gcn_fwd_af -- conv1's whole backward, no gather needed.  part1[P][32*Fa] in W1's own layout.
// large grids (DEPTH <= 4): cap the registers at 64 so that TWO 1024-thread workgroups fit a CU (8 waves per SIMD)
template <bool FIRST, bool AF, int DEPTH>
__global__ void __launch_bounds__(DG_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(DEPTH <= 4 ? 8 : 4)))
k_gcn_bwd32(int N, int F, int numTiles, const int* __restrict__ rowptr_t, const int* __restrict__ colidx_t,
            const float* __restrict__ dinv, const float* __restrict__ gas, const float* __restrict__ Wl,
            const float* __restrict__ xprev, const float* __restrict__ gpprev, float* __restrict__ gas_prev,
            float* __restrict__ part, const float* __restrict__ axin, int Fa, float* __restrict__ part1,
            const int* __restrict__ gpsel) {
  __shared__ __attribute__((aligned(16))) float ght[DG_TILE][DG_LDS_PAD];
  __shared__ __attribute__((aligned(16))) float xt[DG_TILE][DG_LDS_PAD];
  __shared__ float gat[AF ? DG_TILE : 1][DG_LDS_PAD];          // AF: ga_1 tile
  __shared__ float axs[AF ? DG_TILE * DG_AF_MAX_F : 1];        // AF: ax tile [16][Fa]
  float accA = 0.f;                                            // AF: dW1[c][k], thread t = k*32 + c
  DG_DYN_SMEM(float, xs);   // FIRST: [16][F] raw-input tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 3, q = lane & 7;

  // persistent accumulators
  float wreg[8];
  f32x4 accW = {0.f, 0.f, 0.f, 0.f};
  float pb = 0.f;
  // FIRST: dW1 [32 x F] as 2 x ceil(F/16) MFMA tiles, dealt round-robin to the 16 waves (<= 4 tiles per wave for
  // F <= DGCNN_MAX_F = 512); the accumulators persist over this workgroup's tiles
  f32x4 acc1[(2 * (DGCNN_MAX_F / 16)) / 16];
  const int ntile1 = FIRST ? 2 * ((F + 15) >> 4) : 0;
  if (FIRST) {
#pragma unroll
    for (int u = 0; u < (2 * (DGCNN_MAX_F / 16)) / 16; ++u) acc1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else if (wave < 2) {   // B operand of gx = gh . W_l : B[k][n] = W_l[k][nb*16+n]
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wreg[kk] = Wl[(4 * kk + (lane >> 4)) * 32 + wave * 16 + (lane & 15)];
  }

  // each workgroup owns a CONTIGUOUS chunk of tiles, and chunks are laid out XCD-contiguously
  const int chunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int wg = dg_xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int tile_end = min(numTiles, (wg + 1) * chunk);
  for (int tile = wg * chunk; tile < tile_end; ++tile) {
    const int j = tile * DG_TILE + wave;
    if (j < N) {
      const int start = __builtin_amdgcn_readfirstlane(rowptr_t[j]);
      const int end = __builtin_amdgcn_readfirstlane(rowptr_t[j + 1]);
      // everything that does not depend on the gather is loaded before it
      const float dj = dinv[j];
      float4 xrow = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!FIRST && g == 1) xrow = *reinterpret_cast<const float4*>(xprev + (size_t)j * 32 + 4 * q);
      float axv = 0.f;
      if (AF && lane < Fa) axv = axin[(size_t)j * Fa + lane];
      float4 acc = dg_gather_row32<DEPTH>(gas, colidx_t, start, end, j, lane);
      acc.x *= dj; acc.y *= dj; acc.z *= dj; acc.w *= dj;
      if (g == 0) *reinterpret_cast<float4*>(&ght[wave][4 * q]) = acc;
      if (!FIRST && g == 1) *reinterpret_cast<float4*>(&xt[wave][4 * q]) = xrow;
      if (FIRST)
        for (int k = lane; k < F; k += 64) xs[wave * F + k] = xprev[(size_t)j * F + k];
      if (AF && lane < Fa) axs[wave * Fa + lane] = axv;
    } else {
      if (AF && lane < Fa) axs[wave * Fa + lane] = 0.f;
      if (g == 0) *reinterpret_cast<float4*>(&ght[wave][4 * q]) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!FIRST && g == 1) *reinterpret_cast<float4*>(&xt[wave][4 * q]) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (FIRST)
        for (int k = lane; k < F; k += 64) xs[wave * F + k] = 0.f;
    }
    // operands of the MFMA epilogue (SortPooling gradient rows, dst scales): loads issued before the barrier
    float gpp[4] = {0.f, 0.f, 0.f, 0.f}, dnn[4] = {0.f, 0.f, 0.f, 0.f};
    if (!FIRST && wave < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int node = tile * DG_TILE + (lane >> 4) * 4 + r;
        if (node < N) {
          const int gs = *(gpsel ? gpsel + node : &dg_one_word);      // (sparse slabs: see k_gcn_bwd1)
          gpp[r] = (gs ? gpprev + (size_t)node * 32 : dg_zero_row)[wave * 16 + (lane & 15)];
          dnn[r] = dinv[node];
        }
      }
    }
    __syncthreads();
    if (FIRST) {
      // dW1[c][k] += sum_node ght[node][c] * xs[node][k]  on the matrix cores: A[m][kk] = ght[kk][mb*16+m],
      // B[kk][n] = xs[kk][nb*16+n], K = the 16 nodes of the tile (4 MFMA steps)
#pragma unroll
      for (int u = 0; u < (2 * (DGCNN_MAX_F / 16)) / 16; ++u) {
        const int t = u * 16 + wave;
        if (t < ntile1) {
          const int mb = t & 1, nb = t >> 1;
          const int col = nb * 16 + (lane & 15);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int nd = 4 * kk + (lane >> 4);
            const float a = ght[nd][mb * 16 + (lane & 15)];
            const float b = col < F ? xs[nd * F + col] : 0.f;
            acc1[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1[u], 0, 0, 0);
          }
        }
      }
    } else {
      if (wave < 2) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float a = ght[lane & 15][4 * kk + (lane >> 4)];
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[kk], d, 0, 0, 0);
        }
        const int c = wave * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (lane >> 4) * 4 + r;
          const int node = tile * DG_TILE + row;
          float ga = 0.f;
          if (node < N) {
            const float xv = xt[row][c];
            const float gx = d[r] + gpp[r];
            ga = gx * (1.f - xv * xv);
            if (!AF) gas_prev[(size_t)node * 32 + c] = dnn[r] * ga;       // AF: conv1 needs no propagated gradient
            pb += ga;
          }
          if (AF) gat[row][c] = ga;
        }
      } else if (wave < 6) {   // dW block (mb, nb): A[m][k] = ght[k][mb*16+m], B[k][n] = xt[k][nb*16+n]
        const int mb = (wave - 2) >> 1, nb = (wave - 2) & 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = 4 * kk + (lane >> 4);
          const float a = ght[k][mb * 16 + (lane & 15)];
          const float b = xt[k][nb * 16 + (lane & 15)];
          accW = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accW, 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (AF) {
      if ((int)threadIdx.x < 32 * Fa) {
        const int k = threadIdx.x >> 5, c = threadIdx.x & 31;
        float a = accA;
#pragma unroll
        for (int nd = 0; nd < DG_TILE; ++nd) a = fmaf(gat[nd][c], axs[nd * Fa + k], a);
        accA = a;
      }
      if (tile + 1 < tile_end) __syncthreads();
    }
  }

  // write this workgroup's partials
  if (AF && (int)threadIdx.x < 32 * Fa) {
    const int k = threadIdx.x >> 5, c = threadIdx.x & 31;
    part1[(size_t)blockIdx.x * 32 * Fa + c * Fa + k] = accA;
  }
  if (FIRST) {
    float* dst = part + (size_t)blockIdx.x * 32 * F;
#pragma unroll
    for (int u = 0; u < (2 * (DGCNN_MAX_F / 16)) / 16; ++u) {
      const int t = u * 16 + wave;
      if (t < ntile1) {
        const int mb = t & 1, nb = t >> 1;
        const int k = nb * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = mb * 16 + (lane >> 4) * 4 + r;
          if (k < F) dst[c * F + k] = acc1[u][r];     // stored in W1's own [32,F] layout
        }
      }
    }
  } else {
    float* dst = part + (size_t)blockIdx.x * 1056;
    if (wave < 2) {
      // lanes l, l+16, l+32, l+48 hold the same column: fixed-order combine
      pb += __shfl_xor(pb, 16);
      pb += __shfl_xor(pb, 32);
      if (lane < 16) dst[1024 + wave * 16 + lane] = pb;
    } else if (wave < 6) {
      const int mb = (wave - 2) >> 1, nb = (wave - 2) & 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mb * 16 + (lane >> 4) * 4 + r;     // output channel of W_l
        const int col = nb * 16 + (lane & 15);             // input channel
        dst[row * 32 + col] = accW[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NARROW form of the 32-wide layer backward (layers 3 and 2 without the aggregate-first rider; see k_gcn_fwd32n): eight lanes
// per node, 32 nodes = two 16-row tiles per 256-thread workgroup and trip.  Every accumulator sees the additions of
// k_gcn_bwd32 in the same order -- the chunk of tiles a workgroup owns is the same, its tiles are taken in order (two per
// trip), wave w < 2 runs the data-gradient block nb = w of BOTH tiles one after the other (db accumulates tile by tile), wave
// w runs weight-gradient block (mb, nb) = (w >> 1, w & 1) of both tiles: partial rows are bit-identical to the wide form's.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gcn_bwd32n(int N, int numTiles, const int* __restrict__ rowptr_t, const int* __restrict__ colidx_t,
             const float* __restrict__ dinv, const float* __restrict__ gas, const float* __restrict__ Wl,
             const float* __restrict__ xprev, const float* __restrict__ gpprev, float* __restrict__ gas_prev,
             float* __restrict__ part, const int* __restrict__ gpsel) {
  __shared__ __attribute__((aligned(16))) float ght[DG_NB][DG_LDS_PAD];
  __shared__ __attribute__((aligned(16))) float xt[DG_NB][DG_LDS_PAD];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 3, q = lane & 7;
  float wreg[8];
  f32x4 accW = {0.f, 0.f, 0.f, 0.f};
  float pb = 0.f;
  if (wave < 2) {   // B operand of gx = gh . W_l : B[k][n] = W_l[k][nb*16+n]
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) wreg[kk] = Wl[(4 * kk + (lane >> 4)) * 32 + wave * 16 + (lane & 15)];
  }
  const int mbW = wave >> 1, nbW = wave & 1;
  const int chunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int wg = dg_xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int tile_end = min(numTiles, (wg + 1) * chunk);
  for (int tile = wg * chunk; tile < tile_end; tile += 2) {
    const int nt = __builtin_amdgcn_readfirstlane(min(2, tile_end - tile));      // tiles of this trip
    const int j = tile * DG_TILE + wave * 8 + g;
    const bool valid = j < N && (wave >> 1) < nt;
    int start = 0, end = 0;
    float dj = 0.f;
    float4 xrow = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      start = rowptr_t[j]; end = rowptr_t[j + 1]; dj = dinv[j];
      xrow = *reinterpret_cast<const float4*>(xprev + (size_t)j * 32 + 4 * q);
    }
    // operands of the matrix-core epilogue (SortPooling gradient rows, dst scales): loads issued before the gather, every one
    // UNCONDITIONAL on a clamped node (a load under a lane-divergent branch is waited for on the spot: with the flag word of the
    // sparse slabs in front of the row that was eight exposed round trips, 6.9 -> 8.7 us); rows without a flag come from
    // dg_zero_row, rows beyond N / beyond the chunk are never used
    float gpp[2][4], dnn[2][4];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 4; ++r) { gpp[t2][r] = 0.f; dnn[t2][r] = 0.f; }
    if (wave < 2) {
      int nc[2][4], gs[2][4];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          nc[t2][r] = min((tile + t2) * DG_TILE + (lane >> 4) * 4 + r, N - 1);
          gs[t2][r] = *(gpsel ? gpsel + nc[t2][r] : &dg_one_word);
          dnn[t2][r] = dinv[nc[t2][r]];
        }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          gpp[t2][r] = (gs[t2][r] ? gpprev + (size_t)nc[t2][r] * 32 : dg_zero_row)[wave * 16 + (lane & 15)];
    }
    float4 acc = dg_gather_row32_n(gas, colidx_t, start, end - start, j, valid, lane);
    acc.x *= dj; acc.y *= dj; acc.z *= dj; acc.w *= dj;
    *reinterpret_cast<float4*>(&ght[wave * 8 + g][4 * q]) = acc;       // (rows beyond N / beyond the chunk: zeros)
    *reinterpret_cast<float4*>(&xt[wave * 8 + g][4 * q]) = xrow;
    __syncthreads();
    if (wave < 2) {
      const int c = wave * 16 + (lane & 15);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        if (t2 < nt) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const float a = ght[t2 * 16 + (lane & 15)][4 * kk + (lane >> 4)];
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[kk], d, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) * 4 + r;
            const int node = (tile + t2) * DG_TILE + row;
            if (node < N) {
              const float xv = xt[t2 * 16 + row][c];
              const float gx = d[r] + gpp[t2][r];
              const float ga = gx * (1.f - xv * xv);
              gas_prev[(size_t)node * 32 + c] = dnn[t2][r] * ga;
              pb += ga;
            }
          }
        }
      }
    }
    // dW block (mbW, nbW): A[m][k] = ght[k][mb*16+m], B[k][n] = xt[k][nb*16+n], K = the tile's 16 nodes
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      if (t2 < nt) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int k = t2 * 16 + 4 * kk + (lane >> 4);
          const float a = ght[k][mbW * 16 + (lane & 15)];
          const float b = xt[k][nbW * 16 + (lane & 15)];
          accW = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, accW, 0, 0, 0);
        }
      }
    }
    if (tile + 2 < tile_end) __syncthreads();
  }
  float* dst = part + (size_t)blockIdx.x * 1056;
  if (wave < 2) {
    // lanes l, l+16, l+32, l+48 hold the same column: fixed-order combine
    pb += __shfl_xor(pb, 16);
    pb += __shfl_xor(pb, 32);
    if (lane < 16) dst[1024 + wave * 16 + lane] = pb;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = mbW * 16 + (lane >> 4) * 4 + r;     // output channel of W_l
    const int col = nbW * 16 + (lane & 15);             // input channel
    dst[row * 32 + col] = accW[r];
  }
}

// NARROW form of conv1's own backward when conv1 ran linear-first (raw feature width F in (32, DG_LIN_STAGE_MAX_F], 16-byte
// aligned x): only dW_1 [32,F] = gh^T . x is produced (data.x needs no gradient, /root/reference/train.py:36-40).  The raw
// rows of a trip's 32 nodes are ONE contiguous, 64-byte aligned run of 32 F floats (as in k_lin_first32s): 16-byte loads, flat
// copy into the LDS tile [32][F].  The 2 x ceil(F/16) output tiles are dealt round-robin to the FOUR waves (k_gcn_bwd32<FIRST>:
// to sixteen); each accumulator takes the workgroup's tiles in the same order, four matrix steps per tile: identical partials.
#define DG_N1_ACC ((2 * (DG_LIN_STAGE_MAX_F / 16)) / 4)
__global__ void __launch_bounds__(256)
k_gcn_bwd32n1(int N, int F, int numTiles, const int* __restrict__ rowptr_t, const int* __restrict__ colidx_t,
              const float* __restrict__ dinv, const float* __restrict__ gas, const float* __restrict__ xraw,
              float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float ght[DG_NB][DG_LDS_PAD];
  DG_DYN_SMEM(float, xs);    // [32][F]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 3, q = lane & 7;
  f32x4 acc1[DG_N1_ACC];
#pragma unroll
  for (int u = 0; u < DG_N1_ACC; ++u) acc1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntile1 = 2 * ((F + 15) >> 4);
  constexpr int XQ = DG_LIN_STAGE_MAX_F * DG_NB / 4 / 256;     // 16-byte pieces of the raw tile per thread (4 at the largest F)
  const int nq = 8 * F;                                        // 16-byte pieces of a full 32-row tile
  const int chunk = (numTiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int wg = dg_xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int tile_end = min(numTiles, (wg + 1) * chunk);
  for (int tile = wg * chunk; tile < tile_end; tile += 2) {
    const int nt = __builtin_amdgcn_readfirstlane(min(2, tile_end - tile));      // tiles of this trip
    const int r0 = tile * DG_TILE;
    const int j = r0 + wave * 8 + g;
    const bool valid = j < N && (wave >> 1) < nt;
    int start = 0, end = 0;
    float dj = 0.f;
    if (valid) { start = rowptr_t[j]; end = rowptr_t[j + 1]; dj = dinv[j]; }
    const int nfl = max(0, min(nt * DG_TILE, N - r0)) * F;     // floats of the raw tile that exist (rows beyond: zeros)
    const float* xt = xraw + (size_t)r0 * F;
    float4 v[XQ];
    if ((nfl & 3) == 0) {      // (workgroup-uniform; see k_lin_first32s: no piece straddles the end)
#pragma unroll
      for (int i = 0; i < XQ; ++i) {
        const int p = (int)threadIdx.x + 256 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < nq && 4 * p + 3 < nfl) v[i] = *reinterpret_cast<const float4*>(xt + 4 * p);
      }
    } else {                   // the batch's last rows: unconditional dword loads on clamped indices, selected
#pragma unroll
      for (int i = 0; i < XQ; ++i) {
        const int e = 4 * ((int)threadIdx.x + 256 * i);
        const float t0 = xt[min(e, nfl - 1)], t1 = xt[min(e + 1, nfl - 1)], t2 = xt[min(e + 2, nfl - 1)], t3 = xt[min(e + 3, nfl - 1)];
        v[i] = make_float4(e < nfl ? t0 : 0.f, e + 1 < nfl ? t1 : 0.f, e + 2 < nfl ? t2 : 0.f, e + 3 < nfl ? t3 : 0.f);
      }
    }
    float4 acc = dg_gather_row32_n(gas, colidx_t, start, end - start, j, valid, lane);
    acc.x *= dj; acc.y *= dj; acc.z *= dj; acc.w *= dj;
    *reinterpret_cast<float4*>(&ght[wave * 8 + g][4 * q]) = acc;
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
      const int p = (int)threadIdx.x + 256 * i;
      if (p < nq) *reinterpret_cast<float4*>(xs + 4 * p) = v[i];
    }
    __syncthreads();
    // dW1[c][k] += sum_node ght[node][c] * xs[node][k]: A[m][kk] = ght[kk][mb*16+m], B[kk][n] = xs[kk][nb*16+n], K = a tile's 16 nodes
#pragma unroll
    for (int u = 0; u < DG_N1_ACC; ++u) {
      const int t = u * 4 + wave;
      if (t < ntile1) {
        const int mb = t & 1, nb = t >> 1;
        const int col = nb * 16 + (lane & 15);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          if (t2 < nt) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const int nd = t2 * 16 + 4 * kk + (lane >> 4);
              const float a = ght[nd][mb * 16 + (lane & 15)];
              const float b = col < F ? xs[nd * F + col] : 0.f;
              acc1[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1[u], 0, 0, 0);
            }
          }
        }
      }
    }
    if (tile + 2 < tile_end) __syncthreads();
  }
  float* dst = part + (size_t)blockIdx.x * 32 * F;
#pragma unroll
  for (int u = 0; u < DG_N1_ACC; ++u) {
    const int t = u * 4 + wave;
    if (t < ntile1) {
      const int mb = t & 1, nb = t >> 1;
      const int k = nb * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = mb * 16 + (lane >> 4) * 4 + r;
        if (k < F) dst[c * F + k] = acc1[u][r];     // stored in W1's own [32,F] layout
      }
    }
  }
}

int dg_launch_gcn_bwd32(int first, int N, int F, const int32_t* rowptr_t, const int32_t* colidx_t,
                        const float* dinv, const float* gas, const float* Wl, const float* xprev,
                        const float* gpprev, float* gas_prev, float* part, int P32, hipStream_t s,
                        const float* ax, int Fa, float* part1, int E, const int32_t* gpsel) {
  if (N <= 0 || P32 <= 0) return DGCNN_EINVAL;
  const int tiles = dg_cdiv(N, DG_TILE);
  const bool small = tiles <= DG_SMALL_GRID_TILES;
  if (first && F > 32 && F <= DG_LIN_STAGE_MAX_F && ((uintptr_t)xprev & 15) == 0 && dg_use_narrow(N, E)) {
    hipLaunchKernelGGL(k_gcn_bwd32n1, dim3(P32), dim3(256), sizeof(float) * DG_NB * F, s, N, F, tiles, rowptr_t, colidx_t, dinv, gas,
                       xprev, part);
    DG_CHECK_LAUNCH();
    return DGCNN_OK;
  }
  if (!first && !ax && dg_use_narrow(N, E)) {
    hipLaunchKernelGGL(k_gcn_bwd32n, dim3(P32), dim3(256), 0, s, N, tiles, rowptr_t, colidx_t, dinv, gas, Wl, xprev, gpprev,
                       gas_prev, part, gpsel);
    DG_CHECK_LAUNCH();
    return DGCNN_OK;
  }
#define DG_BWD32_LAUNCH(FI, AFV, D, LDS, FF, AX, FA, P1)                                                                 \
  hipLaunchKernelGGL((k_gcn_bwd32<FI, AFV, D>), dim3(P32), dim3(DG_TILE_THREADS), LDS, s, N, FF, tiles, rowptr_t, colidx_t, \
                     dinv, gas, Wl, xprev, gpprev, gas_prev, part, AX, FA, P1, gpsel)
  if (first) {
    if (F < 1 || F > DGCNN_MAX_F) return DGCNN_EINVAL;
    if (small) DG_BWD32_LAUNCH(true, false, DG_DEPTH_SMALL, sizeof(float) * DG_TILE * F, F, nullptr, 0, nullptr);
    else DG_BWD32_LAUNCH(true, false, DG_DEPTH_BWD_BIG, sizeof(float) * DG_TILE * F, F, nullptr, 0, nullptr);
  } else if (ax) {     // conv2 backward carrying conv1's weight gradient (aggregate-first conv1)
    if (Fa < 1 || Fa > DG_AF_MAX_F || !part1) return DGCNN_EINVAL;
    if (small) DG_BWD32_LAUNCH(false, true, DG_DEPTH_SMALL, 0, 32, ax, Fa, part1);
    else DG_BWD32_LAUNCH(false, true, DG_DEPTH_BWD_BIG, 0, 32, ax, Fa, part1);
  } else {
    if (small) DG_BWD32_LAUNCH(false, false, DG_DEPTH_SMALL, 0, 32, nullptr, 0, nullptr);
    else DG_BWD32_LAUNCH(false, false, DG_DEPTH_BWD_BIG, 0, 32, nullptr, 0, nullptr);
  }
#undef DG_BWD32_LAUNCH
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}
