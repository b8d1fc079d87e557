// fused.hip -- graph-per-workgroup fused forward kernel (gfx950 / CDNA4).
//
// At the reference's batch size (50 graphs, ~75 nodes each; /root/reference/train.py:21) one graph's
// whole activation state (n x 32 fp32 per layer = n * 128 B) AND its adjacency fit in a CU's 160 KiB
// LDS, and the per-op chain of gcn.hip/tail.hip is bound by kernel boundaries and dependent HBM/L2
// round trips, not by bandwidth.  k_fused_fwd therefore runs, for ONE graph per workgroup (16 waves),
// the entire forward of /root/reference/model.py:26-45 in a single launch:
//
//     stage this graph's CSR (row pointers + local neighbour ids) and dinv in LDS   (one coalesced pass)
//     conv1 linear (x W1^T, pre-scaled)                          -> H (LDS, [n][32])
//     3 x { half-wave per node, lane = channel: sequential sum over the row's neighbours read from H
//           (ds_read_b32, conflict-free) + self, dst scale, bias, tanh -> X (LDS) and x_l (HBM, saved)
//           next layer's X W^T on v_mfma_f32_16x16x4_f32           -> H (LDS, overwritten in place) }
//     conv4 (32 -> 1): per-node dot + scalar gather              -> x4 (LDS sort keys + HBM)
//     SortPooling (LDS sort) + conv5/pool/conv6/MLP/log_softmax  (dg_readout.h)
//
// Inside the layer loop there is NO global load at all: indices, neighbour rows, scales all come
// from LDS.  HBM sees x, the CSR slice and dinv once, and the x1..x4 slabs written once because
// backward needs them.  Summation order is the canonical one of gcn.hip (sequential over ascending
// neighbours, self last), so this path is bit-identical to the tiled kernels.
//
// Requirements (host hints, verified on the device and reported through the error words): every
// graph has at most nmax nodes, the batch is block-diagonal (an edge leaving its graph is flagged
// and skipped).  A graph with more than emax_lds edges simply reads its neighbour ids from global.
#include "dg_common.h"
#include "dg_readout.h"
#include <hip/hip_ext.h>

__device__ __forceinline__ size_t fg_a16_dev(size_t x) { return (x + 15) & ~(size_t)15; }

#define FG_THREADS 1024
#define FG_SLOTS 32          // node slots per pass: 16 waves x 2 half-waves
#define FG_RS 33             // row stride (floats) of H and X: conflict-free for lane=channel rows AND MFMA A reads

// LDS layout (bytes), dynamic:
//   region0 : max(2*(nmax+1)*RS*4, (nmax+1)*RS*4 + 32*F*4, RD_REGION0_BYTES)   H | X  (aliased later by the readout)
//             row nmax of H is an all-zero row: padded / invalid neighbour slots point at it
//   dv, h4s, x4s : (nmax+1)*4 each ;  rp : (nmax+1)*4 ;  cl : emax_lds*4 (+32 slack) ;  prm : 160*4 ;  small
static inline size_t fg_a16(size_t x) { return (x + 15) & ~(size_t)15; }
static inline size_t fg_region0_bytes(int nmax, int F) {
  const size_t row = (size_t)(nmax + 1) * FG_RS * 4;
  size_t a = 2 * row;
  size_t b = row + (size_t)32 * F * 4;
  size_t r = a > b ? a : b;
  if (r < RD_REGION0_BYTES) r = RD_REGION0_BYTES;
  return fg_a16(r);
}
static inline size_t fg_lds_bytes(int nmax, int F, int emax_lds) {
  return fg_region0_bytes(nmax, F) + 4 * fg_a16((size_t)(nmax + 1) * 4) + fg_a16((size_t)emax_lds * 4 + 32) +
         160 * 4 + RD_SMALL_BYTES + 16;
}
#define FG_LDS_CAP (160 * 1024)

struct FgW {   // GCN parameters (device pointers into the flat buffer)
  const float *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4;
};

#define FG_XPF 4      // conv1: x rows of the first FG_XPF passes are prefetched into registers at kernel start

__global__ void __launch_bounds__(FG_THREADS)
k_fused_fwd(int F, int C, int nmax, int emax_lds, size_t region0_bytes, FgW gw, TailW tw,
            const float* __restrict__ xin, const int* __restrict__ rowptr, const int* __restrict__ colidx,
            const float* __restrict__ dinv, const int* __restrict__ graph_ptr, const int* __restrict__ graph_eptr,
            float* __restrict__ x1, float* __restrict__ x2, float* __restrict__ x3, float* __restrict__ x4,
            float* __restrict__ pooled, int* __restrict__ perm, float* __restrict__ a5g, float* __restrict__ a6g,
            float* __restrict__ a1dg, uint8_t* __restrict__ maskg, float* __restrict__ logp, int training,
            uint64_t seed, unsigned int* __restrict__ err, unsigned int epoch, unsigned long long* dbg) {
#define FG_MARK(k) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[k] = clock64(); } while (0)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  // one round trip: node range and edge range of this graph
  const int n0 = graph_ptr[b], n1 = graph_ptr[b + 1];
  const int e0 = graph_eptr[b], e1 = graph_eptr[b + 1];
  const int n = n1 - n0, ne = e1 - e0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31;
  const bool upper = lane >= 32;
  const int slot = wave * 2 + (upper ? 1 : 0);
  const int nz = nmax;                                   // index of the zero row
  const size_t nb4 = fg_a16_dev((size_t)(nmax + 1) * 4);
  float* H = reinterpret_cast<float*>(smem);
  float* X = H + (size_t)(nmax + 1) * FG_RS;
  char* p = smem + region0_bytes;
  float* dv = reinterpret_cast<float*>(p);  p += nb4;
  float* h4s = reinterpret_cast<float*>(p); p += nb4;
  float* x4s = reinterpret_cast<float*>(p); p += nb4;
  int* rp = reinterpret_cast<int*>(p);      p += nb4;
  int* cl = reinterpret_cast<int*>(p);      p += fg_a16_dev((size_t)emax_lds * 4 + 32);
  float* prm = reinterpret_cast<float*>(p); p += 160 * 4;   // b1|b2|b3 (96) W4 (32) b4 (1)
  char* small = p;
  bool bad = false;
  FG_MARK(0);

  if (n > nmax) {      // host hint violated: flag and produce nothing for this graph (never overrun LDS)
    if (tid == 0) { err[1] = epoch; err[3] = ~epoch; }
    return;
  }
  // ---- second round trip, everything in parallel: parameters, CSR slice, dinv, first x rows ----
  float wreg2[8], wreg3[8];     // B operands of the two MFMA post-steps: B[k][nn] = W[nb*16+nn][k], nb = wave & 1
  {
    const int cc = (wave & 1) * 16 + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      wreg2[kk] = gw.W2[cc * 32 + 4 * kk + (lane >> 4)];
      wreg3[kk] = gw.W3[cc * 32 + 4 * kk + (lane >> 4)];
    }
  }
  float xpre[FG_XPF];           // x[i][c] for i = slot + 32*pf (only the first 32 feature columns)
#pragma unroll
  for (int pf = 0; pf < FG_XPF; ++pf) {
    const int i = slot + FG_SLOTS * pf;
    xpre[pf] = (i < n && c < F) ? xin[(size_t)(n0 + i) * F + c] : 0.f;
  }
  if (tid < 32) { prm[tid] = gw.b1[tid]; prm[32 + tid] = gw.b2[tid]; prm[64 + tid] = gw.b3[tid]; prm[96 + tid] = gw.W4[tid]; }
  if (tid == 32) prm[128] = gw.b4[0];
  const bool ldscol = ne <= emax_lds;            // workgroup-uniform
  for (int t = tid; t <= n; t += FG_THREADS) rp[t] = rowptr[n0 + t] - (ldscol ? e0 : 0);
  for (int t = tid; t < n; t += FG_THREADS) dv[t] = dinv[n0 + t];
  if (ldscol)
    for (int t = tid; t < ne; t += FG_THREADS) {
      const int j = colidx[e0 + t] - n0;
      const bool ok = (unsigned)j < (unsigned)n;
      if (!ok) bad = true;
      cl[t] = ok ? j : nz;
    }
  if (tid < 32) H[nz * FG_RS + tid] = 0.f;
  if (tid == 0) h4s[nz] = 0.f;
  float* Wt = X;                       // [F][32], X is free until the first gather
  for (int t = tid; t < 32 * F; t += FG_THREADS) {
    const int cc = t / F, k = t - cc * F;
    Wt[k * 32 + cc] = gw.W1[t];
  }
  __syncthreads();
  // ---- conv1 linear: H[i][c] = dinv[i] * sum_k x[i][k] W1[c][k]  (sequential fma chain over k) ----
  {
    int pf = 0;
    for (int base = 0; base < n; base += FG_SLOTS, ++pf) {
      const int i = base + slot;
      const bool act = i < n;
      float acc = 0.f;
      for (int k0 = 0; k0 < F; k0 += 32) {
        float xv;
        if (k0 == 0 && pf < FG_XPF) {
          xv = pf == 0 ? xpre[0] : (pf == 1 ? xpre[1] : (pf == 2 ? xpre[2] : xpre[3]));
        } else {
          xv = (act && k0 + c < F) ? xin[(size_t)(n0 + i) * F + k0 + c] : 0.f;
        }
        const int xi = __builtin_bit_cast(int, xv);
        const int kend = F - k0 < 32 ? F - k0 : 32;
        for (int k = 0; k < kend; ++k) {        // broadcast x[i][k0+k] of this half through an SGPR
          const float lo = __builtin_bit_cast(float, DG_RL(xi, k & 31));
          const float hi = __builtin_bit_cast(float, DG_RL(xi, 32 + (k & 31)));
          acc = fmaf(upper ? hi : lo, Wt[(k0 + k) * 32 + c], acc);
        }
      }
      if (act) H[i * FG_RS + c] = dv[i] * acc;
    }
  }
  dg_lds_barrier();
  FG_MARK(1);

  // ---- three 32-wide layers: LDS only, raw LDS barriers (global stores of x_l stay in flight) ----
#pragma unroll
  for (int layer = 0; layer < 3; ++layer) {
    float* xout = layer == 0 ? x1 : (layer == 1 ? x2 : x3);
    const float bc = prm[layer * 32 + c];
    const float w4c = prm[96 + c];
    // gather phase: half-wave per destination node, lane = channel; cooperative index fetch
    for (int base = 0; base < n; base += FG_SLOTS) {
      const int i = base + slot;
      const bool act = i < n;
      const int ii = act ? i : 0;
      const int start = act ? rp[ii] : 0, end = act ? rp[ii + 1] : 0;
      float acc;
      if (ldscol) {
        acc = dg_coop_gather32<true>(start, end, nz, c, upper, [&](int e) { return cl[e]; },
                                     [&](int j) { return H[j * FG_RS + c]; });
      } else {
        acc = dg_coop_gather32<true>(
            start, end, nz, c, upper,
            [&](int e) {
              const int jj = colidx[e] - n0;
              const bool ok = (unsigned)jj < (unsigned)n;
              if (!ok) bad = true;
              return ok ? jj : nz;
            },
            [&](int j) { return H[j * FG_RS + c]; });
      }
      acc += H[ii * FG_RS + c];
      float val = 0.f;
      if (act) {
        val = tanhf(fmaf(dv[i], acc, bc));
        X[i * FG_RS + c] = val;
        xout[(size_t)(n0 + i) * 32 + c] = val;
      }
      if (layer == 2) {     // conv4's linear (32 -> 1): per-channel products, fixed-order half-wave sum
        const float pacc = dg_half_sum(val * w4c);
        if (act && c == 0) h4s[i] = dv[i] * pacc;
      }
    }
    dg_lds_barrier();
    if (layer < 2) {
      // next layer's linear on MFMA: 16x16 blocks (tile, nb) of [n x 32] = X . Wn^T, written to H in place
      const int nb = wave & 1;
      const int tiles = (n + 15) >> 4;
      for (int tile = wave >> 1; tile < tiles; tile += 8) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        const int arow = tile * 16 + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const float a = arow < n ? X[arow * FG_RS + 4 * kk + (lane >> 4)] : 0.f;
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, layer == 0 ? wreg2[kk] : wreg3[kk], d, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = tile * 16 + (lane >> 4) * 4 + r;
          if (row < n) H[row * FG_RS + nb * 16 + (lane & 15)] = dv[row] * d[r];
        }
      }
      dg_lds_barrier();
    }
    FG_MARK(2 + layer);
  }

  // ---- conv4 aggregation (F = 1): half-wave per node, 32 neighbour values per fetch, sequential sum ----
  {
    const float b4s = prm[128];
    for (int base = 0; base < n; base += FG_SLOTS) {
      const int i = base + slot;
      const bool act = i < n;
      const int ii = act ? i : 0;
      const int start = act ? rp[ii] : 0, end = act ? rp[ii + 1] : 0;
      float s;
      if (ldscol) {
        s = dg_coop_gather1(start, end, c, upper, [&](int e) { return cl[e]; }, [&](int j) { return h4s[j]; });
      } else {
        s = dg_coop_gather1(
            start, end, c, upper,
            [&](int e) {
              const int jj = colidx[e] - n0;
              const bool ok = (unsigned)jj < (unsigned)n;
              if (!ok) bad = true;
              return ok ? jj : nz;
            },
            [&](int j) { return h4s[j]; });
      }
      s += h4s[ii];
      if (act && c == 0) {
        const float v4 = tanhf(fmaf(dv[i], s, b4s));
        x4s[i] = v4;
        x4[n0 + i] = v4;
      }
    }
  }
  if (bad) { err[1] = epoch; err[3] = ~epoch; }     // an edge left its graph: batch is not block-diagonal
  __syncthreads();      // full barrier: x1..x4 of this graph are complete and visible to this workgroup
  FG_MARK(5);

  // ---- SortPooling + dense tail (keys from LDS, rows from the slabs just written) ----
  const RdSmem M = dg_rd_carve(smem, small);
  dg_readout_fwd_body(M, b, n0, n, C, tw, x4s, 0, x1, x2, x3, x4, pooled, perm, a5g, a6g, a1dg, maskg, logp,
                      training, seed, dbg);
#undef FG_MARK
}

static thread_local unsigned long long* g_fg_dbg = nullptr;
void dg_fused_set_debug(unsigned long long* p) { g_fg_dbg = p; }

// choose how many neighbour ids to keep in LDS for (nmax, F, emax): all of them if they fit, else none
static int fg_choose_emax_lds(int nmax, int F, int emax) {
  if (emax > 0 && fg_lds_bytes(nmax, F, emax) <= FG_LDS_CAP) return emax;
  return 0;
}

int dg_launch_fused_fwd(int N, int B, int F, int C, int nmax, int emax, const float* params, const DgParams* pl,
                        const float* x, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                        const int32_t* graph_ptr, const int32_t* graph_eptr, float* x1, float* x2, float* x3, float* x4,
                        float* pooled, int32_t* perm, float* a5, float* a6, float* a1d, uint8_t* drop_mask, float* logp,
                        int training, uint64_t seed, int32_t* err, uint32_t epoch, hipStream_t s,
                        hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (N <= 0 || B <= 0 || nmax <= 0 || nmax > DGCNN_FUSED_MAX_NODES) return DGCNN_EINVAL;
  const int emax_lds = fg_choose_emax_lds(nmax, F, emax);
  const size_t r0 = fg_region0_bytes(nmax, F);
  const size_t lds = fg_lds_bytes(nmax, F, emax_lds);
  if (lds > FG_LDS_CAP) return DGCNN_EUNSUPPORTED;
  static bool attr_set = false;      // raise the dynamic-LDS cap once per process (idempotent)
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                            FG_LDS_CAP) != hipSuccess)
      return DGCNN_ELAUNCH;
    attr_set = true;
  }
  FgW gw;
  gw.W1 = params + pl->off[0]; gw.b1 = params + pl->off[1];
  gw.W2 = params + pl->off[2]; gw.b2 = params + pl->off[3];
  gw.W3 = params + pl->off[4]; gw.b3 = params + pl->off[5];
  gw.W4 = params + pl->off[6]; gw.b4 = params + pl->off[7];
  hipExtLaunchKernelGGL(k_fused_fwd, dim3(B), dim3(FG_THREADS), lds, s, ev_start, ev_stop, 0, F, C, nmax, emax_lds, r0,
                        gw, dg_tail_w(params, pl), x, rowptr, colidx, dinv, graph_ptr, graph_eptr, x1, x2, x3, x4, pooled, perm,
                        a5, a6, a1d, drop_mask, logp, training, seed, reinterpret_cast<unsigned int*>(err), epoch,
                        g_fg_dbg);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// largest per-graph node count the fused path accepts for F input features (0 = never)
int dg_fused_max_nodes(int F) {
  int best = 0;
  for (int nm = 16; nm <= DGCNN_FUSED_MAX_NODES; nm += 16)
    if (fg_lds_bytes(nm, F, 0) <= FG_LDS_CAP) best = nm;
  return best;
}
