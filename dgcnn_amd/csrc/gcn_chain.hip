// gcn_chain.hip -- the four graph convolutions of one graph as ONE chain inside ONE workgroup (gfx950).
//
// Same layer arithmetic as gcn.hip / gcn_dense.hip (PyG GCNConv + tanh, /root/reference/model.py:13-16,30-33):
//     x_l[i] = tanh( dinv[i] * sum_{j in N(i)+{i}} hs_l[j] + b_l ),      hs_l[j] = dinv[j] * (x_{l-1} W_l^T)[j]
// evaluated per graph as dense block products (A+I)_g . HS_g on the bf16 matrix cores from the bit-packed adjacency
// (exact in fp32 through the three-way bf16 split, see gcn_dense.hip), but the pre-scaled linear outputs hs_2, hs_3,
// h4s NEVER LEAVE THE CU: a workgroup owns a graph, keeps hs in LDS, and walks conv1 -> conv2 -> conv3 -> conv4.
// Per graph and layer the per-layer kernels pay a global round trip (rows of hs written by one launch, re-read, split
// and staged by the next: ~10 k cycles per work item, profiles/r02); here a layer is LDS reads -> MFMAs -> epilogue.
// HBM traffic: xs, bitmap, dinv in; ax, x1, x2, x3, x4 out (saved for backward and SortPooling).
//
// Everything is computed TRANSPOSED so that no operand ever needs an LDS round trip to change layout:
//   * block product  out^T[c][m] = sum_k HS[k][c] * Adj[m][k]  on v_mfma_f32_16x16x32_bf16 with A := HS^T, B := Adj^T.
//     The accumulator lane (nl = lane & 15, kq = lane >> 4) then holds, for node m = nl of the tile, the FOUR CONSECUTIVE
//     output columns 4kq .. 4kq+3 (+16 for the second block): a 16-byte piece of the node's row -- row stores go straight
//     from registers to global memory, 16 B per lane.
//   * HS^T operand: hs lives in LDS ROW-MAJOR as three bf16 parts, [part][16-column plane][k][16] (32-byte rows), and is
//     read with ds_read_b64_tr_b16 (hardware transpose read, tools/probes/tr16_probe.hip): two reads deliver the eight
//     k-values of one column.  32 lanes read 8 consecutive rows = 256 contiguous bytes: conflict-free.  The sum over k
//     is order-free, so lane group kg takes k = 4kg..4kg+3 and 16+4kg..16+4kg+3 of each 32-row word; the adjacency
//     operand takes the matching two nibbles of the bitmap word.
//   * next layer's linear step  hs^T[o][m] = sum_k W[o][k] * x[m][k]  on v_mfma_f32_16x16x4_f32 with A := W, B := x^T:
//     step s of lane group kq consumes k = 16(s>>2) + 4kq + (s&3) -- exactly the activated values the lane already holds.
//     The result lane again holds four consecutive columns of its node: split into three bf16 parts it is ONE 8-byte
//     LDS store per part and plane into the next layer's row-major image (slot swizzled by (k>>2)&3: conflict-free).
//   * conv4 (32 -> 1): the three parts of h4s are three COLUMNS of one B operand; one MFMA per bitmap word.
// Results: deterministic, independent of batch composition (a graph's numbers depend on its own rows only); they differ
// from the per-layer kernels by fp32 rounding (different k order inside the fp32 matrix instruction).
#include "dg_common.h"
#include "dg_prep.h"
#include <hip/hip_ext.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define CH_LDS __attribute__((address_space(3)))

struct ChW { const float *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4; };

__device__ __forceinline__ void ch_split3(float h, unsigned& p0, unsigned& p1, unsigned& p2) {      // h = p0 + p1 + p2, exact
  const unsigned u0 = __float_as_uint(h) & 0xffff0000u;
  const float r1 = h - __uint_as_float(u0);
  const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(u1);
  p0 = u0 >> 16; p1 = u1 >> 16; p2 = __float_as_uint(r2) >> 16;
}

template <int WAVES, int TPW, bool PING>
struct ChCfg {
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int ROWS = 16 * WAVES * TPW;        // node bound of the class
  static constexpr int KW = ROWS / 32;                 // bitmap words per row
  static constexpr int PS = ROWS * 32;                 // bytes of one (part, plane): ROWS rows of 16 bf16
  static constexpr int BUF = 6 * PS;                   // three parts x two planes
  static constexpr int NBUF = PING ? 2 : 1;
  static constexpr int OFF_W1 = NBUF * BUF;            // operand-order weight tables [2 ob][8 s][64 lanes] fp32
  static constexpr int OFF_W2 = OFF_W1 + 4096;
  static constexpr int OFF_W3 = OFF_W2 + 4096;
  static constexpr int OFF_BT = OFF_W3 + 4096;         // b1 | b2 | b3 | w4  (4 x 32 floats)
  static constexpr int OFF_DV = OFF_BT + 512;          // dinv of the graph's nodes, 0 beyond n  [ROWS]
  static constexpr int OFF_H4 = OFF_DV + 4 * ROWS;     // h4s as three bf16 parts [3][ROWS]
  static constexpr int OFF_TAB = OFF_H4 + 6 * ROWS;    // nibble -> four bf16
  static constexpr int TOTAL = OFF_TAB + 128;
  static constexpr int STAGE_IT = (ROWS * 8 + THREADS - 1) / THREADS;     // xs staging items per thread (2 planes x 4 slots)
};

// one (part, plane) operand of a 32-row word: two transpose reads = the lane's eight k-values of its column
__device__ __forceinline__ bf16x8 ch_read_hsT(const char* p) {
  const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 CH_LDS*)(p));
  const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 CH_LDS*)(p + 512));       // rows +16
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8 ch_bits_operand(unsigned w, int kg, const uint2* __restrict__ tab) {
  const uint2 lo = tab[(w >> (4 * kg)) & 15u], hi = tab[(w >> (16 + 4 * kg)) & 15u];
  bf16x8 b;
  unsigned* bu = reinterpret_cast<unsigned*>(&b);
  bu[0] = lo.x; bu[1] = lo.y; bu[2] = hi.x; bu[3] = hi.y;
  return b;
}

template <int WAVES, int TPW, bool PING>
__global__ void __launch_bounds__(64 * WAVES)
k_chain_fwd(int N, int F, const int* __restrict__ graph_ptr, const unsigned* __restrict__ bits, const float* __restrict__ dinv,
            const float* __restrict__ xs, ChW gw, float* __restrict__ axg, float* __restrict__ x1, float* __restrict__ x2,
            float* __restrict__ x3, float* __restrict__ x4, int nmin, const int* __restrict__ sched, const int* __restrict__ nbig_p) {
  using C = ChCfg<WAVES, TPW, PING>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  // sched != null: block b takes entry b of the schedule's large-graph range [0, nbig) (dg_prep.h); else graph b
  int n0, n;
  if (sched) {
    if ((int)blockIdx.x >= nbig_p[0]) return;
    n0 = sched[2 * blockIdx.x]; n = sched[2 * blockIdx.x + 1];
  } else {
    n0 = graph_ptr[blockIdx.x]; n = graph_ptr[blockIdx.x + 1] - n0;
  }
  if (n <= nmin || n > C::ROWS) return;                 // another size class' launch owns this graph (or: flagged by graph prep)
  const int K32 = (n + 31) >> 5, T = (n + 15) >> 4, RU = 32 * K32;
  const int S = 1 << dgd_class(n);
  const int NBF = F > 16 ? 2 : 1;

  char* H0 = smem;
  char* H1 = PING ? smem + C::BUF : smem;
  float* W1op = reinterpret_cast<float*>(smem + C::OFF_W1);
  float* W2op = reinterpret_cast<float*>(smem + C::OFF_W2);
  float* W3op = reinterpret_cast<float*>(smem + C::OFF_W3);
  float* bt = reinterpret_cast<float*>(smem + C::OFF_BT);
  float* dv = reinterpret_cast<float*>(smem + C::OFF_DV);
  unsigned short* h4p = reinterpret_cast<unsigned short*>(smem + C::OFF_H4);
  uint2* tab = reinterpret_cast<uint2*>(smem + C::OFF_TAB);

  // ---- prologue: every global load of the graph in flight at once ---------------------------------------------------
  unsigned wb[TPW][C::KW];
  {
    const unsigned* bp = bits + (size_t)N * (S - 1) + (size_t)n0 * S;
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int m = 16 * (wave + ti * WAVES) + nl;
#pragma unroll
      for (int u = 0; u < C::KW; ++u) wb[ti][u] = (m < n && u < K32) ? bp[(size_t)m * S + u] : 0u;
    }
  }
  float sv[C::STAGE_IT][4];
#pragma unroll
  for (int j = 0; j < C::STAGE_IT; ++j) {               // conv1 operand xs = dinv * x: item (plane, row k, 4-column slot q)
    const int it = tid + j * C::THREADS;
    const int nb = it >= 4 * RU ? 1 : 0, rem = it - nb * 4 * RU, k = rem >> 2, q = rem & 3;
    const bool ok = it < 4 * RU * NBF && k < n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 16 * nb + 4 * q + i;
      sv[j][i] = (ok && c < F) ? xs[(size_t)(n0 + k) * F + c] : 0.f;
    }
  }
  for (int k = tid; k < C::ROWS; k += C::THREADS) dv[k] = k < n ? dinv[n0 + k] : 0.f;
  for (int e = tid; e < 1024; e += C::THREADS) {        // weights in MFMA-operand order: [ob][s][lane] = W[16ob + (lane&15)][kappa(s, lane>>4)]
    const int ob = e >> 9, s = (e >> 6) & 7, l = e & 63;
    const int kap = 16 * (s >> 2) + 4 * (l >> 4) + (s & 3), o = 16 * ob + (l & 15);
    W2op[e] = gw.W2[o * 32 + kap];
    W3op[e] = gw.W3[o * 32 + kap];
    W1op[e] = kap < F ? gw.W1[o * F + kap] : 0.f;
  }
  if (tid < 128) {
    const int which = tid >> 5, idx = tid & 31;
    const float* src = which == 0 ? gw.b1 : (which == 1 ? gw.b2 : (which == 2 ? gw.b3 : gw.W4));
    bt[tid] = src[idx];
  }
  const float b4s = gw.b4[0];
  if (tid < 16) tab[tid] = make_uint2(((tid & 1) ? 0x3f80u : 0u) | ((tid & 2) ? 0x3f800000u : 0u),
                                      ((tid & 4) ? 0x3f80u : 0u) | ((tid & 8) ? 0x3f800000u : 0u));
  {   // rows 16T .. RU-1 (at most 16) are read by the block products but written by no tile: zero them once
    const int gr = RU - 16 * T;
    for (int it = tid; it < C::NBUF * 6 * gr * 2; it += C::THREADS) {
      const int piece = it & 1, row = 16 * T + ((it >> 1) % gr), pp = (it >> 1) / gr;       // pp: (buffer, part, plane)
      *reinterpret_cast<uint4*>(smem + (size_t)pp * C::PS + row * 32 + 16 * piece) = make_uint4(0u, 0u, 0u, 0u);
    }
    for (int it = tid; it < 3 * gr; it += C::THREADS) h4p[(it / gr) * C::ROWS + 16 * T + (it % gr)] = 0;
  }
  // xs -> H0 (all rows < RU of the planes conv1 reads, zeros where k >= n or column >= F)
#pragma unroll
  for (int j = 0; j < C::STAGE_IT; ++j) {
    const int it = tid + j * C::THREADS;
    const int nb = it >= 4 * RU ? 1 : 0, rem = it - nb * 4 * RU, k = rem >> 2, q = rem & 3;
    if (it < 4 * RU * NBF) {
      unsigned qq[3][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ch_split3(sv[j][i], qq[0][i], qq[1][i], qq[2][i]);
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint2*>(H0 + (p * 2 + nb) * C::PS + k * 32 + 8 * (q ^ ((k >> 2) & 3))) =
            make_uint2(qq[p][0] | (qq[p][1] << 16), qq[p][2] | (qq[p][3] << 16));
    }
  }
  __syncthreads();

  // lane constants of the LDS addressing
  const int rdoff = (4 * kq + (nl >> 2)) * 32 + 8 * ((nl & 3) ^ kq);        // transpose reads: row 4kg + i/4, slot (i%4) ^ kg
  float dn[TPW];
  int wroff[TPW];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    const int m = 16 * (wave + ti * WAVES) + nl;
    dn[ti] = dv[m];
    wroff[ti] = m * 32 + 8 * (kq ^ ((nl >> 2) & 3));
  }
  const float4 bz = make_float4(0.f, 0.f, 0.f, 0.f);

  // block product of tile ti with the hs image Hc: acc[nb] (nb < NBP planes)
  auto product = [&](const char* Hc, int ti, int NBP, f32x4 (&acc)[2]) {
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* hp = Hc + rdoff;
#pragma unroll
    for (int u = 0; u < C::KW; ++u) {
      if (u < K32) {
        const unsigned w = wb[ti][u];
        if (__builtin_amdgcn_ballot_w64(w != 0u) != 0ull) {
          const bf16x8 bop = ch_bits_operand(w, kq, tab);
          if (NBP == 2) {
            bf16x8 a[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int nb = 0; nb < 2; ++nb) a[p][nb] = ch_read_hsT(hp + (p * 2 + nb) * C::PS + u * 1024);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][nb], bop, acc[nb], 0, 0, 0);
          } else {
            bf16x8 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = ch_read_hsT(hp + (p * 2) * C::PS + u * 1024);
#pragma unroll
            for (int p = 0; p < 3; ++p) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p], bop, acc[0], 0, 0, 0);
          }
        }
      }
    }
  };
  // activated tile v (this lane: node m, columns 4kq..4kq+3 and 16+4kq..) -> its rows to global, then the next layer's
  // pre-scaled linear output hs = dn * (v W^T) as registers (same layout)
  auto rows_and_linear = [&](const f32x4 (&v)[2], int ti, float* __restrict__ xout, const float* __restrict__ Wop, f32x4 (&hs)[2]) {
    const int m = 16 * (wave + ti * WAVES) + nl;
    if (m < n) {
      float* dst = xout + (size_t)(n0 + m) * 32 + 4 * kq;
      *reinterpret_cast<float4*>(dst) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
      *reinterpret_cast<float4*>(dst + 16) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
    }
    float wv[2][8];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int s = 0; s < 8; ++s) wv[ob][s] = Wop[(ob * 8 + s) * 64 + lane];
    f32x4 d2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) d2[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob][s], v[s >> 2][s & 3], d2[ob], 0, 0, 0);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) hs[ob][r] = dn[ti] * d2[ob][r];
  };
  auto write_hs = [&](char* Hn, int ti, const f32x4 (&hs)[2]) {
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      unsigned qq[3][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ch_split3(hs[ob][r], qq[0][r], qq[1][r], qq[2][r]);
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint2*>(Hn + (p * 2 + ob) * C::PS + wroff[ti]) =
            make_uint2(qq[p][0] | (qq[p][1] << 16), qq[p][2] | (qq[p][3] << 16));
    }
  };
  auto bias4 = [&](int which, int nb) { return *reinterpret_cast<const float4*>(bt + 32 * which + 16 * nb + 4 * kq); };

  f32x4 hsv[TPW][2];
  // ---- conv1, aggregate-first: ax = dn * (Adj . xs) (saved), x1 = tanh(ax W1^T + b1), hs2 = dn * (x1 W2^T) ------------
  {
    const float4 b0 = bias4(0, 0), b1v = bias4(0, 1);
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int t = wave + ti * WAVES, m = 16 * t + nl;
      if (t < T) {
        f32x4 acc[2];
        product(H0, ti, NBF, acc);
        f32x4 axv[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            axv[nb][r] = dn[ti] * acc[nb][r];
            const int c = 16 * nb + 4 * kq + r;
            if (c < F && m < n) axg[(size_t)(n0 + m) * F + c] = axv[nb][r];
          }
        f32x4 pre[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 8; ++s)
          if (16 * (s >> 2) + (s & 3) < F) {             // (uniform: the step's smallest k)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
              pre[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(W1op[(ob * 8 + s) * 64 + lane], axv[s >> 2][s & 3], pre[ob], 0, 0, 0);
          }
        f32x4 v[2];
        v[0][0] = dg_tanh(pre[0][0] + b0.x); v[0][1] = dg_tanh(pre[0][1] + b0.y);
        v[0][2] = dg_tanh(pre[0][2] + b0.z); v[0][3] = dg_tanh(pre[0][3] + b0.w);
        v[1][0] = dg_tanh(pre[1][0] + b1v.x); v[1][1] = dg_tanh(pre[1][1] + b1v.y);
        v[1][2] = dg_tanh(pre[1][2] + b1v.z); v[1][3] = dg_tanh(pre[1][3] + b1v.w);
        rows_and_linear(v, ti, x1, W2op, hsv[ti]);
      }
    }
    if (!PING) dg_lds_barrier();
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
      if (wave + ti * WAVES < T) write_hs(H1, ti, hsv[ti]);
    dg_lds_barrier();
  }
  // ---- conv2: from H1; hs3 = dn * (x2 W3^T) -> H0 ---------------------------------------------------------------------
  auto layer32 = [&](const char* Hc, int which, const f32x4& accA, const f32x4& accB, f32x4 (&v)[2], float dnv) {
    const float4 b0 = bias4(which, 0), b1v = bias4(which, 1);
    v[0][0] = dg_tanh(fmaf(dnv, accA[0], b0.x)); v[0][1] = dg_tanh(fmaf(dnv, accA[1], b0.y));
    v[0][2] = dg_tanh(fmaf(dnv, accA[2], b0.z)); v[0][3] = dg_tanh(fmaf(dnv, accA[3], b0.w));
    v[1][0] = dg_tanh(fmaf(dnv, accB[0], b1v.x)); v[1][1] = dg_tanh(fmaf(dnv, accB[1], b1v.y));
    v[1][2] = dg_tanh(fmaf(dnv, accB[2], b1v.z)); v[1][3] = dg_tanh(fmaf(dnv, accB[3], b1v.w));
  };
  {
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
      if (wave + ti * WAVES < T) {
        f32x4 acc[2], v[2];
        product(H1, ti, 2, acc);
        layer32(H1, 1, acc[0], acc[1], v, dn[ti]);
        rows_and_linear(v, ti, x2, W3op, hsv[ti]);
      }
    if (!PING) dg_lds_barrier();
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
      if (wave + ti * WAVES < T) write_hs(H0, ti, hsv[ti]);
    dg_lds_barrier();
  }
  // ---- conv3: from H0; the next linear step is 32 -> 1: h4s = dn * (x3 . w4) -> three bf16 parts -------------------------
  {
    const float4 w0 = bias4(3, 0), w1 = bias4(3, 1);
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int t = wave + ti * WAVES, m = 16 * t + nl;
      if (t < T) {
        f32x4 acc[2], v[2];
        product(H0, ti, 2, acc);
        layer32(H0, 2, acc[0], acc[1], v, dn[ti]);
        if (m < n) {
          float* dst = x3 + (size_t)(n0 + m) * 32 + 4 * kq;
          *reinterpret_cast<float4*>(dst) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
          *reinterpret_cast<float4*>(dst + 16) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
        }
        float p = v[0][0] * w0.x;
        p = fmaf(v[0][1], w0.y, p); p = fmaf(v[0][2], w0.z, p); p = fmaf(v[0][3], w0.w, p);
        p = fmaf(v[1][0], w1.x, p); p = fmaf(v[1][1], w1.y, p); p = fmaf(v[1][2], w1.z, p); p = fmaf(v[1][3], w1.w, p);
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        if (kq == 0) {
          unsigned q0, q1, q2;
          ch_split3(dn[ti] * p, q0, q1, q2);
          h4p[m] = (unsigned short)q0; h4p[C::ROWS + m] = (unsigned short)q1; h4p[2 * C::ROWS + m] = (unsigned short)q2;
        }
      }
    }
    dg_lds_barrier();
  }
  // ---- conv4 (32 -> 1): the three parts of h4s are columns 0..2 of ONE B operand; x4 = tanh(dinv * sum + b4) ----------
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    const int t = wave + ti * WAVES;
    if (t < T) {
      f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
      const unsigned short* hq = h4p + min(nl, 2) * C::ROWS + 4 * kq;
#pragma unroll
      for (int u = 0; u < C::KW; ++u) {
        if (u < K32) {
          const unsigned w = wb[ti][u];
          if (__builtin_amdgcn_ballot_w64(w != 0u) != 0ull) {
            const bf16x8 aop = ch_bits_operand(w, kq, tab);
            uint2 lo = *reinterpret_cast<const uint2*>(hq + 32 * u), hi = *reinterpret_cast<const uint2*>(hq + 32 * u + 16);
            if (nl >= 3) { lo = make_uint2(0u, 0u); hi = make_uint2(0u, 0u); }
            bf16x8 bop;
            unsigned* bu = reinterpret_cast<unsigned*>(&bop);
            bu[0] = lo.x; bu[1] = lo.y; bu[2] = hi.x; bu[3] = hi.y;
            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aop, bop, a4, 0, 0, 0);
          }
        }
      }
      // lane (j = nl, kq) holds part j's sums for rows 4kq + r: total = (part0 + part1) + part2, gathered in lane j = 0
      float tot[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = __shfl_xor(a4[r], 1), s2 = __shfl_xor(a4[r], 2);      // lane j = 0 receives lanes 1 and 2
        tot[r] = (a4[r] + s1) + s2;       // (a DPP quad_perm form of this was mis-compiled by ROCm 7.2: one DPP move reused for all r)
      }
      if (nl == 0) {
        const int mm = 16 * t + 4 * kq;
        const float4 dq = *reinterpret_cast<const float4*>(dv + mm);
        const float dd[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (mm + r < n) x4[n0 + mm + r] = dg_tanh(fmaf(dd[r], tot[r], b4s));
      }
    }
  }
  (void)bz;
}

// =================================================================================================================
// PERSISTENT form for graphs of <= 128 nodes (the bulk of every TU-shaped batch): workgroups stay resident, stage the
// weight tables ONCE, and deal themselves graphs from the SCHEDULE graph preparation left behind (dg_prep.h: graphs ranked
// by tile count, largest first; workgroup w of G takes ranks w, 2G-1-w, 2G+w, ... -- snake order, equal sums, no atomics,
// and the same assignment every run).  Everything a graph needs from global memory (its block of the bitmap, dinv, xs)
// is requested one graph AHEAD into a handful of registers and lands while the current graph's four layers run; the
// schedule entry is requested two graphs ahead.  Every one of these loads is UNCONDITIONAL on a clamped address and is
// selected when consumed: behind a per-lane `if` the compiler parks the result in a temporary and waits for it on the
// spot (a returning atomic per graph, tried first, was waited for with vmcnt(0) right where it was issued).  A graph's
// life inside the workgroup: registers -> LDS images (one barrier) -> conv1..conv4 (three barriers).
// =================================================================================================================
#ifdef CH_TIMING
#define CH_T(k) do { if (dbg && tid == 0) { const unsigned long long now_ = clock64(); dbg[blockIdx.x * 16 + (k)] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define CH_T(k) do { } while (0)
#endif
struct ChP {
  static constexpr int THREADS = 512, ROWS = 128, KW = 4, PS = ROWS * 32, BUF = 6 * PS;
  static constexpr int OFF_W1 = 2 * BUF, OFF_W2 = OFF_W1 + 4096, OFF_W3 = OFF_W2 + 4096, OFF_BT = OFF_W3 + 4096;
  static constexpr int OFF_DV = OFF_BT + 512;            // two sets (graph parity) of: dinv [ROWS] f32
  static constexpr int OFF_H4 = OFF_DV + 2 * 4 * ROWS;   //                           h4s parts [3][ROWS] bf16
  static constexpr int OFF_BL = OFF_H4 + 2 * 6 * ROWS;   //                           bitmap rows [ROWS][4] u32
  static constexpr int OFF_TAB = OFF_BL + 2 * 16 * ROWS;
  static constexpr int OFF_CTL = OFF_TAB + 128;
  static constexpr int TOTAL = OFF_CTL + 64;
};

template <int XI>      // xs items (row, 4-column slot) per thread: 1 covers F <= 16, 2 covers F <= 32
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4)))       // <= 128 registers: two workgroups per CU
k_chain_fwd_p(int N, int B, int F, const int* __restrict__ sched, const int* __restrict__ nbig_p, const unsigned* __restrict__ bits,
              const float* __restrict__ dinv, const float* __restrict__ xs, ChW gw, float* __restrict__ axg,
              float* __restrict__ x1, float* __restrict__ x2, float* __restrict__ x3, float* __restrict__ x4,
              unsigned long long* __restrict__ dbg) {
  using C = ChP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  char* H0 = smem;
  char* H1 = smem + C::BUF;
  float* W1op = reinterpret_cast<float*>(smem + C::OFF_W1);
  float* W2op = reinterpret_cast<float*>(smem + C::OFF_W2);
  float* W3op = reinterpret_cast<float*>(smem + C::OFF_W3);
  float* bt = reinterpret_cast<float*>(smem + C::OFF_BT);
  uint2* tab = reinterpret_cast<uint2*>(smem + C::OFF_TAB);
  int* ctl = reinterpret_cast<int*>(smem + C::OFF_CTL);
#ifdef CH_TIMING
  unsigned long long tprev_ = clock64();
  if (dbg && tid == 0) for (int k = 0; k < 16; ++k) dbg[blockIdx.x * 16 + k] = 0;
#endif
  // ---- schedule: entries [nbig, B) are the graphs of this size class; entry of round r for this workgroup -----------------
  const int nbig = nbig_p[0], ns = B - nbig, G = (int)gridDim.x, w = (int)blockIdx.x;
  auto entry_of = [&](int r) {               // {n0, n}; n = 0 past the end (load clamped, selected)
    const int li = r * G + ((r & 1) ? G - 1 - w : w);
    const int2 e = *reinterpret_cast<const int2*>(sched + 2 * (nbig + min(li, max(ns - 1, 0))));
    return make_int2(e.x, (li < ns && e.y <= ChP::ROWS) ? e.y : 0);
  };
  int2 eC = entry_of(0), eN = entry_of(1);
  // ---- once per workgroup: weight tables in MFMA-operand order, biases, nibble table -----------------------------------
  // (COALESCED loads -- thread e takes element e -- scattered into operand order [ob][s][lane] on the LDS side: 512
  //  workgroups gathering the same 12 KB in operand order at the same moment took 17 us of set-up)
  {
    float w2[2], w3[2], w1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + 512 * j;
      w2[j] = gw.W2[e]; w3[j] = gw.W3[e]; w1[j] = e < 32 * F ? gw.W1[e] : 0.f;
    }
    float bv = 0.f;
    if (tid < 128) {
      const int which = tid >> 5, idx = tid & 31;
      const float* src = which == 0 ? gw.b1 : (which == 1 ? gw.b2 : (which == 2 ? gw.b3 : gw.W4));
      bv = src[idx];
    }
    auto slot_of = [](int o, int k) {       // W[o][k] -> [ob = o >> 4][s = 4 (k >> 4) + (k & 3)][lane = (o & 15) + 16 ((k >> 2) & 3)]
      return (((o >> 4) * 8 + ((k >> 4) << 2) + (k & 3)) << 6) + (o & 15) + (((k >> 2) & 3) << 4);
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) {           // conv1's table: entries with k >= F are zero (written here, by destination index)
      const int e = tid + 512 * j, s_ = (e >> 6) & 7, l = e & 63;
      if (16 * (s_ >> 2) + 4 * (l >> 4) + (s_ & 3) >= F) W1op[e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + 512 * j;
      W2op[slot_of(e >> 5, e & 31)] = w2[j]; W3op[slot_of(e >> 5, e & 31)] = w3[j];
      if (e < 32 * F) { const int o = e / F; W1op[slot_of(o, e - o * F)] = w1[j]; }
    }
    if (tid < 128) bt[tid] = bv;
    if (tid < 16) tab[tid] = make_uint2(((tid & 1) ? 0x3f80u : 0u) | ((tid & 2) ? 0x3f800000u : 0u),
                                        ((tid & 4) ? 0x3f80u : 0u) | ((tid & 8) ? 0x3f800000u : 0u));
  }
  const float b4s = gw.b4[0];
  __syncthreads();
  int n0 = __builtin_amdgcn_readfirstlane(eC.x), n = __builtin_amdgcn_readfirstlane(eC.y);
  const int lg = F <= 4 ? 0 : (F <= 8 ? 1 : (F <= 16 ? 2 : 3));      // 4-column slots with data per row: 2^lg
  const int NBF = F > 16 ? 2 : 1;
  // prefetch registers of the graph about to be staged
  unsigned pbit = 0u; float pdv = 0.f; float pxs[XI][4];
  // (loads are issued UNCONDITIONALLY on clamped addresses and selected when they are consumed: behind a per-lane `if` the
  //  compiler loads into temporaries and waits for them on the spot -- the prefetch would not be one)
  auto prefetch = [&](int pn0, int pn) {
    const int pS = 1 << dgd_class(max(pn, 1));
    pbit = bits[(size_t)N * (pS - 1) + (size_t)pn0 * pS + min(tid, max(pn * pS - 1, 0))];
    pdv = dinv[pn0 + min(tid, max(pn - 1, 0))];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int it = tid + 512 * j, k = min(it >> lg, max(pn - 1, 0)), qq = it & ((1 << lg) - 1);
      const float* xr = xs + (size_t)(pn0 + k) * F;
#pragma unroll
      for (int i = 0; i < 4; ++i) pxs[j][i] = xr[min(4 * qq + i, F - 1)];
    }
  };
  prefetch(n0, n);
  const int rdoff = (4 * kq + (nl >> 2)) * 32 + 8 * ((nl & 3) ^ kq);
  const int mrow = 16 * wave + nl;
  const int wroff = mrow * 32 + 8 * (kq ^ ((nl >> 2) & 3));
  int par = 0;
  CH_T(0);                                                // 0: set-up (tables, first tickets)
  for (int r = 0; r * G < ns; ++r) {
    float* dv = reinterpret_cast<float*>(smem + C::OFF_DV) + par * C::ROWS;
    unsigned short* h4p = reinterpret_cast<unsigned short*>(smem + C::OFF_H4) + par * 3 * C::ROWS;
    unsigned* bl = reinterpret_cast<unsigned*>(smem + C::OFF_BL) + par * 4 * C::ROWS;
    const int K32 = (n + 31) >> 5, T = (n + 15) >> 4, RU = 32 * K32;
    const int S = 1 << dgd_class(max(n, 1));
    // ---- stage the graph: registers -> LDS images ---------------------------------------------------------------------
    if (tid < n * S) bl[tid] = pbit;                      // bitmap rows, stride S words
    if (tid < C::ROWS) dv[tid] = tid < n ? pdv : 0.f;     // (0 beyond n)
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int it = tid + 512 * j, k = it >> lg, qq = it & ((1 << lg) - 1);
      if (k < n) {
        unsigned sp[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ch_split3(4 * qq + i < F ? pxs[j][i] : 0.f, sp[0][i], sp[1][i], sp[2][i]);
        const int nb = qq >> 2, sl = qq & 3;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint2*>(H0 + (p * 2 + nb) * C::PS + k * 32 + 8 * (sl ^ ((k >> 2) & 3))) =
              make_uint2(sp[p][0] | (sp[p][1] << 16), sp[p][2] | (sp[p][3] << 16));
      }
    }
    // zeros conv1 reads but nobody wrote: rows n..RU-1 of its planes, and the slots beyond the feature width
    for (int it = tid; it < RU * 4 * NBF; it += 512) {
      const int k = it / (4 * NBF), qq = it - k * 4 * NBF;
      if (!(k < n && qq < (1 << lg))) {
        const int nb = qq >> 2, sl = qq & 3;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint2*>(H0 + (p * 2 + nb) * C::PS + k * 32 + 8 * (sl ^ ((k >> 2) & 3))) = make_uint2(0u, 0u);
      }
    }
    {   // rows 16T .. RU-1 of every later image (H1 whole, H0's planes conv1 does not use, h4s): written by no tile
      const int gr = RU - 16 * T;
      for (int it = tid; it < 12 * gr * 2; it += 512) {
        const int piece = it & 1, row = 16 * T + ((it >> 1) % gr), pp = (it >> 1) / gr;
        if (pp >= 6 || (pp & 1) >= NBF)
          *reinterpret_cast<uint4*>(smem + (size_t)pp * C::PS + row * 32 + 16 * piece) = make_uint4(0u, 0u, 0u, 0u);
      }
      for (int it = tid; it < 3 * gr; it += 512) h4p[(it / gr) * C::ROWS + 16 * T + (it % gr)] = 0;
    }
    CH_T(1);                                              // 1: wait for the prefetched data + staging stores
    dg_lds_barrier();
    CH_T(2);                                              // 2: barrier
    const int n0N = __builtin_amdgcn_readfirstlane(eN.x), nN = __builtin_amdgcn_readfirstlane(eN.y);     // (requested a graph ago)
    eN = entry_of(r + 2);
    prefetch(n0N, nN);
    unsigned wb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wb[u] = (mrow < n && u < K32) ? bl[mrow * S + u] : 0u;
    const float dn = dv[mrow];
    CH_T(3);                                              // 3: issue of the next graph's loads
    const bool live = wave < T;
    // block product of this wave's tile with the image Hc.  Straight-line code per word count (no per-word branch: the
    // reads of word u+1 are scheduled under the matrix instructions of word u)
    auto productK = [&](auto kc, auto nbc, const char* Hc, f32x4 (&acc)[2]) {
      constexpr int KC = decltype(kc)::value, NBP = decltype(nbc)::value;
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* hp = Hc + rdoff;
      bf16x8 bop[KC];
#pragma unroll
      for (int u = 0; u < KC; ++u) bop[u] = ch_bits_operand(wb[u], kq, tab);
#pragma unroll
      for (int u = 0; u < KC; ++u) {
        bf16x8 a[3][NBP];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int nb = 0; nb < NBP; ++nb) a[p][nb] = ch_read_hsT(hp + (p * 2 + nb) * C::PS + u * 1024);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int nb = 0; nb < NBP; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][nb], bop[u], acc[nb], 0, 0, 0);
      }
    };
    auto product = [&](const char* Hc, int NBP, f32x4 (&acc)[2]) {
      using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
      if (NBP == 2) {
        if (K32 == 1) productK(I1{}, I2{}, Hc, acc); else if (K32 == 2) productK(I2{}, I2{}, Hc, acc);
        else if (K32 == 3) productK(I3{}, I2{}, Hc, acc); else productK(I4{}, I2{}, Hc, acc);
      } else {
        if (K32 == 1) productK(I1{}, I1{}, Hc, acc); else if (K32 == 2) productK(I2{}, I1{}, Hc, acc);
        else if (K32 == 3) productK(I3{}, I1{}, Hc, acc); else productK(I4{}, I1{}, Hc, acc);
      }
    };
    auto rows_and_linear = [&](const f32x4 (&v)[2], float* __restrict__ xout, const float* __restrict__ Wop, char* Hn) {
      if (mrow < n) {
        float* dst = xout + (size_t)(n0 + mrow) * 32 + 4 * kq;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
        *reinterpret_cast<float4*>(dst + 16) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
      }
      float wv[2][8];
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int s = 0; s < 8; ++s) wv[ob][s] = Wop[(ob * 8 + s) * 64 + lane];
      // four independent accumulator chains (k-steps of the first / second 16 input columns), combined at the end: the
      // dependent-accumulator latency of the fp32 matrix instruction is 40 cycles, its issue interval 32
      f32x4 d2[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ob = 0; ob < 2; ++ob)
            d2[ob][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob][4 * h + s], v[h][s], d2[ob][h], 0, 0, 0);
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        unsigned sp[3][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ch_split3(dn * (d2[ob][0][r] + d2[ob][1][r]), sp[0][r], sp[1][r], sp[2][r]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint2*>(Hn + (p * 2 + ob) * C::PS + wroff) = make_uint2(sp[p][0] | (sp[p][1] << 16), sp[p][2] | (sp[p][3] << 16));
      }
    };
    auto bias4 = [&](int which, int nb) { return *reinterpret_cast<const float4*>(bt + 32 * which + 16 * nb + 4 * kq); };
    auto act32 = [&](int which, const f32x4 (&acc)[2], f32x4 (&v)[2]) {
      const float4 b0 = bias4(which, 0), b1v = bias4(which, 1);
      v[0][0] = dg_tanh(fmaf(dn, acc[0][0], b0.x)); v[0][1] = dg_tanh(fmaf(dn, acc[0][1], b0.y));
      v[0][2] = dg_tanh(fmaf(dn, acc[0][2], b0.z)); v[0][3] = dg_tanh(fmaf(dn, acc[0][3], b0.w));
      v[1][0] = dg_tanh(fmaf(dn, acc[1][0], b1v.x)); v[1][1] = dg_tanh(fmaf(dn, acc[1][1], b1v.y));
      v[1][2] = dg_tanh(fmaf(dn, acc[1][2], b1v.z)); v[1][3] = dg_tanh(fmaf(dn, acc[1][3], b1v.w));
    };
    // ---- conv1 (aggregate-first) ---------------------------------------------------------------------------------------
    if (live) {
      const float4 b0 = bias4(0, 0), b1v = bias4(0, 1);
      f32x4 acc[2];
      product(H0, NBF, acc);
      f32x4 axv[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          axv[nb][r] = dn * acc[nb][r];
          const int c = 16 * nb + 4 * kq + r;
          if (c < F && mrow < n) axg[(size_t)(n0 + mrow) * F + c] = axv[nb][r];
        }
      f32x4 pre[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (16 * (s >> 2) + (s & 3) < F) {
#pragma unroll
          for (int ob = 0; ob < 2; ++ob)
            pre[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(W1op[(ob * 8 + s) * 64 + lane], axv[s >> 2][s & 3], pre[ob], 0, 0, 0);
        }
      f32x4 v[2];
      v[0][0] = dg_tanh(pre[0][0] + b0.x); v[0][1] = dg_tanh(pre[0][1] + b0.y);
      v[0][2] = dg_tanh(pre[0][2] + b0.z); v[0][3] = dg_tanh(pre[0][3] + b0.w);
      v[1][0] = dg_tanh(pre[1][0] + b1v.x); v[1][1] = dg_tanh(pre[1][1] + b1v.y);
      v[1][2] = dg_tanh(pre[1][2] + b1v.z); v[1][3] = dg_tanh(pre[1][3] + b1v.w);
      rows_and_linear(v, x1, W2op, H1);
    }
    CH_T(4);
    dg_lds_barrier();
    CH_T(5);
    // ---- conv2 ---------------------------------------------------------------------------------------------------------
    if (live) {
      f32x4 acc[2], v[2];
      product(H1, 2, acc);
      act32(1, acc, v);
      rows_and_linear(v, x2, W3op, H0);
    }
    CH_T(6);
    dg_lds_barrier();
    CH_T(7);
    // ---- conv3 (next linear step 32 -> 1) --------------------------------------------------------------------------------
    if (live) {
      const float4 w0 = bias4(3, 0), w1 = bias4(3, 1);
      f32x4 acc[2], v[2];
      product(H0, 2, acc);
      act32(2, acc, v);
      if (mrow < n) {
        float* dst = x3 + (size_t)(n0 + mrow) * 32 + 4 * kq;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
        *reinterpret_cast<float4*>(dst + 16) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
      }
      float p = v[0][0] * w0.x;
      p = fmaf(v[0][1], w0.y, p); p = fmaf(v[0][2], w0.z, p); p = fmaf(v[0][3], w0.w, p);
      p = fmaf(v[1][0], w1.x, p); p = fmaf(v[1][1], w1.y, p); p = fmaf(v[1][2], w1.z, p); p = fmaf(v[1][3], w1.w, p);
      p += __shfl_xor(p, 16);
      p += __shfl_xor(p, 32);
      if (kq == 0) {
        unsigned q0, q1, q2;
        ch_split3(dn * p, q0, q1, q2);
        h4p[mrow] = (unsigned short)q0; h4p[C::ROWS + mrow] = (unsigned short)q1; h4p[2 * C::ROWS + mrow] = (unsigned short)q2;
      }
    }
    CH_T(8);
    dg_lds_barrier();
    CH_T(9);
    // ---- conv4 -----------------------------------------------------------------------------------------------------------
    if (live) {
      f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
      const unsigned short* hq = h4p + min(nl, 2) * C::ROWS + 4 * kq;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < K32) {
          const unsigned w = wb[u];
          if (__builtin_amdgcn_ballot_w64(w != 0u) != 0ull) {
            const bf16x8 aop = ch_bits_operand(w, kq, tab);
            uint2 lo = *reinterpret_cast<const uint2*>(hq + 32 * u), hi = *reinterpret_cast<const uint2*>(hq + 32 * u + 16);
            if (nl >= 3) { lo = make_uint2(0u, 0u); hi = make_uint2(0u, 0u); }
            bf16x8 bop;
            unsigned* bu = reinterpret_cast<unsigned*>(&bop);
            bu[0] = lo.x; bu[1] = lo.y; bu[2] = hi.x; bu[3] = hi.y;
            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aop, bop, a4, 0, 0, 0);
          }
        }
      }
      float tot[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = __shfl_xor(a4[r], 1), s2 = __shfl_xor(a4[r], 2);
        tot[r] = (a4[r] + s1) + s2;
      }
      if (nl == 0) {
        const int mm = 16 * wave + 4 * kq;
        const float4 dq = *reinterpret_cast<const float4*>(dv + mm);
        const float dd[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (mm + r < n) x4[n0 + mm + r] = dg_tanh(fmaf(dd[r], tot[r], b4s));
      }
    }
    CH_T(10);
#ifdef CH_TIMING
    if (dbg && tid == 0) dbg[blockIdx.x * 16 + 11] += 1;
#endif
    n0 = n0N; n = nN; par ^= 1;
  }
}

// =================================================================================================================
// The same persistent kernel with FOUR waves per workgroup and TWO 16-row tiles per wave (graphs of <= 128 nodes), one
// LDS image of hs instead of two: 39 KB of LDS and <= 128 registers, i.e. FOUR workgroups = four graphs in flight per CU
// instead of two (the per-layer time of a graph is a latency chain, ~2-3 k cycles, that barely stretches when more graphs
// share the CU: profiles/r03 phase clocks).  A wave's two tiles share every HS^T operand it reads (half the LDS read
// traffic per matrix instruction) and their dependent chains interleave.  The image is overwritten in place, so a layer
// is  product -> barrier -> store next image -> barrier.
// =================================================================================================================
#ifndef CH_LOCKSTEP
#define CH_LOCKSTEP 0         // 1: the epilogues of a wave's two tiles run in lock step (more overlap, ~20 more registers: three
#endif                        // workgroups per CU instead of four)
template <int WAVES, int W1S>
struct ChQ {
  static constexpr int THREADS = 64 * WAVES, ROWS = 32 * WAVES, KW = WAVES, PS = ROWS * 32, BUF = 6 * PS;
  static constexpr int PB = ROWS * KW / THREADS;         // bitmap words of a graph per thread
  static constexpr int WJ = 1024 / THREADS;              // weight-matrix elements per thread
  static constexpr int OFF_W1 = BUF, OFF_W2 = OFF_W1 + W1S * 512, OFF_W3 = OFF_W2 + 4096, OFF_BT = OFF_W3 + 4096;
  static constexpr int OFF_DV = OFF_BT + 512;            // two sets (graph parity): dinv [ROWS] f32
  static constexpr int OFF_H4 = OFF_DV + 2 * 4 * ROWS;   // two sets: h4s parts [3][ROWS] bf16
  static constexpr int OFF_BL = OFF_H4 + 2 * 6 * ROWS;   // bitmap rows, <= KW words each
  static constexpr int OFF_TAB = OFF_BL + 4 * KW * ROWS;
  static constexpr int TOTAL = OFF_TAB + 128;
};

// WAVES = 4: graphs of <= 128 nodes, four workgroups per CU; WAVES = 8: <= 256 nodes (one LDS image of 48 KB), two per CU
// XI: xs items (row, 4-column slot) per thread; W1S: k-steps of conv1's weight table (4: F <= 16, 8: F <= 32)
template <int WAVES, int XI, int W1S>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4)))      // <= 128 registers
k_chain_fwd_q(int N, int B, int F, const int* __restrict__ sched, const int* __restrict__ nbig_p, const int* __restrict__ graph_ptr,
              const unsigned* __restrict__ bits,
              const float* __restrict__ dinv, const float* __restrict__ xs, ChW gw, float* __restrict__ axg,
              float* __restrict__ x1, float* __restrict__ x2, float* __restrict__ x3, float* __restrict__ x4,
              unsigned long long* __restrict__ dbg) {
  using C = ChQ<WAVES, W1S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  char* H = smem;
  float* W1op = reinterpret_cast<float*>(smem + C::OFF_W1);
  float* W2op = reinterpret_cast<float*>(smem + C::OFF_W2);
  float* W3op = reinterpret_cast<float*>(smem + C::OFF_W3);
  float* bt = reinterpret_cast<float*>(smem + C::OFF_BT);
  unsigned* bl = reinterpret_cast<unsigned*>(smem + C::OFF_BL);
  uint2* tab = reinterpret_cast<uint2*>(smem + C::OFF_TAB);
#ifdef CH_TIMING
  unsigned long long tprev_ = clock64();
  if (dbg && tid == 0) { for (int k = 0; k < 16; ++k) dbg[blockIdx.x * 16 + k] = 0; dbg[blockIdx.x * 16 + 12] = wall_clock64(); }
#endif
  // sched == null (batches of at most one graph per workgroup: no schedule was built): entry r = graph r*G + w of graph_ptr
  const int nbig = sched ? nbig_p[0] : 0, ns = B - nbig, G = (int)gridDim.x, w = (int)blockIdx.x;
  auto entry_of = [&](int r) {               // {n0, n}; n = 0 past the end (load clamped, selected)
    const int li = r * G + ((r & 1) ? G - 1 - w : w), lc = min(li, max(ns - 1, 0));
    int2 e;
    if (sched) e = *reinterpret_cast<const int2*>(sched + 2 * (nbig + lc));
    else { e.x = graph_ptr[lc]; e.y = graph_ptr[lc + 1] - e.x; }
    return make_int2(e.x, (li < ns && e.y <= C::ROWS) ? e.y : 0);
  };
  int2 eC = entry_of(0), eN = entry_of(1);
  const int lg = F <= 4 ? 0 : (F <= 8 ? 1 : (F <= 16 ? 2 : 3));      // 4-column slots with data per row: 2^lg
  const int NBF = F > 16 ? 2 : 1;
  unsigned pbit[C::PB]; float pdv = 0.f; float pxs[XI][4];
  auto prefetch = [&](int pn0, int pn) {      // (unconditional loads on clamped addresses, selected when consumed)
    const int pS = 1 << dgd_class(max(pn, 1));
    const unsigned* bp = bits + (size_t)N * (pS - 1) + (size_t)pn0 * pS;
    const int last = max(pn * pS - 1, 0);
#pragma unroll
    for (int j = 0; j < C::PB; ++j) pbit[j] = bp[min(tid + C::THREADS * j, last)];
    pdv = dinv[pn0 + min(tid, max(pn - 1, 0))];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int it = tid + C::THREADS * j, k = min(it >> lg, max(pn - 1, 0)), qq = it & ((1 << lg) - 1);
      const float* xr = xs + (size_t)(pn0 + k) * F;
#pragma unroll
      for (int i = 0; i < 4; ++i) pxs[j][i] = xr[min(4 * qq + i, F - 1)];
    }
  };
  int n0 = __builtin_amdgcn_readfirstlane(eC.x), n = __builtin_amdgcn_readfirstlane(eC.y);
  prefetch(n0, n);                            // the first graph's data travels while the weight tables are set up
  // ---- once per workgroup: weight tables in MFMA-operand order (coalesced loads, scattered on the LDS side) ------------
  {
    float w2[C::WJ], w3[C::WJ], w1[C::WJ];
#pragma unroll
    for (int j = 0; j < C::WJ; ++j) {
      const int e = tid + C::THREADS * j;
      w2[j] = gw.W2[e]; w3[j] = gw.W3[e]; w1[j] = e < 32 * F ? gw.W1[e] : 0.f;
    }
    float bv = 0.f;
    if (tid < 128) {
      const int which = tid >> 5, idx = tid & 31;
      const float* src = which == 0 ? gw.b1 : (which == 1 ? gw.b2 : (which == 2 ? gw.b3 : gw.W4));
      bv = src[idx];
    }
    // W[o][k] -> [ob = o >> 4][s = 4 (k >> 4) + (k & 3)][lane = (o & 15) + 16 ((k >> 2) & 3)], S steps per ob
    auto slot_of = [](int o, int k, int S) {
      return (((o >> 4) * S + ((k >> 4) << 2) + (k & 3)) << 6) + (o & 15) + (((k >> 2) & 3) << 4);
    };
    for (int e = tid; e < 2 * W1S * 64; e += C::THREADS) {      // conv1's table: entries with k >= F are zero
      const int s_ = (e >> 6) % W1S, l = e & 63;
      if (16 * (s_ >> 2) + 4 * (l >> 4) + (s_ & 3) >= F) W1op[e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < C::WJ; ++j) {
      const int e = tid + C::THREADS * j;
      W2op[slot_of(e >> 5, e & 31, 8)] = w2[j]; W3op[slot_of(e >> 5, e & 31, 8)] = w3[j];
      if (e < 32 * F) { const int o = e / F; W1op[slot_of(o, e - o * F, W1S)] = w1[j]; }
    }
    if (tid < 128) bt[tid] = bv;
    if (tid < 16) tab[tid] = make_uint2(((tid & 1) ? 0x3f80u : 0u) | ((tid & 2) ? 0x3f800000u : 0u),
                                        ((tid & 4) ? 0x3f80u : 0u) | ((tid & 8) ? 0x3f800000u : 0u));
  }
  const float b4s = gw.b4[0];
  __syncthreads();
  const int rdoff = (4 * kq + (nl >> 2)) * 32 + 8 * ((nl & 3) ^ kq);
  const int mrow0 = 16 * wave + nl, mrow1 = mrow0 + 16 * WAVES;         // this lane's node in tile wave / tile wave + WAVES
  const int wsl = 8 * (kq ^ ((nl >> 2) & 3));
  int par = 0;
  CH_T(0);                                                // 0: set-up
  for (int r = 0; r * G < ns; ++r) {
    float* dv = reinterpret_cast<float*>(smem + C::OFF_DV) + par * C::ROWS;
    unsigned short* h4p = reinterpret_cast<unsigned short*>(smem + C::OFF_H4) + par * 3 * C::ROWS;
    const int K32 = (n + 31) >> 5, T = (n + 15) >> 4, RU = 32 * K32;
    const int S = 1 << dgd_class(max(n, 1));
    // graphs of <= ROWS/2 nodes keep TWO images, rows [0, ROWS/2) and [ROWS/2, ROWS) of every plane, and alternate between
    // them: a layer's output image is not the one still being read, so the barrier in front of its stores is not needed
    const int pong = (2 * n <= C::ROWS) ? (C::ROWS / 2) * 32 : 0;
    // ---- stage the graph: registers -> LDS images -----------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < C::PB; ++j)
      if (tid + C::THREADS * j < n * S) bl[tid + C::THREADS * j] = pbit[j];
    if (tid < C::ROWS) dv[tid] = tid < n ? pdv : 0.f;
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int it = tid + C::THREADS * j, k = it >> lg, qq = it & ((1 << lg) - 1);
      if (k < n) {
        unsigned sp[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ch_split3(4 * qq + i < F ? pxs[j][i] : 0.f, sp[0][i], sp[1][i], sp[2][i]);
        const int nb = qq >> 2, sl = qq & 3;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint2*>(H + (p * 2 + nb) * C::PS + k * 32 + 8 * (sl ^ ((k >> 2) & 3))) =
              make_uint2(sp[p][0] | (sp[p][1] << 16), sp[p][2] | (sp[p][3] << 16));
      }
    }
    {   // zeros conv1 reads but nobody wrote: rows n..RU-1 of its planes, and the slots beyond the feature width
      const int sh = NBF + 1;                              // 4 * NBF slots per row
      for (int it = tid; it < (RU << sh); it += C::THREADS) {
        const int k = it >> sh, qq = it & ((1 << sh) - 1);
        if (!(k < n && qq < (1 << lg))) {
          const int nb = qq >> 2, sl = qq & 3;
#pragma unroll
          for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint2*>(H + (p * 2 + nb) * C::PS + k * 32 + 8 * (sl ^ ((k >> 2) & 3))) = make_uint2(0u, 0u);
        }
      }
    }
    if (RU > 16 * T) {   // rows 16T .. RU-1 (sixteen) of the planes conv1 does not use, and of h4s: written by no tile
      if (tid < 192) {
        const int piece = tid & 1, row = 16 * T + ((tid >> 1) & 15), pp = tid >> 5;
        if ((pp & 1) >= NBF) *reinterpret_cast<uint4*>(H + pp * C::PS + row * 32 + 16 * piece) = make_uint4(0u, 0u, 0u, 0u);
      }
      if (tid < 48) h4p[(tid >> 4) * C::ROWS + 16 * T + (tid & 15)] = 0;
      if (pong && tid < 192)
        *reinterpret_cast<uint4*>(H + pong + (tid >> 5) * C::PS + (16 * T + ((tid >> 1) & 15)) * 32 + 16 * (tid & 1)) = make_uint4(0u, 0u, 0u, 0u);
    }
    CH_T(1);                                              // 1: wait for the prefetched data + staging stores
    dg_lds_barrier();
    CH_T(2);
    const int n0N = __builtin_amdgcn_readfirstlane(eN.x), nN = __builtin_amdgcn_readfirstlane(eN.y);     // (requested a graph ago)
    eN = entry_of(r + 2);
    prefetch(n0N, nN);
    // this lane's bitmap rows stay in LDS (re-read per layer: 2 x K32 words; in registers they cost 16 at eight waves)
    const unsigned* bl0 = bl + min(mrow0, max(n - 1, 0)) * S;
    const unsigned* bl1 = bl + min(mrow1, max(n - 1, 0)) * S;
    const bool rv0 = mrow0 < n, rv1 = mrow1 < n;
    const float dn[2] = {dv[mrow0], dv[mrow1]};
    const int mrow[2] = {mrow0, mrow1};
    CH_T(3);
    const bool live0 = wave < T, live1 = wave + WAVES < T;
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

    // block products of this wave's tiles with the image: every HS^T operand read once, used by both tiles; the reads of
    // word u+1 are issued before the matrix instructions of word u
    auto product = [&](auto nbc, const char* Hc, f32x4 (&acc)[2][2]) {
      constexpr int NBP = decltype(nbc)::value;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) { acc[ti][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[ti][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      const char* hp = Hc + rdoff;
#pragma unroll
      for (int u = 0; u < C::KW; ++u) {
        if (u < K32) {
          const bf16x8 bop0 = ch_bits_operand(rv0 ? bl0[u] : 0u, kq, tab);
          const bf16x8 bop1 = ch_bits_operand(rv1 ? bl1[u] : 0u, kq, tab);
          bf16x8 a[3][NBP];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int nb = 0; nb < NBP; ++nb) a[p][nb] = ch_read_hsT(hp + (p * 2 + nb) * C::PS + u * 1024);
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int nb = 0; nb < NBP; ++nb) acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][nb], bop0, acc[0][nb], 0, 0, 0);
          if (live1) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int nb = 0; nb < NBP; ++nb) acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][nb], bop1, acc[1][nb], 0, 0, 0);
          }
        }
      }
    };
    auto bias4 = [&](int which, int nb) { return *reinterpret_cast<const float4*>(bt + 32 * which + 16 * nb + 4 * kq); };
    // NT tiles in lock step: activated rows to global, next layer's pre-scaled linear output hs (registers, same layout)
    auto rows_and_linear = [&](auto ntc, auto t0c, const f32x4 (&v)[2][2], float* __restrict__ xout, const float* __restrict__ Wop,
                               f32x4 (&hs)[2][2]) {
      constexpr int T0 = decltype(t0c)::value, NT = T0 + decltype(ntc)::value;       // tiles T0 .. NT-1
#pragma unroll
      for (int ti = T0; ti < NT; ++ti)
        if (mrow[ti] < n) {
          float* dst = xout + (size_t)(n0 + mrow[ti]) * 32 + 4 * kq;
          *reinterpret_cast<float4*>(dst) = make_float4(v[ti][0][0], v[ti][0][1], v[ti][0][2], v[ti][0][3]);
          *reinterpret_cast<float4*>(dst + 16) = make_float4(v[ti][1][0], v[ti][1][1], v[ti][1][2], v[ti][1][3]);
        }
      float wv[2][8];
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int s = 0; s < 8; ++s) wv[ob][s] = Wop[(ob * 8 + s) * 64 + lane];
      // four independent accumulator chains per wave: (tile x output block) with two tiles, (output block x input half)
      // with one -- the dependent-accumulator latency of the fp32 matrix instruction is 40 cycles, its issue interval 32
      constexpr int NH = (NT - T0) == 2 ? 1 : 2;
      f32x4 d2[2][2][NH];
#pragma unroll
      for (int ti = T0; ti < NT; ++ti)
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
          for (int h = 0; h < NH; ++h) d2[ti][ob][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int ti = T0; ti < NT; ++ti)
              d2[ti][ob][h % NH] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ob][4 * h + s], v[ti][h][s], d2[ti][ob][h % NH], 0, 0, 0);
#pragma unroll
      for (int ti = T0; ti < NT; ++ti)
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) hs[ti][ob][rr] = dn[ti] * (NH == 2 ? d2[ti][ob][0][rr] + d2[ti][ob][NH - 1][rr] : d2[ti][ob][0][rr]);
    };
    auto write_hs = [&](char* Hn, int ti, const f32x4 (&hs)[2]) {
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        unsigned sp[3][4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) ch_split3(hs[ob][rr], sp[0][rr], sp[1][rr], sp[2][rr]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          *reinterpret_cast<uint2*>(Hn + (p * 2 + ob) * C::PS + mrow[ti] * 32 + wsl) =
              make_uint2(sp[p][0] | (sp[p][1] << 16), sp[p][2] | (sp[p][3] << 16));
      }
    };
    auto act32 = [&](auto ntc, auto t0c, int which, const f32x4 (&acc)[2][2], f32x4 (&v)[2][2]) {
      constexpr int T0 = decltype(t0c)::value, NT = T0 + decltype(ntc)::value;
      const float4 b0 = bias4(which, 0), b1v = bias4(which, 1);
#pragma unroll
      for (int ti = T0; ti < NT; ++ti) {
        v[ti][0][0] = dg_tanh(fmaf(dn[ti], acc[ti][0][0], b0.x)); v[ti][0][1] = dg_tanh(fmaf(dn[ti], acc[ti][0][1], b0.y));
        v[ti][0][2] = dg_tanh(fmaf(dn[ti], acc[ti][0][2], b0.z)); v[ti][0][3] = dg_tanh(fmaf(dn[ti], acc[ti][0][3], b0.w));
        v[ti][1][0] = dg_tanh(fmaf(dn[ti], acc[ti][1][0], b1v.x)); v[ti][1][1] = dg_tanh(fmaf(dn[ti], acc[ti][1][1], b1v.y));
        v[ti][1][2] = dg_tanh(fmaf(dn[ti], acc[ti][1][2], b1v.z)); v[ti][1][3] = dg_tanh(fmaf(dn[ti], acc[ti][1][3], b1v.w));
      }
    };
    f32x4 hsv[2][2];
    // ---- conv1 (aggregate-first): ax = dn (Adj xs) saved, x1 = tanh(ax W1^T + b1), hs2 = dn (x1 W2^T) ----------------------
    if (live0) {
      f32x4 acc[2][2];
      if (NBF == 2) product(I2{}, H, acc); else product(I1{}, H, acc);
      auto conv1_tail = [&](auto ntc, auto t0c) {
        constexpr int T0 = decltype(t0c)::value, NT = T0 + decltype(ntc)::value;
        const float4 b0 = bias4(0, 0), b1v = bias4(0, 1);
        f32x4 axv[2][2], pre[2][2], v[2][2];
#pragma unroll
        for (int ti = T0; ti < NT; ++ti)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            pre[ti][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              axv[ti][nb][rr] = dn[ti] * acc[ti][nb][rr];
              const int c = 16 * nb + 4 * kq + rr;
              if (c < F && mrow[ti] < n) axg[(size_t)(n0 + mrow[ti]) * F + c] = axv[ti][nb][rr];
            }
          }
#pragma unroll
        for (int s = 0; s < W1S; ++s)
          if (16 * (s >> 2) + (s & 3) < F) {
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
              const float wv = W1op[(ob * W1S + s) * 64 + lane];
#pragma unroll
              for (int ti = T0; ti < NT; ++ti)
                pre[ti][ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, axv[ti][s >> 2][s & 3], pre[ti][ob], 0, 0, 0);
            }
          }
#pragma unroll
        for (int ti = T0; ti < NT; ++ti) {
          v[ti][0][0] = dg_tanh(pre[ti][0][0] + b0.x); v[ti][0][1] = dg_tanh(pre[ti][0][1] + b0.y);
          v[ti][0][2] = dg_tanh(pre[ti][0][2] + b0.z); v[ti][0][3] = dg_tanh(pre[ti][0][3] + b0.w);
          v[ti][1][0] = dg_tanh(pre[ti][1][0] + b1v.x); v[ti][1][1] = dg_tanh(pre[ti][1][1] + b1v.y);
          v[ti][1][2] = dg_tanh(pre[ti][1][2] + b1v.z); v[ti][1][3] = dg_tanh(pre[ti][1][3] + b1v.w);
        }
        rows_and_linear(ntc, t0c, v, x1, W2op, hsv);
      };
      using I0 = std::integral_constant<int, 0>;
      if (CH_LOCKSTEP) { if (live1) conv1_tail(I2{}, I0{}); else conv1_tail(I1{}, I0{}); }
      else { conv1_tail(I1{}, I0{}); if (live1) conv1_tail(I1{}, I1{}); }
    }
    CH_T(4);
    if (!pong) dg_lds_barrier();                          // every wave has read the image: it can be overwritten
    if (live0) write_hs(H + pong, 0, hsv[0]);
    if (live1) write_hs(H + pong, 1, hsv[1]);
    dg_lds_barrier();
    CH_T(5);
    // ---- conv2 -----------------------------------------------------------------------------------------------------------
    if (live0) {
      f32x4 acc[2][2], v[2][2];
      product(I2{}, H + pong, acc);
      using I0 = std::integral_constant<int, 0>;
      if (CH_LOCKSTEP && live1) { act32(I2{}, I0{}, 1, acc, v); rows_and_linear(I2{}, I0{}, v, x2, W3op, hsv); }
      else {
        act32(I1{}, I0{}, 1, acc, v); rows_and_linear(I1{}, I0{}, v, x2, W3op, hsv);
        if (live1) { act32(I1{}, I1{}, 1, acc, v); rows_and_linear(I1{}, I1{}, v, x2, W3op, hsv); }
      }
    }
    CH_T(6);
    if (!pong) dg_lds_barrier();
    if (live0) write_hs(H, 0, hsv[0]);
    if (live1) write_hs(H, 1, hsv[1]);
    dg_lds_barrier();
    CH_T(7);
    // ---- conv3 (next linear step 32 -> 1: h4s = dn (x3 . w4), three bf16 parts in their own buffer) -------------------------
    if (live0) {
      const float4 w0 = bias4(3, 0), w1 = bias4(3, 1);
      f32x4 acc[2][2], v[2][2];
      product(I2{}, H, acc);
      auto conv3_tail = [&](auto ntc) {
        constexpr int NT = decltype(ntc)::value;
        act32(ntc, std::integral_constant<int, 0>{}, 2, acc, v);
        float pp[2];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          if (mrow[ti] < n) {
            float* dst = x3 + (size_t)(n0 + mrow[ti]) * 32 + 4 * kq;
            *reinterpret_cast<float4*>(dst) = make_float4(v[ti][0][0], v[ti][0][1], v[ti][0][2], v[ti][0][3]);
            *reinterpret_cast<float4*>(dst + 16) = make_float4(v[ti][1][0], v[ti][1][1], v[ti][1][2], v[ti][1][3]);
          }
          float p = v[ti][0][0] * w0.x;
          p = fmaf(v[ti][0][1], w0.y, p); p = fmaf(v[ti][0][2], w0.z, p); p = fmaf(v[ti][0][3], w0.w, p);
          p = fmaf(v[ti][1][0], w1.x, p); p = fmaf(v[ti][1][1], w1.y, p); p = fmaf(v[ti][1][2], w1.z, p); p = fmaf(v[ti][1][3], w1.w, p);
          pp[ti] = p;
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) pp[ti] += __shfl_xor(pp[ti], 16);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) pp[ti] += __shfl_xor(pp[ti], 32);
        if (kq == 0) {
#pragma unroll
          for (int ti = 0; ti < NT; ++ti) {
            unsigned q0, q1, q2;
            ch_split3(dn[ti] * pp[ti], q0, q1, q2);
            h4p[mrow[ti]] = (unsigned short)q0; h4p[C::ROWS + mrow[ti]] = (unsigned short)q1; h4p[2 * C::ROWS + mrow[ti]] = (unsigned short)q2;
          }
        }
      };
      if (live1) conv3_tail(I2{}); else conv3_tail(I1{});
    }
    CH_T(8);
    dg_lds_barrier();
    CH_T(9);
    // ---- conv4: the three parts of h4s are columns 0..2 of ONE B operand ---------------------------------------------------
    if (live0) {
      f32x4 a4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      const unsigned short* hq = h4p + min(nl, 2) * C::ROWS + 4 * kq;
#pragma unroll
      for (int u = 0; u < C::KW; ++u) {
        if (u < K32) {
          uint2 lo = *reinterpret_cast<const uint2*>(hq + 32 * u), hi = *reinterpret_cast<const uint2*>(hq + 32 * u + 16);
          if (nl >= 3) { lo = make_uint2(0u, 0u); hi = make_uint2(0u, 0u); }
          bf16x8 bop;
          unsigned* bu = reinterpret_cast<unsigned*>(&bop);
          bu[0] = lo.x; bu[1] = lo.y; bu[2] = hi.x; bu[3] = hi.y;
          a4[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch_bits_operand(rv0 ? bl0[u] : 0u, kq, tab), bop, a4[0], 0, 0, 0);
          if (live1) a4[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch_bits_operand(rv1 ? bl1[u] : 0u, kq, tab), bop, a4[1], 0, 0, 0);
        }
      }
      float s1[2][4], s2[2][4];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { s1[ti][rr] = __shfl_xor(a4[ti][rr], 1); s2[ti][rr] = __shfl_xor(a4[ti][rr], 2); }
      if (nl == 0) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          const int mm = 16 * (wave + WAVES * ti) + 4 * kq;
          const float4 dq = *reinterpret_cast<const float4*>(dv + mm);
          const float dd[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            if (mm + rr < n) x4[n0 + mm + rr] = dg_tanh(fmaf(dd[rr], (a4[ti][rr] + s1[ti][rr]) + s2[ti][rr], b4s));
        }
      }
    }
    CH_T(10);
#ifdef CH_TIMING
    if (dbg && tid == 0) dbg[blockIdx.x * 16 + 11] += 1;
#endif
    n0 = n0N; n = nN; par ^= 1;
  }
#ifdef CH_TIMING
  if (dbg && tid == 0) dbg[blockIdx.x * 16 + 13] = wall_clock64();
#endif
}

// ---- host launcher ----------------------------------------------------------------------------------------------------
// size classes: graphs of <= 128 nodes (8 waves, one 16-row tile each, hs ping-pongs between two LDS images: 62 KB, two
// workgroups per CU) and 129..512 nodes (16 waves x two tiles, one LDS image: 111 KB).  Each launch walks all B graphs
// and leaves the other class' graphs alone; the second launch is skipped when the host's max_nodes hint rules it out.
#ifndef CH_GRID
#define CH_GRID 512                      // eight-wave form: two persistent workgroups per CU (LDS 70 KB each)
#endif
#ifndef CH_QW
#define CH_QW 8                          // waves per workgroup of the two-tile form: 8 -> graphs of <= 256 nodes, two workgroups per CU;
#endif                                   // 4 -> <= 128 nodes, four per CU (measured: ...)
#ifndef CH_GRID_Q
#define CH_GRID_Q (CH_QW == 8 ? 512 : 1024)
#endif
#define CH_SMALL_ROWS (32 * CH_QW)
int dg_chain_max_nodes() { return 512; }
int dg_chain_small_rows() { return CH_SMALL_ROWS; }
// batches of at most one graph per persistent workgroup need no schedule (and graph preparation no planning pass)
int dg_chain_needs_schedule(int B) { return B > CH_GRID_Q ? 1 : 0; }

int dg_launch_chain_fwd(int N, int B, int F, int max_nodes, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv,
                        const float* xs, const float* params, const DgParams* pl, float* ax, float* x1, float* x2, float* x3,
                        float* x4, int32_t* dmap, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  // dmap == null: no schedule was built (dg_chain_needs_schedule(B) == 0): one graph per workgroup straight from graph_ptr
  if (N <= 0 || B <= 0 || F < 1 || F > DG_AF_MAX_F || !graph_ptr || !bits || !dinv || !xs) return DGCNN_EINVAL;
  if (!dmap && dg_chain_needs_schedule(B)) return DGCNN_EINVAL;
  ChW gw;
  gw.W1 = params + pl->off[0]; gw.b1 = params + pl->off[1]; gw.W2 = params + pl->off[2]; gw.b2 = params + pl->off[3];
  gw.W3 = params + pl->off[4]; gw.b3 = params + pl->off[5]; gw.W4 = params + pl->off[6]; gw.b4 = params + pl->off[7];
  using CL = ChCfg<16, 2, false>;
  static bool attr_set = false;
  if (!attr_set) {
#define CH_ATTR(W, XI, WS) (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fwd_q<W, XI, WS>), \
                            hipFuncAttributeMaxDynamicSharedMemorySize, ChQ<W, WS>::TOTAL) != hipSuccess)
    if (CH_ATTR(CH_QW, 1, 4) || CH_ATTR(CH_QW, 2, 4) || CH_ATTR(CH_QW, 4, 8) ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fwd_p<1>), hipFuncAttributeMaxDynamicSharedMemorySize, ChP::TOTAL) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fwd_p<2>), hipFuncAttributeMaxDynamicSharedMemorySize, ChP::TOTAL) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fwd<16, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            CL::TOTAL) != hipSuccess)
      return DGCNN_ELAUNCH;
    attr_set = true;
  }
  const int* sched = dmap ? dmap + dgd_sched0(N, B) : nullptr;
  const int* nbig = dmap ? dmap + DGD_NBIG + (CH_SMALL_ROWS == 256 ? 1 : 0) : nullptr;      // graphs above the size class = first entry of the class
#ifdef CH_USE_P8        // measurement builds: the eight-wave, one-tile-per-wave form (two workgroups per CU)
  const int grid = B < CH_GRID ? B : CH_GRID;
  if (F <= 16)
    hipExtLaunchKernelGGL((k_chain_fwd_p<1>), dim3(grid), dim3(512), ChP::TOTAL, s, ev_start, ev_stop, 0, N, B, F, sched, nbig,
                          bits, dinv, xs, gw, ax, x1, x2, x3, x4, dg_debug_buffer());
  else
    hipExtLaunchKernelGGL((k_chain_fwd_p<2>), dim3(grid), dim3(512), ChP::TOTAL, s, ev_start, ev_stop, 0, N, B, F, sched, nbig,
                          bits, dinv, xs, gw, ax, x1, x2, x3, x4, dg_debug_buffer());
#else
  const int grid = B < CH_GRID_Q ? B : CH_GRID_Q;
  // XI = xs items per thread: rows x 4-column slots with data / threads = 2^lg / 2 (at least 1)
#define CH_LQ(XI, WS) hipExtLaunchKernelGGL((k_chain_fwd_q<CH_QW, XI, WS>), dim3(grid), dim3(64 * CH_QW), ChQ<CH_QW, WS>::TOTAL, s, ev_start, \
                                            ev_stop, 0, N, B, F, sched, nbig, graph_ptr, bits, dinv, xs, gw, ax, x1, x2, x3, x4, dg_debug_buffer())
  if (F <= 8) CH_LQ(1, 4); else if (F <= 16) CH_LQ(2, 4); else CH_LQ(4, 8);
#undef CH_LQ
#endif
  DG_CHECK_LAUNCH();
  if (max_nodes <= 0 || max_nodes > CH_SMALL_ROWS) {
    hipLaunchKernelGGL((k_chain_fwd<16, 2, false>), dim3(B), dim3(CL::THREADS), CL::TOTAL, s, N, F, graph_ptr, bits, dinv, xs, gw,
                       ax, x1, x2, x3, x4, CH_SMALL_ROWS, sched, nbig);
    DG_CHECK_LAUNCH();
  }
  return DGCNN_OK;
}
