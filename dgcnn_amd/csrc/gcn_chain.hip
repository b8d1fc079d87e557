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
#include "dg_readout.h"
#include "dg_tail_body.h"
#include <hip/hip_ext.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#ifdef DG_EMU
#define CH_LDS
#else
#define CH_LDS __attribute__((address_space(3)))
#endif

struct ChW { const float *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4; };

__device__ __forceinline__ void ch_split3(float h, unsigned& p0, unsigned& p1, unsigned& p2) {      // h = p0 + p1 + p2, exact
  const unsigned u0 = __float_as_uint(h) & 0xffff0000u;
  const float r1 = h - __uint_as_float(u0);
  const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(u1);
  p0 = u0 >> 16; p1 = u1 >> 16; p2 = __float_as_uint(r2) >> 16;
}

// four values at once, straight to the three packed image words: part p of (a, b, c, d) as {lo = bf16(a) | bf16(b) << 16, hi = c | d}.
// v_perm_b32 takes the HIGH halves of two registers in one instruction, so the parts are never shifted down and or-ed
// together: 4 x (and, sub, and, sub) + 6 perms = 22 instructions per four values instead of 34.
__device__ __forceinline__ void ch_split3_pack4(float a, float b, float c, float d, uint2 (&out)[3]) {
  const float v[4] = {a, b, c, d};
  unsigned q0[4], q1[4], q2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    q0[i] = __float_as_uint(v[i]);
    const float r1 = v[i] - __uint_as_float(q0[i] & 0xffff0000u);
    q1[i] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(q1[i] & 0xffff0000u);
    q2[i] = __float_as_uint(r2);
  }
  // perm(s0, s1, sel): result = {s0.b3, s0.b2, s1.b3, s1.b2} = hi16(s0) << 16 | hi16(s1)
  out[0] = make_uint2(__builtin_amdgcn_perm(q0[1], q0[0], 0x07060302u), __builtin_amdgcn_perm(q0[3], q0[2], 0x07060302u));
  out[1] = make_uint2(__builtin_amdgcn_perm(q1[1], q1[0], 0x07060302u), __builtin_amdgcn_perm(q1[3], q1[2], 0x07060302u));
  out[2] = make_uint2(__builtin_amdgcn_perm(q2[1], q2[0], 0x07060302u), __builtin_amdgcn_perm(q2[3], q2[2], 0x07060302u));
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

__device__ __forceinline__ float ch_row16_sum(float v);

template <int WAVES, int TPW, bool PING>
__global__ void __launch_bounds__(64 * WAVES)
k_chain_fwd(int N, int F, const int* __restrict__ graph_ptr, const unsigned* __restrict__ bits, const float* __restrict__ dinv,
            const float* __restrict__ xs, ChW gw, float* __restrict__ axg, float* __restrict__ x1, float* __restrict__ x2,
            float* __restrict__ x3, float* __restrict__ x4, int nmin, const int* __restrict__ sched, const int* __restrict__ nbig_p) {
  using C = ChCfg<WAVES, TPW, PING>;
  DG_DYN_SMEM(char, smem);
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
// PERSISTENT form: workgroups stay resident, stage the
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
#elif defined(CH_FINE)      // absolute stamps of workgroup 0 in slots 48.. (tools/phase_step_kernel.py; the one-launch training kernel only)
#define CH_T(k) do { if (dbg && blockIdx.x == 0 && tid == 0) dbg[48 + (k)] = clock64(); } while (0)
#else
#define CH_T(k) do { } while (0)
#endif
// =================================================================================================================
// WAVES waves per workgroup, up to TWO 16-row tiles per wave (tiles `wave` and `wave + WAVES`): graphs of up to 32*WAVES
// nodes.  A wave's two tiles share every HS^T operand it reads.  Graphs of at most half that size keep two images (the
// two halves of every plane) and alternate; larger ones overwrite the one image in place:
// product -> barrier -> store next image -> barrier.  Instantiations: 8 waves (<= 256 nodes, two workgroups per CU:
// the persistent form for large batches), 16 waves (<= 512 nodes, one workgroup per CU: small batches, one graph per
// workgroup, where the largest graph's critical path is the kernel's duration -- one tile per wave up to 256 nodes),
// 4 waves (<= 128 nodes, four per CU: measured no faster than 8, kept as a measurement build).
// =================================================================================================================
#ifndef CH_LOCKSTEP
#define CH_LOCKSTEP 0         // 1: the epilogues of a wave's two tiles run in lock step (more overlap, ~20 more registers: three
#endif                        // workgroups per CU instead of four)
template <int WAVES, int W1S, int MAXN = 32 * WAVES>     // MAXN: largest graph admitted (sizes the bitmap image; <= 32 * WAVES)
struct ChQ {
  static constexpr int THREADS = 64 * WAVES, ROWS = 32 * WAVES, KW = MAXN / 32, PS = ROWS * 32, BUF = 6 * PS;
  static constexpr int PB = (MAXN * KW + THREADS - 1) / THREADS;         // bitmap words of a graph per thread
  static constexpr int WJ = 1024 / THREADS;              // weight-matrix elements per thread
  static constexpr int OFF_W1 = BUF, OFF_W2 = OFF_W1 + W1S * 512, OFF_W3 = OFF_W2 + 4096, OFF_BT = OFF_W3 + 4096;
  static constexpr int OFF_DV = OFF_BT + 512;            // two sets (graph parity): dinv [ROWS] f32
  static constexpr int OFF_H4 = OFF_DV + 2 * 4 * ROWS;   // two sets: h4s parts [3][ROWS] bf16
  static constexpr int OFF_BL = OFF_H4 + 2 * 6 * ROWS;   // bitmap rows, <= KW words each
  static constexpr int OFF_TAB = OFF_BL + 4 * KW * MAXN;
  static constexpr int TOTAL = OFF_TAB + 128;
};

// WAVES = 4: graphs of <= 128 nodes, four workgroups per CU; WAVES = 8: <= 256 nodes (one LDS image of 48 KB), two per CU
// XI: xs items (row, 4-column slot) per thread; W1S: k-steps of conv1's weight table (4: F <= 16, 8: F <= 32)
// BF (the bf16 leg, BASELINE config 3): the pre-scaled linear outputs hs_2, hs_3 are kept as ONE bf16 part (round to nearest
// even: what "hs stored in bf16" means when hs never leaves the CU) and X.W^T runs on v_mfma_f32_16x16x32_bf16 with both
// operands rounded to bf16 -- one matrix instruction per 16 output columns instead of eight fp32 ones; conv1's raw
// features, its own linear step, the 32 -> 1 step and every sum stay fp32.
__device__ __forceinline__ unsigned ch_f2bf(float f) {      // round to nearest even (finite inputs)
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
template <int WAVES, int XI, int W1S, bool LOOP, int MAXN = 32 * WAVES, bool BF = false>      // LOOP = false: exactly one graph per workgroup (grid = B), nothing is prefetched
__device__ __forceinline__ void
ch_chain_body(int N, int B, int F, const int* __restrict__ sched, const int* __restrict__ nbig_p, const int* __restrict__ graph_ptr,
              const unsigned* __restrict__ bits,
              const float* __restrict__ dinv, const float* __restrict__ xs, const ChW& gw, float* __restrict__ axg,
              float* __restrict__ x1, float* __restrict__ x2, float* __restrict__ x3, float* __restrict__ x4,
              unsigned long long* __restrict__ dbg, float* x4_lds = nullptr) {
  // x4_lds (the one-launch training kernel): conv4's outputs are ALSO left in LDS, indexed by local node -- they are the
  // SortPooling keys the same workgroup reads next, and from LDS it need not wait for its own global stores to land first
  using C = ChQ<WAVES, W1S, MAXN>;
  DG_DYN_SMEM(char, smem);
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
    return make_int2(e.x, (li < ns && e.y <= MAXN) ? e.y : 0);
  };
  int2 eC = entry_of(0), eN = LOOP ? entry_of(1) : make_int2(0, 0);
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
  // The loads are issued here, BEHIND the first graph's prefetch; one graph per workgroup (LOOP = false): the table stores wait
  // until the graph is staged -- the weights are cold every step (the optimizer has just rewritten them on other XCDs), the
  // graph's data is not, loads return in order, and every load in between is unconditional so that the compiler can count:
  // the staging's ~2 k cycles of VALU work run under the weights' latency instead of behind it (and one barrier goes).
  float w2r[C::WJ], w3r[C::WJ], w1r[C::WJ];
#pragma unroll
  for (int j = 0; j < C::WJ; ++j) {
    const int e = tid + C::THREADS * j;
    w2r[j] = gw.W2[e]; w3r[j] = gw.W3[e]; w1r[j] = gw.W1[min(e, 32 * F - 1)];
  }
  const float* bsrc_ = (tid >> 5) == 0 ? gw.b1 : ((tid >> 5) == 1 ? gw.b2 : ((tid >> 5) == 2 ? gw.b3 : gw.W4));
  const float bvr = bsrc_[tid & 31];                      // (unconditional; lanes >= 128 re-read W4)
  const float b4v = gw.b4[0];
  auto store_tables = [&]() {
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
      if (BF) {     // bf16 A operands of X.W^T: [ob][lane][8]: W[16 ob + (lane & 15)][k], k = 4kq + j (j < 4), 16 + 4kq + j - 4
        const int o = e >> 5, k = e & 31;
        const int d = ((((o >> 4) << 6) + (o & 15) + (((k >> 2) & 3) << 4)) << 3) + (k & 3) + ((k >> 4) << 2);
        reinterpret_cast<unsigned short*>(W2op)[d] = (unsigned short)ch_f2bf(w2r[j]);
        reinterpret_cast<unsigned short*>(W3op)[d] = (unsigned short)ch_f2bf(w3r[j]);
      } else {
      W2op[slot_of(e >> 5, e & 31, 8)] = w2r[j]; W3op[slot_of(e >> 5, e & 31, 8)] = w3r[j];
      }
      if (e < 32 * F) { const int o = e / F; W1op[slot_of(o, e - o * F, W1S)] = w1r[j]; }
    }
    if (tid < 128) bt[tid] = bvr;
    if (tid < 16) tab[tid] = make_uint2(((tid & 1) ? 0x3f80u : 0u) | ((tid & 2) ? 0x3f800000u : 0u),
                                        ((tid & 4) ? 0x3f80u : 0u) | ((tid & 8) ? 0x3f800000u : 0u));
  };
  float b4s = 0.f;
  if (LOOP) {
    store_tables();
    __syncthreads();
    // (consumed HERE, where nothing else is in flight: its first use used to be conv4's epilogue, behind the layers' CONDITIONAL
    //  row stores -- the compiler cannot count those, so it waited with vmcnt(0) there: every x4 store of a lane drained the
    //  previous one, eight store round trips in a row per graph)
    b4s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b4v)));
  }
  const int rdoff = (4 * kq + (nl >> 2)) * 32 + 8 * ((nl & 3) ^ kq);
  const int mrow0 = 16 * wave + nl, mrow1 = mrow0 + 16 * WAVES;         // this lane's node in tile wave / tile wave + WAVES
  const int wsl = 8 * (kq ^ ((nl >> 2) & 3));
  int par = 0;
  CH_T(0);                                                // 0: set-up
  for (int r = 0; LOOP ? r * G < ns : r < 1; ++r) {
    float* dv = reinterpret_cast<float*>(smem + C::OFF_DV) + par * C::ROWS;
    unsigned short* h4p = reinterpret_cast<unsigned short*>(smem + C::OFF_H4) + par * 3 * C::ROWS;
    const int K32 = (n + 31) >> 5, T = (n + 15) >> 4, RU = 32 * K32;
    const int S = 1 << dgd_class(max(n, 1));
    // (a per-iteration copy of the thread index the compiler cannot see through: otherwise every address piece of the staging
    //  below is loop-invariant, gets hoisted out of the graph walk and SPILLED -- and each reload is a scratch load followed by
    //  s_waitcnt vmcnt(0), which also waits for the next graph's prefetch: 7 of them made conv4's 3 MFMAs take 5.7 k cycles)
    int tl = tid;
    DG_OPAQUE_V(tl);
    // graphs of <= ROWS/2 nodes keep TWO images, rows [0, ROWS/2) and [ROWS/2, ROWS) of every plane, and alternate between
    // them: a layer's output image is not the one still being read, so the barrier in front of its stores is not needed
    const int pong = (2 * n <= C::ROWS) ? (C::ROWS / 2) * 32 : 0;
    // ---- zeros first: they depend on nothing that was loaded, so their ~1 k cycles of stores are issued while the graph's data is
    //      still in flight (they used to stand behind the data stores, i.e. behind the wait for the prefetch).  Disjoint from every
    //      slot the staging below writes: no ordering between the two is needed.
    {   // zeros conv1 reads but nobody wrote: rows n..RU-1 of its planes, and the slots beyond the feature width
      const int sh = NBF + 1;                              // 4 * NBF slots per row
      for (int it = tl; it < (RU << sh); it += C::THREADS) {
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
      if (tl < 192) {
        const int piece = tl & 1, row = 16 * T + ((tl >> 1) & 15), pp = tl >> 5;
        if ((pp & 1) >= NBF) *reinterpret_cast<uint4*>(H + pp * C::PS + row * 32 + 16 * piece) = make_uint4(0u, 0u, 0u, 0u);
      }
      if (tl < 48) h4p[(tl >> 4) * C::ROWS + 16 * T + (tl & 15)] = 0;
      if (pong && tl < 192)
        *reinterpret_cast<uint4*>(H + pong + (tl >> 5) * C::PS + (16 * T + ((tl >> 1) & 15)) * 32 + 16 * (tl & 1)) = make_uint4(0u, 0u, 0u, 0u);
    }
    // ---- stage the graph: registers -> LDS images -----------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < C::PB; ++j)
      if (tl + C::THREADS * j < n * S) bl[tl + C::THREADS * j] = pbit[j];
    if (tl < C::ROWS) dv[tl] = tl < n ? pdv : 0.f;
    CH_T(14);
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int it = tl + C::THREADS * j, k = it >> lg, qq = it & ((1 << lg) - 1);
      if (k < n) {
        uint2 pk[3];
        ch_split3_pack4(4 * qq + 0 < F ? pxs[j][0] : 0.f, 4 * qq + 1 < F ? pxs[j][1] : 0.f, 4 * qq + 2 < F ? pxs[j][2] : 0.f,
                        4 * qq + 3 < F ? pxs[j][3] : 0.f, pk);
        const int nb = qq >> 2, sl = qq & 3;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(H + (p * 2 + nb) * C::PS + k * 32 + 8 * (sl ^ ((k >> 2) & 3))) = pk[p];
      }
    }
    CH_T(15);
    CH_T(1);                                              // 1: wait for the prefetched data + staging stores
    if (!LOOP) {          // (one graph per workgroup: the weight tables go to LDS only now, see the loads above)
      store_tables();
      b4s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b4v)));
    }
    dg_lds_barrier();
    CH_T(2);
    int n0N = 0, nN = 0;
    if (LOOP) {
      n0N = __builtin_amdgcn_readfirstlane(eN.x); nN = __builtin_amdgcn_readfirstlane(eN.y);     // (requested a graph ago)
      eN = entry_of(r + 2);
      prefetch(n0N, nN);
    }
    // this lane's bitmap rows stay in LDS (re-read per layer: 2 x K32 words; in registers they cost 16 at eight waves) ...
    const unsigned* bl0 = bl + min(mrow0, max(n - 1, 0)) * S;
    const unsigned* bl1 = bl + min(mrow1, max(n - 1, 0)) * S;
    // TWO = false (graphs of at most 16 * WAVES nodes: the one-launch training kernel): a wave never has a second tile -- known
    // at compile time, so none of its code exists -- and ... the ONE tile's words are read into registers once per graph: the
    // word read and the nibble-table lookup it feeds were two DEPENDENT LDS round trips per word in front of every layer's matrix
    // instructions (the per-word `if (u < K32)` blocks keep the compiler from hoisting them itself)
    constexpr bool TWO = MAXN > 16 * WAVES;
    const bool rv0 = mrow0 < n, rv1 = TWO && mrow1 < n;
    unsigned wreg[TWO ? 1 : C::KW];
    if (!TWO) {
#pragma unroll
      for (int u = 0; u < C::KW; ++u) wreg[u] = bl0[min(u, S - 1)];
#pragma unroll
      for (int u = 0; u < C::KW; ++u) wreg[u] = (rv0 && u < K32) ? wreg[u] : 0u;
    }
    const float dn[2] = {dv[mrow0], dv[mrow1]};
    const int mrow[2] = {mrow0, mrow1};
    CH_T(3);
    const bool live0 = wave < T, live1 = TWO && wave + WAVES < T;
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

    // block products of this wave's tiles with the image: every HS^T operand read once, used by both tiles; the reads of
    // word u+1 are issued before the matrix instructions of word u
    auto product = [&](auto nbc, auto pc, const char* Hc, f32x4 (&acc)[2][2]) {
      constexpr int NBP = decltype(nbc)::value, NP = decltype(pc)::value;       // planes, bf16 parts of the image
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) { acc[ti][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[ti][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      const char* hp = Hc + rdoff;
#pragma unroll
      for (int u = 0; u < C::KW; ++u) {
        if (u < K32) {
          const bf16x8 bop0 = ch_bits_operand(TWO ? (rv0 ? bl0[u] : 0u) : wreg[u], kq, tab);
          const bf16x8 bop1 = ch_bits_operand(rv1 ? bl1[u] : 0u, kq, tab);
          bf16x8 a[NP][NBP];
#pragma unroll
          for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int nb = 0; nb < NBP; ++nb) a[p][nb] = ch_read_hsT(hp + (p * 2 + nb) * C::PS + u * 1024);
#pragma unroll
          for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int nb = 0; nb < NBP; ++nb) acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][nb], bop0, acc[0][nb], 0, 0, 0);
          if (live1) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
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
      if (BF) {      // bf16 leg: ONE matrix instruction per 16 output columns, both operands rounded to bf16
        const uint4* wb = reinterpret_cast<const uint4*>(Wop);
        bf16x8 wa[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) { const uint4 q = wb[ob * 64 + lane]; wa[ob] = __builtin_bit_cast(bf16x8, q); }
#pragma unroll
        for (int ti = T0; ti < NT; ++ti) {
          uint4 xq;
          xq.x = ch_f2bf(v[ti][0][0]) | (ch_f2bf(v[ti][0][1]) << 16); xq.y = ch_f2bf(v[ti][0][2]) | (ch_f2bf(v[ti][0][3]) << 16);
          xq.z = ch_f2bf(v[ti][1][0]) | (ch_f2bf(v[ti][1][1]) << 16); xq.w = ch_f2bf(v[ti][1][2]) | (ch_f2bf(v[ti][1][3]) << 16);
          const bf16x8 xb = __builtin_bit_cast(bf16x8, xq);
#pragma unroll
          for (int ob = 0; ob < 2; ++ob) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ob], xb, d, 0, 0, 0);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) hs[ti][ob][rr] = dn[ti] * d[rr];
          }
        }
        return;
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
      if (BF) {      // one bf16 part, round to nearest even
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
          *reinterpret_cast<uint2*>(Hn + ob * C::PS + mrow[ti] * 32 + wsl) =
              make_uint2(ch_f2bf(hs[ob][0]) | (ch_f2bf(hs[ob][1]) << 16), ch_f2bf(hs[ob][2]) | (ch_f2bf(hs[ob][3]) << 16));
        return;
      }
#pragma unroll
      for (int ob = 0; ob < 2; ++ob) {
        uint2 pk[3];
        ch_split3_pack4(hs[ob][0], hs[ob][1], hs[ob][2], hs[ob][3], pk);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(Hn + (p * 2 + ob) * C::PS + mrow[ti] * 32 + wsl) = pk[p];
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
      using I3 = std::integral_constant<int, 3>;
      if (NBF == 2) product(I2{}, I3{}, H, acc); else product(I1{}, I3{}, H, acc);
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
      product(I2{}, std::integral_constant<int, BF ? 1 : 3>{}, H + pong, acc);
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
      product(I2{}, std::integral_constant<int, BF ? 1 : 3>{}, H, acc);
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
#ifdef CH_RACE_DELAY
    // test build only (tests/test_gpu_chain.py, variants/lib_racedelay.so): waves 0, 3, 6 enter the iteration's LAST phase ~30 k
    // cycles late, so that every other wave is done with it -- and, were the loop-end barrier missing, would be re-staging `bl`
    // for the next graph -- long before these waves read their bitmap words
    if (LOOP && wave % 3 == 0) { for (int q = 0; q < 4; ++q) __builtin_amdgcn_s_sleep(127); }
#endif
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
          a4[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch_bits_operand(TWO ? (rv0 ? bl0[u] : 0u) : wreg[u], kq, tab), bop, a4[0], 0, 0, 0);
          if (live1) a4[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch_bits_operand(rv1 ? bl1[u] : 0u, kq, tab), bop, a4[1], 0, 0, 0);
        }
      }
      float s1[2][4], s2[2][4];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { s1[ti][rr] = __shfl_xor(a4[ti][rr], 1); s2[ti][rr] = __shfl_xor(a4[ti][rr], 2); }
      if (nl == 0) {
        int kqv = kq;
        DG_OPAQUE_V(kqv);        // (see the staging: nothing below may be hoisted out of the graph walk)
#pragma unroll
        for (int ti = 0; ti < (TWO ? 2 : 1); ++ti) {
          const int mm = 16 * (wave + WAVES * ti) + 4 * kqv;
          const float4 dq = *reinterpret_cast<const float4*>(dv + mm);
          const float dd[4] = {dq.x, dq.y, dq.z, dq.w};
          // (all four values first, branch-free, then the predicated stores: as four `if (row < n) { tanh; store; store }` blocks
          //  each block was its own chain of LDS read -> transcendental -> store)
          float xq[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) xq[rr] = dg_tanh(fmaf(dd[rr], (a4[ti][rr] + s1[ti][rr]) + s2[ti][rr], b4s));
          if (!TWO && x4_lds) {
            if (mm < n) *reinterpret_cast<float4*>(x4_lds + mm) = make_float4(xq[0], xq[1], xq[2], xq[3]);      // (rows >= n: never read)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) if (mm + rr < n) x4[n0 + mm + rr] = xq[rr];
          } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
              if (mm + rr < n) { x4[n0 + mm + rr] = xq[rr]; if (x4_lds) x4_lds[mm + rr] = xq[rr]; }
          }
        }
      }
    }
    CH_T(10);
#ifdef CH_TIMING
    if (dbg && tid == 0) dbg[blockIdx.x * 16 + 11] += 1;
#endif
    // Loop end.  Two tiles per wave (TWO): conv4 reads its bitmap words from the LDS rows `bl` word by word; a wave that is done
    // with its tiles must not stage the NEXT graph's rows over them while another wave is still in conv4 -> one LDS-only barrier.
    // One tile per wave (!TWO): conv4 reads nothing the next staging writes -- its bitmap words are in registers (`wreg`, read
    // behind the staging barrier, and every wave has passed three more barriers since), h4p / dv alternate with `par`, the image H
    // was last read in conv3's product in front of the barrier at CH_T(8), the tables are constant -> no barrier needed.
#ifndef CH_NO_LOOP_END_BARRIER      // (the unsafe form exists only to show that the race-delay test catches its absence: profiles/r06_race_test.txt)
    if (LOOP && TWO) dg_lds_barrier();
#endif
    n0 = n0N; n = nN; par ^= 1;
  }
#ifdef CH_TIMING
  if (dbg && tid == 0) dbg[blockIdx.x * 16 + 13] = wall_clock64();
#endif
}

template <int WAVES, int XI, int W1S, bool LOOP, bool BF = false>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4)))      // <= 128 registers
k_chain_fwd_q(int N, int B, int F, const int* __restrict__ sched, const int* __restrict__ nbig_p, const int* __restrict__ graph_ptr,
              const unsigned* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ xs, ChW gw,
              float* __restrict__ axg, float* __restrict__ x1, float* __restrict__ x2, float* __restrict__ x3, float* __restrict__ x4,
              unsigned long long* __restrict__ dbg) {
  ch_chain_body<WAVES, XI, W1S, LOOP, 32 * WAVES, BF>(N, B, F, sched, nbig_p, graph_ptr, bits, dinv, xs, gw, axg, x1, x2, x3, x4, dbg);
}

// =================================================================================================================
// Training step of a SMALL batch (one graph per workgroup): the graph-chain forward, the readout forward (SortPooling +
// dense tail, dg_readout.h) and the readout backward (dg_tail_body.h) of a graph in ONE launch -- all three are per-graph
// chains on the same 1024 threads; the x1..x4 rows the chain just wrote are read back by the workgroup that wrote them.
// Rider range (blocks >= B): phase A of the next batch's graph preparation, as on k_readout_tail.
// =================================================================================================================
// conv4's backward (+ the start of conv3's) for ONE graph of <= 256 nodes on 16 waves, one tile per wave -- the first half of
// k_chain_bwd_a, appended to the one-launch training kernel of small batches (the separate k_gcn_bwd1 launch, 4.9 us at the
// dispatch floor, goes away): gh4 = dinv (Adj gas4), gas3 = dinv (gh4 W4 + gp3)(1 - x3^2) -> global, {dW4, db3} -> this
// graph's row of pa4.  `bl` (bitmap rows, stride S), `dv` (dinv) and `tab` are the chain forward's LDS images of this graph;
// g4p [3][ROWS] bf16, g4t [16 waves][16] and slots [16 waves][64] are scratch.
__device__ __forceinline__ void ch_conv4_bwd_graph(int n0, int n, const unsigned* bl, const float* dv, const uint2* tab,
                                                   unsigned short* g4p, int ROWS, float* g4t_all, float* slots_all,
                                                   const float* __restrict__ gas4, const float* W4,
                                                   const float* __restrict__ x3, const float* __restrict__ gp3,
                                                   float* __restrict__ gas3, float* __restrict__ pa4row) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  const int K32 = (n + 31) >> 5, T = (n + 15) >> 4, RU = 32 * K32;
  const int S = 1 << dgd_class(max(n, 1));
  float* g4t = g4t_all + wave * 16;
  float* slot = slots_all + wave * 64;
  if (tid < RU) {
    unsigned q0, q1, q2;
    ch_split3(tid < n ? gas4[n0 + tid] : 0.f, q0, q1, q2);
    g4p[tid] = (unsigned short)q0; g4p[ROWS + tid] = (unsigned short)q1; g4p[2 * ROWS + tid] = (unsigned short)q2;
  }
  if (lane < 64 && tid < 1024) slot[lane] = 0.f;
  const int m = 16 * wave + nl;
  const bool live = wave < T, ok = m < n;
  float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa, ga_ = xa, gb_ = xa, w4a = xa, w4b = xa;
  if (live) {       // operands of the epilogue, requested before the barrier
    const size_t ro = (size_t)(n0 + min(m, n - 1)) * 32 + 4 * kq;
    xa = *reinterpret_cast<const float4*>(x3 + ro); xb = *reinterpret_cast<const float4*>(x3 + ro + 16);
    ga_ = *reinterpret_cast<const float4*>(gp3 + ro); gb_ = *reinterpret_cast<const float4*>(gp3 + ro + 16);
    w4a = *reinterpret_cast<const float4*>(W4 + 4 * kq); w4b = *reinterpret_cast<const float4*>(W4 + 16 + 4 * kq);
  }
  dg_lds_barrier();
  if (live) {
    const unsigned* blr = bl + min(m, n - 1) * S;
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* hq = g4p + min(nl, 2) * ROWS + 4 * kq;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u < K32) {
        uint2 lo = *reinterpret_cast<const uint2*>(hq + 32 * u), hi = *reinterpret_cast<const uint2*>(hq + 32 * u + 16);
        if (nl >= 3) { lo = make_uint2(0u, 0u); hi = make_uint2(0u, 0u); }
        bf16x8 bop;
        unsigned* bu = reinterpret_cast<unsigned*>(&bop);
        bu[0] = lo.x; bu[1] = lo.y; bu[2] = hi.x; bu[3] = hi.y;
        a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch_bits_operand(ok ? blr[u] : 0u, kq, tab), bop, a4, 0, 0, 0);
      }
    }
    float tot[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) tot[rr] = (a4[rr] + __shfl_xor(a4[rr], 1)) + __shfl_xor(a4[rr], 2);
    if (nl == 0) {
      const float4 dq = *reinterpret_cast<const float4*>(dv + 16 * wave + 4 * kq);
      *reinterpret_cast<float4*>(g4t + 4 * kq) = make_float4(dq.x * tot[0], dq.y * tot[1], dq.z * tot[2], dq.w * tot[3]);
    }
    DG_LOCKSTEP();
    const float gh = g4t[nl];
    const float dn = dv[m];
    const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
    const float gv[8] = {ga_.x, ga_.y, ga_.z, ga_.w, gb_.x, gb_.y, gb_.z, gb_.w};
    const float wv[8] = {w4a.x, w4a.y, w4a.z, w4a.w, w4b.x, w4b.y, w4b.z, w4b.w};
    float go[8], s4[8], s3[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float ga = ok ? fmaf(gh, wv[c], gv[c]) * (1.f - xv[c] * xv[c]) : 0.f;
      go[c] = dn * ga;
      s4[c] = ch_row16_sum(ok ? gh * xv[c] : 0.f);
      s3[c] = ch_row16_sum(ga);
    }
    if (ok) {
      float* dst = gas3 + (size_t)(n0 + m) * 32 + 4 * kq;
      *reinterpret_cast<float4*>(dst) = make_float4(go[0], go[1], go[2], go[3]);
      *reinterpret_cast<float4*>(dst + 16) = make_float4(go[4], go[5], go[6], go[7]);
    }
    if (nl == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int col = 16 * (c >> 2) + 4 * kq + (c & 3);
        slot[col] = s4[c]; slot[32 + col] = s3[c];
      }
    }
  }
  dg_lds_barrier();
  if (tid < 64) {
    float a = 0.f;
    for (int wv_ = 0; wv_ < 16; ++wv_) a += slots_all[wv_ * 64 + tid];          // fixed order (dead waves left zeros)
    pa4row[tid] = a;
  }
}

// The WHOLE GCN backward (conv4, conv3, conv2 carrying conv1's weight gradient: aggregate-first conv1) of ONE graph of <= 256
// nodes on 16 waves, one tile per wave, appended to the one-launch training kernel: the arithmetic of ch_conv4_bwd_graph /
// k_chain_bwd_a / k_chain_bwd_b per graph, with gas3 AND gas2 in ONE LDS image (gas2 overwrites gas3 in place behind a
// barrier) -- nothing of the GCN backward crosses a launch boundary, the step is this kernel + k_wgrad.
// Differences from the multi-launch forms, all for the latency chain of a single graph:
//   * the SortPooling gradient arrives in LDS in its sparse form (TbLds: 30 rows + a node -> row map + gas4): no dense slabs;
//   * the block product runs ONCE (lane = node); the lane = column copies the weight gradients need (gh, x2, x1, gp1) come from
//     wave-private LDS transposes of a 16 x 32 tile -- no second product in the other orientation, no transposed global gathers
//     (a dword-per-lane load costs the CU's address path ~14 cycles per wave-instruction: 20 of them per wave were 2.2 k cycles);
//   * column sums over the tile's 16 nodes (dW4, db3, db2) through the same tile instead of 4-step DPP row sums per value;
//   * W3 / W2 are read from the FORWARD's operand-order tables, still in LDS (a gather with 4-way bank conflicts, 16 reads),
//     W4 from its bias table: no global reload, no table rebuild;
//   * every global operand of a layer (x3 | x2 | x1, ax rows) is requested one phase ahead.
//   H: the image, plane stride PS; bl/dv/tab: the chain forward's LDS images of this graph; W3op / W2op / W4: the forward's
//   tables; g4p [3][ROWS] bf16 and g4t [16 waves][16]: conv4's scratch; gt_all [16 waves][16][CH_GT_LD]: tiles;
//   slots [16 waves][128]: dW4 | db3 | db2 | db1; red: >= 32 x 4 KB of LDS that is dead by the end (cross-wave sums).
// Output: this graph's row of pa4 (dW4 | db3), pb3 (dW3 [32][32] | db2), pb2 (dW2 | db1), pb1 (dW1 [32][Fa]).
#define CH_GT_LD 36
template <int NBA>
__device__ __forceinline__ void ch_gcn_bwd_graph(int n0, int n, int Fa, char* H, int PS, const unsigned* bl, const float* dv,
                                                 const uint2* tab, const float* W3op, const float* W2op, const float* W4,
                                                 unsigned short* g4p, int ROWS, float* g4t_all, float* gt_all,
                                                 float* slots_all, float* red, const TbLds L,
                                                 const float* __restrict__ x3, const float* __restrict__ x2,
                                                 const float* __restrict__ x1, const float* __restrict__ axg,
                                                 float* __restrict__ pa4row, float* __restrict__ pb3row,
                                                 float* __restrict__ pb2row, float* __restrict__ pb1row,
                                                 unsigned long long* dbg = nullptr) {
#define GB_MARK(k) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[k] = clock64(); } while (0)
#ifdef CH_FINE
#define GB_FINE(k) GB_MARK(k)
#else
#define GB_FINE(k) do { } while (0)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  const int K32 = (n + 31) >> 5, T = (n + 15) >> 4, RU = 32 * K32;
  const int S = 1 << dgd_class(max(n, 1));
  float* slot = slots_all + wave * 128;
  float* gt = gt_all + wave * (16 * CH_GT_LD);
  float* g4t = g4t_all + wave * 16;
  const int rdoff = (4 * kq + (nl >> 2)) * 32 + 8 * ((nl & 3) ^ kq);
  const int wsl = 8 * (kq ^ ((nl >> 2) & 3));
  const int m = 16 * wave + nl, mt = 16 * wave;
  const bool live = wave < T, ok = m < n;
  const unsigned* blr = bl + min(m, max(n - 1, 0)) * S;
  unsigned wreg[8];                  // this lane's bitmap row, once for the three layers (see the forward)
#pragma unroll
  for (int u = 0; u < 8; ++u) wreg[u] = blr[min(u, S - 1)];
#pragma unroll
  for (int u = 0; u < 8; ++u) wreg[u] = (ok && u < K32) ? wreg[u] : 0u;
  const float dn = dv[m];
  // index of W[o = 16 (s>>2) + 4 kq + (s&3)][k = 16 kb + nl] in the forward's table: wlane + ((s>>2) * 8 + kb * 4) * 64 + (s&3)
  const int wlane = ((nl & 3) << 6) + 4 * kq + ((nl >> 2) << 4);
  f32x4 accW3[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  f32x4 accW2[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  f32x4 accA[2][NBA];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBA; ++nb) accA[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // block product of this wave's tile with the image, lane = node: acc[nb][rr] = (Adj . img)[node nl][16 nb + 4 kq + rr]
  auto product = [&](f32x4 (&acc)[2]) {
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* hp = H + rdoff;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u < K32) {
        const bf16x8 bop = ch_bits_operand(wreg[u], kq, tab);
        bf16x8 a[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) a[p][nb] = ch_read_hsT(hp + (p * 2 + nb) * PS + u * 1024);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][nb], bop, acc[nb], 0, 0, 0);
      }
    }
  };
  // the wave's 16 x 32 tile (row = node of the tile): a lane puts the 8 values of its node (columns 4 kq .. +3, 16 + 4 kq .. +3) ...
  auto tile_put = [&](const float (&v)[8]) {
    DG_LOCKSTEP();      // (every lane is done with the tile's previous contents)
    *reinterpret_cast<float4*>(gt + nl * CH_GT_LD + 4 * kq) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(gt + nl * CH_GT_LD + 16 + 4 * kq) = make_float4(v[4], v[5], v[6], v[7]);
  };
  // ... and takes the lane = column layout: out[mb][s] = tile[node 4 kq + s][16 mb + nl]      (same wave: program order)
  auto tile_cols = [&](float (&out)[2][4]) {
    DG_LOCKSTEP();      // (every lane has put its row)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) out[mb][s_] = gt[(4 * kq + s_) * CH_GT_LD + 16 * mb + nl];
  };
  // ... or the column sums over the 16 nodes: lane l receives column l & 31 (both halves of the wave hold it)
  auto tile_colsum = [&]() {
    DG_LOCKSTEP();
    const float* cp = gt + (lane >> 5) * (8 * CH_GT_LD) + (lane & 31);
    float a = cp[0];
#pragma unroll
    for (int r = 1; r < 8; ++r) a += cp[r * CH_GT_LD];
    return a + __shfl_xor(a, 32);
  };
  // rows of the sparse SortPooling gradient (columns c0 .. of gpL) of this lane's node, zero for a node that was not selected
  auto gp_row = [&](int slot_, int c0, float (&g)[8]) {
    // (two unconditional 16-byte reads on a clamped row, selected afterwards: written as `slot >= 0 ? load : 0` they became
    //  eight predicated dword reads, each in an exec-mask block of its own)
    const float* r = L.gpL + max(slot_, 0) * 96 + c0 + 4 * kq;
    const float4 a = *reinterpret_cast<const float4*>(r), b = *reinterpret_cast<const float4*>(r + 16);
    const bool on = slot_ >= 0;
    g[0] = on ? a.x : 0.f; g[1] = on ? a.y : 0.f; g[2] = on ? a.z : 0.f; g[3] = on ? a.w : 0.f;
    g[4] = on ? b.x : 0.f; g[5] = on ? b.y : 0.f; g[6] = on ? b.z : 0.f; g[7] = on ? b.w : 0.f;
  };
  auto image_store = [&](const float (&go)[8]) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      uint2 pk[3];
      ch_split3_pack4(go[4 * hb], go[4 * hb + 1], go[4 * hb + 2], go[4 * hb + 3], pk);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(H + (p * 2 + hb) * PS + m * 32 + wsl) = pk[p];
    }
  };
  const size_t ro = (size_t)(n0 + min(m, max(n - 1, 0))) * 32 + 4 * kq;
  const int myslot = ok ? L.slotmap[m] : -1;
  // ======== conv4 backward (+ the start of conv3's): gh4 = dinv (Adj gas4), gas3 = dinv (gh4 W4 + gp3)(1 - x3^2) -> image ========
  // (UNCONDITIONAL loads on clamped addresses, dead waves included: behind `if (live)` the compiler parks the result in a temporary
  //  and waits for it on the spot -- the whole round trip exposed where the load was meant to be a phase ahead)
  const float4 x3a = *reinterpret_cast<const float4*>(x3 + ro), x3b = *reinterpret_cast<const float4*>(x3 + ro + 16);
  const float4 x2a = *reinterpret_cast<const float4*>(x2 + ro), x2b = *reinterpret_cast<const float4*>(x2 + ro + 16);      // (conv3's, a phase ahead)
  if (tid < RU) {
    unsigned q0, q1, q2;
    ch_split3(tid < 256 ? L.gas4L[tid] : 0.f, q0, q1, q2);        // (zero beyond the selected nodes)
    g4p[tid] = (unsigned short)q0; g4p[ROWS + tid] = (unsigned short)q1; g4p[2 * ROWS + tid] = (unsigned short)q2;
  }
  if (RU > 16 * T && tid < 192)      // rows 16T .. RU-1 of the image: written by no tile
    *reinterpret_cast<uint4*>(H + (tid >> 5) * PS + (16 * T + ((tid >> 1) & 15)) * 32 + 16 * (tid & 1)) = make_uint4(0u, 0u, 0u, 0u);
  GB_FINE(40);
  dg_lds_barrier();
  GB_FINE(41);
  if (live) {
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* hq = g4p + min(nl, 2) * ROWS + 4 * kq;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u < K32) {
        uint2 lo = *reinterpret_cast<const uint2*>(hq + 32 * u), hi = *reinterpret_cast<const uint2*>(hq + 32 * u + 16);
        if (nl >= 3) { lo = make_uint2(0u, 0u); hi = make_uint2(0u, 0u); }
        bf16x8 bop;
        unsigned* bu = reinterpret_cast<unsigned*>(&bop);
        bu[0] = lo.x; bu[1] = lo.y; bu[2] = hi.x; bu[3] = hi.y;
        a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch_bits_operand(wreg[u], kq, tab), bop, a4, 0, 0, 0);
      }
    }
    GB_FINE(42);
    float tot[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) tot[rr] = (a4[rr] + __shfl_xor(a4[rr], 1)) + __shfl_xor(a4[rr], 2);
    if (nl == 0) {
      const float4 dq = *reinterpret_cast<const float4*>(dv + mt + 4 * kq);
      *reinterpret_cast<float4*>(g4t + 4 * kq) = make_float4(dq.x * tot[0], dq.y * tot[1], dq.z * tot[2], dq.w * tot[3]);
    }
    DG_LOCKSTEP();
    const float gh = g4t[nl];
    const float4 w4a = *reinterpret_cast<const float4*>(W4 + 4 * kq), w4b = *reinterpret_cast<const float4*>(W4 + 16 + 4 * kq);
    const float xv[8] = {x3a.x, x3a.y, x3a.z, x3a.w, x3b.x, x3b.y, x3b.z, x3b.w};
    const float wv[8] = {w4a.x, w4a.y, w4a.z, w4a.w, w4b.x, w4b.y, w4b.z, w4b.w};
    float gv[8], go[8], t4[8], t3[8];
    gp_row(myslot, 64, gv);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float ga = ok ? fmaf(gh, wv[c], gv[c]) * (1.f - xv[c] * xv[c]) : 0.f;
      go[c] = dn * ga;
      t4[c] = ok ? gh * xv[c] : 0.f;
      t3[c] = ga;
    }
    GB_FINE(43);
    tile_put(t4);
    const float s4 = tile_colsum();
    tile_put(t3);
    const float s3 = tile_colsum();
    if (lane < 32) { slot[lane] = s4; slot[32 + lane] = s3; }          // dW4 | db3
    GB_FINE(44);
    image_store(go);
    GB_FINE(45);
  }
  dg_lds_barrier();
  GB_MARK(17);
  // ======== conv3 backward ========
  float go2[8];
  // conv2's global operands, a phase ahead (unconditional, see above)
  const float4 x1a = *reinterpret_cast<const float4*>(x1 + ro), x1b = *reinterpret_cast<const float4*>(x1 + ro + 16);
  float aN[NBA][4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    const int mm = mt + 4 * kq + s_;
    const size_t rr_ = (size_t)(n0 + min(mm, max(n - 1, 0)));
#pragma unroll
    for (int nb = 0; nb < NBA; ++nb) aN[nb][s_] = axg[rr_ * Fa + min(16 * nb + nl, Fa - 1)];
  }
  if (live) {
    const float xv[8] = {x2a.x, x2a.y, x2a.z, x2a.w, x2b.x, x2b.y, x2b.z, x2b.w};
    float xm[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) xm[c] = ok ? xv[c] : 0.f;
    float xN[2][4];                  // x2 in the lane = column layout: rows 4kq + s of column 16nb + nl (B operand of dW3)
    tile_put(xm);
    tile_cols(xN);
    GB_FINE(32);
    f32x4 accT[2];
    product(accT);
    GB_FINE(33);
    float ghT[8], ghN[2][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) ghT[c] = dn * accT[c >> 2][c & 3];
    tile_put(ghT);
    tile_cols(ghN);
    GB_FINE(34);
    f32x4 gx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};       // gx2 = gh W3 (A = W3 gathered, B = gh lane = node)
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        gx[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(W3op[wlane + (((s_ >> 2) * 8 + kb * 4) << 6) + (s_ & 3)], ghT[s_], gx[kb], 0, 0, 0);
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)     // dW3 += gh^T x2
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          accW3[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ghN[mb][s_], xN[nb][s_], accW3[mb][nb], 0, 0, 0);
    GB_FINE(35);
    float gpv[8], t2[8];
    gp_row(myslot, 32, gpv);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float ga = ok ? (gx[c >> 2][c & 3] + gpv[c]) * (1.f - xv[c] * xv[c]) : 0.f;
      go2[c] = dn * ga;
      t2[c] = ga;
    }
    tile_put(t2);                      // (db2's column sums are taken behind the image store: the round trip hides under its VALU work)
  }
  GB_FINE(36);
  dg_lds_barrier();                  // every wave has read the gas3 image: gas2 overwrites it in place
  GB_FINE(37);
  if (live) {
    image_store(go2);
    const float s2 = tile_colsum();
    if (lane < 32) slot[64 + lane] = s2;                                // db2
  }
  GB_FINE(38);
  dg_lds_barrier();
  GB_MARK(18);
  // ======== conv2 backward + conv1's weight gradient ========
  if (live) {
    const float x1v[8] = {x1a.x, x1a.y, x1a.z, x1a.w, x1b.x, x1b.y, x1b.z, x1b.w};
    float xm[8], gm[8], xN[2][4], gN[2][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) xm[c] = ok ? x1v[c] : 0.f;
    tile_put(xm);
    tile_cols(xN);
    gp_row(myslot, 0, gm);               // (rows >= n: myslot = -1 -> zeros)
    tile_put(gm);
    tile_cols(gN);
    f32x4 accT[2];
    product(accT);
    float ghT[8], ghN[2][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) ghT[c] = dn * accT[c >> 2][c & 3];
    tile_put(ghT);
    tile_cols(ghN);
    f32x4 gx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};       // gx1 (lane = column): A = gh lane = node, B = W2 gathered
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        gx[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ghT[s_], W2op[wlane + (((s_ >> 2) * 8 + kb * 4) << 6) + (s_ & 3)], gx[kb], 0, 0, 0);
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)       // dW2 += gh^T x1
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          accW2[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ghN[mb][s_], xN[nb][s_], accW2[mb][nb], 0, 0, 0);
    float gaN[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float sb = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        gaN[kb][s_] = (gx[kb][s_] + gN[kb][s_]) * (1.f - xN[kb][s_] * xN[kb][s_]);      // (rows >= n: gx = 0, gp = 0)
        sb += gaN[kb][s_];
      }
      sb += __shfl_xor(sb, 16);
      sb += __shfl_xor(sb, 32);
      if (kq == 0) slot[96 + 16 * kb + nl] = sb;                                         // db1
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)       // dW1 += ga1^T ax
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBA; ++nb) {
          const float av = (mt + 4 * kq + s_ < n && 16 * nb + nl < Fa) ? aN[nb][s_] : 0.f;
          accA[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gaN[mb][s_], av, accA[mb][nb], 0, 0, 0);
        }
  }
  dg_lds_barrier();                  // every wave is done with the image and the tables: `red` may alias them
  GB_MARK(19);
  // ---- the graph's partial rows: the live waves' accumulators summed in wave order -------------------------------------------
  float* red2 = red + 16 * 1024;
  // dW1 joins the first round when its 16 x 32 x Fa floats fit behind the live waves' dW3 rows (T <= 12, Fa <= 8)
  const bool one_round = T <= 12 && 32 * Fa <= 256;
  float* redA = red + 12 * 1024;
  auto put_acc_a = [&](float* my, int ld) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int f = nb * 16 + nl;
          if (f < ld) my[(mb * 16 + kq * 4 + rr) * ld + f] = accA[mb][nb][rr];
        }
  };
  // (a wave's accumulators go out LANE-MAJOR -- element ((mb, nb), lane, rr) at ((2 mb + nb) 64 + lane) 4 + rr: one 16-byte store
  //  per accumulator, conflict-free -- and summing thread t takes element t of that order and works out which weight it is:
  //  row-major [o][k] stores from the accumulator layout were 32 dword stores per lane with 4-way bank conflicts)
  if (live) {
    float* my3 = red + wave * 1024;
    float* my2 = red2 + wave * 1024;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        *reinterpret_cast<float4*>(my3 + ((2 * mb + nb) * 64 + lane) * 4) = make_float4(accW3[mb][nb][0], accW3[mb][nb][1], accW3[mb][nb][2], accW3[mb][nb][3]);
        *reinterpret_cast<float4*>(my2 + ((2 * mb + nb) * 64 + lane) * 4) = make_float4(accW2[mb][nb][0], accW2[mb][nb][1], accW2[mb][nb][2], accW2[mb][nb][3]);
      }
    if (one_round) put_acc_a(redA + wave * 256, 8);
  }
  dg_lds_barrier();
  {
    const int q_ = tid >> 8, l_ = (tid >> 2) & 63;
    const int wel = ((q_ >> 1) * 16 + (l_ >> 4) * 4 + (tid & 3)) * 32 + (q_ & 1) * 16 + (l_ & 15);      // this thread's element of W [32][32]
    float a3 = 0.f, a2 = 0.f;
#pragma unroll 1
    for (int w0 = 0; w0 < T; w0 += 4) {      // four waves' rows per trip: eight reads in flight (all 32 at once spilled)
      float v3[4], v2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int wc = min(w0 + u, T - 1); v3[u] = red[wc * 1024 + tid]; v2[u] = red2[wc * 1024 + tid]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (w0 + u < T) { a3 += v3[u]; a2 += v2[u]; }
    }
    pb3row[wel] = a3; pb2row[wel] = a2;
    if (tid < 128) {                   // dW4 | db3 | db2 | db1
      float a = 0.f;
      for (int wv_ = 0; wv_ < T; ++wv_) a += slots_all[wv_ * 128 + tid];
      if (tid < 64) pa4row[tid] = a; else if (tid < 96) pb3row[1024 + tid - 64] = a; else pb2row[1024 + tid - 96] = a;
    }
    if (one_round && tid >= 128 && tid < 128 + 32 * Fa) {
      const int t = tid - 128, o = t / Fa, f = t - o * Fa;
      float a = 0.f;
      for (int wv_ = 0; wv_ < T; ++wv_) a += redA[wv_ * 256 + o * 8 + f];
      pb1row[t] = a;                                                                   // W1's own [32,Fa] layout
    }
  }
  GB_MARK(20);
  if (!one_round) {
    dg_lds_barrier();
    if (live) put_acc_a(red + wave * 1024, 32);
    dg_lds_barrier();
    for (int t = tid; t < 32 * Fa; t += 1024) {
      const int o = t / Fa, f = t - o * Fa;
      float a = 0.f;
      for (int wv_ = 0; wv_ < T; ++wv_) a += red[wv_ * 1024 + o * 32 + f];
      pb1row[t] = a;                                                                   // W1's own [32,Fa] layout
    }
  }
}

#ifndef CH_RIDER_ITEMS
#define CH_RIDER_ITEMS 2        // fused preparation (both phases in the training kernel's launch): work items per rider thread
#endif
#define CH_EVAL_MAXN 512       // largest graph of the one-launch evaluation kernel's two-tiles-per-wave form (round 6)
#define CH_TRAIN_MAXN 256      // largest graph of the one-launch training kernel (host hint max_nodes, verified: a larger one is flagged)
// the rider range of a one-launch kernel (blocks >= B): phase A of the next batch's graph preparation (or its whole assembly from a
// prepared dataset), and -- `rd.fused_b` -- phase B of the same batch behind it in the same launch.  Shared by the training kernel
// and the evaluation kernel.
__device__ __forceinline__ void ch_rider_block(int rb, const DgPrepRider& rd) {
  if (rb < rd.nblk) {
    if (rd.fused_b > 0) {
      // phase A with agent-coherent stores of what phase B reads (dg_prep.h: no fence -- a release would write back this XCD's
      // whole L2); __syncthreads() waits for every store of the workgroup (vmcnt(0)), then ONE relaxed increment publishes it
#pragma unroll 1
      for (int it = 0; it < CH_RIDER_ITEMS; ++it)      // (CH_RIDER_ITEMS items per thread: both phases' workgroups resident at once)
        dg_rider_phase_a<true>((rb * CH_RIDER_ITEMS + it) * RD_THREADS + (int)threadIdx.x, rd);
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(rd.sync_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      dg_rider_phase_a(rb * RD_THREADS + (int)threadIdx.x, rd);
    }
    return;
  }
  // phase B of the same batch in the same launch: wait until EVERY phase-A workgroup has published (bounded; they were all
  // dispatched before this workgroup), then the body that otherwise rides on k_wgrad.  The LDS row buffer is this launch's
  // dynamic LDS, which a rider workgroup does not use otherwise.
  DG_DYN_SMEM(char, rsm);
  if (threadIdx.x == 0) {
    unsigned int spins = 0;
    // (relaxed polls and NO acquire fence: an acquire invalidates this XCD's L2, out of which the graph workgroups of the same
    //  launch live -- one per phase-B workgroup made the launch 80 us instead of 36; phase B reads phase A's outputs by
    //  agent-coherent loads instead)
    while ((int)(__hip_atomic_load(rd.sync_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - rd.sync_target) < 0) {
      __builtin_amdgcn_s_sleep(16);
      // (never seen on an exclusive device; the batch is flagged through a word of its own -- err[4] / err[6]: "an in-launch wait
      //  timed out", not "layout promise violated" -- and its structures stay incomplete)
      if (++spins > (1u << 23)) { rd.err[4] = rd.epoch; rd.err[6] = ~rd.epoch; break; }
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < CH_RIDER_ITEMS; ++it) {
    const int tb = ((rb - rd.nblk) * CH_RIDER_ITEMS + it) * RD_THREADS + (int)threadIdx.x;
    dg_prep_fast_b_body<RD_THREADS, true, true>(tb, rd.ei, rd.E, rd.N, rd.B, rd.rowptr, rd.colidx, rd.graph_ptr, rd.graph_eptr, rd.dinv,
                                                rd.err, rd.epoch, rd.x, rd.xs, rd.F, rd.batch, rd.bits, rd.dmap, rd.edge_check == 1,
                                                rd.max_nodes, reinterpret_cast<unsigned int*>(rsm));
  }
  // (no planning workgroup here: the launcher fuses only riders without an item table)
}
// Reverse-edge half of the coalesced + undirected promise, verified by the one-launch kernels themselves (api.hip: edge_check == 2,
// batches of DG_INSYM_MIN_B graphs or more, where phase B of the preparation rides on k_wgrad and its per-edge searches were that
// launch's duration): the graph's bit matrix must be SYMMETRIC.  Checked on the chain forward's LDS image of the bitmap (`bl`,
// row stride S; untouched by the readout) by the 14 waves that idle while waves 0 and 1 run conv5 on the matrix cores: item
// (row i, word k) -- for every set bit j of the word, bit i of row j must be set; an asymmetric pair flags the batch (err[1],
// epoch-tagged).  Nothing waits for it.  (First placement: the tile-less waves during conv1 -- the step kernel 38.7 -> 40.5 us at
// 256 graphs, they share the SIMDs of the waves that carry the chain.)
#ifndef CH_EARLY_WPRE
#define CH_EARLY_WPRE 0      // 1: classifier_1's rows for the backward requested in the forward's fc2 phase (candidate of round 5, never yet run on a GPU: off; 0: at the backward's start)
#endif
struct ChSymHook {
  const unsigned int* bl; int n, S, K32; unsigned int* err; unsigned int epoch;
  float4* wpre; const float* Wf1;      // training kernel: classifier_1's rows for the backward, requested in the forward's fc2 phase
  __device__ __forceinline__ void at_fc2(int tid) const {
    if (!wpre) return;
    // (same mapping as dg_tail_bwd_body: 704 threads = 8 row groups x 88 column quads, 16 x 16-byte loads each; unconditional on a
    //  clamped thread so that the loads are countable and nothing waits for them here)
    const int tc = min(tid, 2 * DGCNN_FLAT - 1), rg = tc / 88, mq = tc - rg * 88;
    const float* wc = Wf1 + (size_t)(rg * 16) * DGCNN_FLAT + 4 * mq;
#pragma unroll
    for (int j = 0; j < 16; ++j) wpre[j] = *reinterpret_cast<const float4*>(wc + (size_t)j * DGCNN_FLAT);
  }
  __device__ __forceinline__ void operator()(int wv, int lane) const {
    if (!err || wv < 2) return;
    unsigned int okb = 1u;
    for (int it = ((wv - 2) << 6) + lane; it < 8 * n; it += 14 * 64) {
      const int i = it >> 3, k = it & 7;
      unsigned int wq = k < K32 ? bl[i * S + min(k, S - 1)] : 0u;
      const unsigned int* colw = bl + (i >> 5);
      while (wq) {
        const int j0 = 32 * k + __builtin_ctz(wq); wq &= wq - 1u;
        const int j1 = wq ? 32 * k + __builtin_ctz(wq) : j0; wq &= wq - (wq ? 1u : 0u);
        const unsigned int r0 = colw[min(j0, n - 1) * S], r1 = colw[min(j1, n - 1) * S];      // (bits beyond n: flagged by graph preparation)
        okb &= (r0 & r1) >> (i & 31);
      }
    }
    if (!(okb & 1u)) { err[1] = epoch; err[3] = ~epoch; }
  }
};
struct ChTail {
  unsigned int* err; unsigned int epoch; int insym;      // insym: verify the bitmap's symmetry here (see ch_chain_body)
  const float* W4; float* gas3; float* pa4; int P1;      // conv4's backward rides along when pa4 != null (P1 >= B rows)
  float *pb3, *pb2, *pb1;                                // != null: conv3 / conv2 / conv1's backward too (one partial row per graph)
  int C; TailW w; float* pooled; int* perm; float *a5g, *a6g, *a1dg; uint8_t* maskg; float* logp; int training; uint64_t seed;
  const int64_t* y; float loss_scale; float *dlogit, *gz1g, *gz6g, *gz5g, *gp1, *gp2, *gp3, *gas4, *gb4p, *lossv, *ptail;
};
// LDS of the one-launch training kernel behind the chain forward's plan: scratch of the GCN backward that must not alias
// anything the backward still reads (the cross-wave sums of dW3 / dW2 take the first 128 KB)
template <int W1S>
struct ChTrainLds {
  using Q = ChQ<16, W1S, CH_TRAIN_MAXN>;
  static constexpr int PS = CH_TRAIN_MAXN * 32;          // plane stride of the gas3 / gas2 image (one tile per wave: 256 rows)
  static constexpr int OFF_GT = 6 * PS;                  // behind the image: [16 waves][16][CH_GT_LD] tiles
  static constexpr int OFF_GPL = OFF_GT + 16 * 16 * CH_GT_LD * 4;      // the selected nodes' gradient rows [30][96]
  static_assert(OFF_GPL + 30 * 96 * 4 <= Q::OFF_W1, "image + tiles + gradient rows inside the forward's image region");
  static_assert(OFF_GPL >= 65536, "the gradient rows are written while the readout's LDS plan (first 64 KB) is live");
  static constexpr int BASE = Q::TOTAL > 131072 ? Q::TOTAL : 131072;      // (the cross-wave sums take the first 128 KB)
  static constexpr int OFF_SL = BASE;                    // [16 waves][128]: dW4 | db3 | db2 | db1
  static constexpr int OFF_G4T = OFF_SL + 8192;          // [16 waves][16]: gh4 of the wave's tile
  static constexpr int TOTAL = OFF_G4T + 1024;
};
template <int XI, int W1S, bool BF = false>      // BF: the bf16 leg's forward (hs_2 / hs_3 as one bf16 part); the backward is fp32 in either case
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4)))
k_chain_readout_tail(int N, int B, int F, const int* __restrict__ graph_ptr, const unsigned* __restrict__ bits,
                     const float* __restrict__ dinv, const float* __restrict__ xs, ChW gw, float* __restrict__ axg,
                     float* __restrict__ x1, float* __restrict__ x2, float* __restrict__ x3, float* __restrict__ x4, ChTail t,
                     unsigned long long* __restrict__ dbg, DgPrepRider rd) {
  if ((int)blockIdx.x >= B) { ch_rider_block((int)blockIdx.x - B, rd); return; }
  if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[15] = clock64();
  const int yb = (threadIdx.x < 64) ? (int)t.y[blockIdx.x] : 0;
  // this graph's node range, read ONCE (scalar loads; the three later phases re-read it behind their barriers: a scalar-memory
  // round trip in front of each phase's first address)
  const int gn0 = graph_ptr[blockIdx.x], gn = graph_ptr[blockIdx.x + 1] - gn0;
  if (threadIdx.x == 0 && gn > CH_TRAIN_MAXN) { t.err[1] = t.epoch; t.err[3] = ~t.epoch; }
  // (the keys' LDS copy lives in the unused second parity set of the dinv array: beyond the readout's LDS plan, which aliases
  //  the images, and untouched until conv4's backward at the end of this kernel reads the FIRST set)
  DG_DYN_SMEM(char, smem);
  float* keys_lds = reinterpret_cast<float*>(smem + ChQ<16, W1S, CH_TRAIN_MAXN>::OFF_DV) + ChQ<16, W1S, CH_TRAIN_MAXN>::ROWS;
#ifdef CH_FINE
  unsigned long long* const chain_dbg = dbg;
#else
  unsigned long long* const chain_dbg = nullptr;
#endif
  ch_chain_body<16, XI, W1S, false, CH_TRAIN_MAXN, BF>(N, B, F, nullptr, nullptr, graph_ptr, bits, dinv, xs, gw, axg, x1, x2, x3, x4,
                                                       chain_dbg, keys_lds);
  __syncthreads();        // (full barrier, vmcnt(0): this graph's x1..x4 rows are written; the LDS images are dead)
  if (BF && t.pa4 && t.pb3) {
    // bf16 leg: the forward's W2 / W3 tables hold bf16 operands; the fp32 backward below reads fp32 tables in the fp32 forward's
    // operand order -- rebuilt here, in place (the forward is done with them; nothing of the readout touches the region)
    float* W2op = reinterpret_cast<float*>(smem + ChQ<16, W1S, CH_TRAIN_MAXN>::OFF_W2);
    float* W3op = reinterpret_cast<float*>(smem + ChQ<16, W1S, CH_TRAIN_MAXN>::OFF_W3);
    const int e = threadIdx.x, o = e >> 5, k = e & 31;
    const int d = (((o >> 4) * 8 + ((k >> 4) << 2) + (k & 3)) << 6) + (o & 15) + (((k >> 2) & 3) << 4);
    const float w2 = gw.W2[e], w3 = gw.W3[e];
    W2op[d] = w2; W3op[d] = w3;
  }
  TbExt ext{};
  float4 wpre_regs[16];      // classifier_1's rows for the backward half (this thread's 16 x 16 bytes), in flight from the forward's fc2 phase on
  {
    const RdSmem M = dg_rd_carve(smem, smem + RD_REGION0_BYTES);
    const float* sp = reinterpret_cast<const float*>(M.region0);
    ext.sp = sp; ext.W5s = sp + 2912; ext.W6s = sp + 2912 + NW5; ext.lg = M.lg;
    ext.flat = M.flat; ext.a5s = M.a5s; ext.a1s = M.a1s; ext.sel = M.sel;
    ext.yb = yb;
    ext.wf2s = M.cpart;       // (rows of the first 16 classes, staged by the forward half; the backward prefetches 8, reads the rest in place)
    ext.x4l = keys_lds; ext.dvl = reinterpret_cast<const float*>(smem + ChQ<16, W1S, CH_TRAIN_MAXN>::OFF_DV);
    ext.n0 = gn0; ext.n = gn;
    const int b = blockIdx.x;
    const int n0 = gn0, n = gn;
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[14] = clock64();
    const int nn_ = min(gn, CH_TRAIN_MAXN);
    const ChSymHook hook{reinterpret_cast<const unsigned int*>(smem + ChQ<16, W1S, CH_TRAIN_MAXN>::OFF_BL), nn_, 1 << dgd_class(max(nn_, 1)),
                         (nn_ + 31) >> 5, t.insym ? t.err : nullptr, t.epoch, CH_EARLY_WPRE ? wpre_regs : nullptr, t.w.Wf1};
    ext.wpre = CH_EARLY_WPRE ? wpre_regs : nullptr;
    dg_readout_fwd_body(M, b, n0, n, t.C, t.w, keys_lds, 0, x1, x2, x3, x4, t.pooled, t.perm, t.a5g, t.a6g, t.a1dg, t.maskg, t.logp,
                        t.training, t.seed, dbg, hook);
  }
  __syncthreads();
  using C = ChQ<16, W1S, CH_TRAIN_MAXN>;
  using X = ChTrainLds<W1S>;
  const bool full = t.pa4 && t.pb3;      // the whole GCN backward of this graph follows in this workgroup
  TbLds L{};
  if (full) {   // ... and takes the SortPooling gradient from LDS (regions nothing of the readout / its backward touches)
    L.gpL = reinterpret_cast<float*>(smem + X::OFF_GPL);
    L.gas4L = reinterpret_cast<float*>(smem + C::OFF_H4 + 3072);     // (the second parity set of h4s: unused with one graph per workgroup)
    L.slotmap = reinterpret_cast<int*>(smem + C::OFF_H4 + 4096);
  }
  dg_tail_bwd_body<false, true, true, true>((int)blockIdx.x, B, t.C, t.w, graph_ptr, t.perm, dinv, x4, t.a5g, t.a6g, t.a1dg, t.logp, nullptr, t.y,
                                            t.loss_scale, t.training, t.dlogit, t.gz1g, t.gz6g, t.gz5g, t.gp1, t.gp2, t.gp3, t.gas4, t.gb4p, t.lossv,
                                            t.ptail, t.pooled, dbg, ext, nullptr, nullptr, true, -1, L);
  if (full) {
    // (the body ended with a barrier; nothing below reads what this workgroup stored to global memory since the chain's barrier)
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[16] = clock64();
    const int b = blockIdx.x;
    const int n0 = gn0, n = min(gn, CH_TRAIN_MAXN);
    constexpr int NBA = W1S == 8 ? 2 : 1;
#ifdef CH_REPEAT_BWD      // measurement build: the GCN backward twice (second pass: warm instruction cache; results are garbage)
    int reps_ = 2;
    DG_OPAQUE_S(reps_);
#pragma unroll 1
    for (int rep_ = 0; rep_ < reps_; ++rep_) {
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0 && rep_ == 1) { for (int k = 16; k < 22; ++k) dbg[k + 8] = dbg[k]; dbg[16] = clock64(); }
#endif
    // W2 / W3 / W4: the FORWARD's operand-order tables and its bias table, still in LDS
    ch_gcn_bwd_graph<NBA>(n0, n, F, smem, X::PS, reinterpret_cast<const unsigned*>(smem + C::OFF_BL),
                          reinterpret_cast<const float*>(smem + C::OFF_DV), reinterpret_cast<const uint2*>(smem + C::OFF_TAB),
                          reinterpret_cast<const float*>(smem + C::OFF_W3), reinterpret_cast<const float*>(smem + C::OFF_W2),
                          reinterpret_cast<const float*>(smem + C::OFF_BT) + 96, reinterpret_cast<unsigned short*>(smem + C::OFF_H4), C::ROWS,
                          reinterpret_cast<float*>(smem + X::OFF_G4T), reinterpret_cast<float*>(smem + X::OFF_GT),
                          reinterpret_cast<float*>(smem + X::OFF_SL), reinterpret_cast<float*>(smem), L, x3, x2, x1, axg,
                          t.pa4 + (size_t)b * 64, t.pb3 + (size_t)b * 1056, t.pb2 + (size_t)b * 1056, t.pb1 + (size_t)b * 32 * F, dbg);
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[21] = clock64();
#ifdef CH_REPEAT_BWD
    __syncthreads();
    }
#endif
  } else if (t.pa4) {
    __syncthreads();      // (vmcnt(0): this graph's gas4 and gp3 rows are written; the readout's LDS plan is dead)
    const int b = blockIdx.x;
    const int n0 = gn0, n = min(gn, CH_TRAIN_MAXN);
    ch_conv4_bwd_graph(n0, n, reinterpret_cast<const unsigned*>(smem + C::OFF_BL), reinterpret_cast<const float*>(smem + C::OFF_DV),
                       reinterpret_cast<const uint2*>(smem + C::OFF_TAB), reinterpret_cast<unsigned short*>(smem + C::OFF_H4), C::ROWS,
                       reinterpret_cast<float*>(smem + 65536), reinterpret_cast<float*>(smem + 65536 + 1024), t.gas4, t.W4, x3, t.gp3,
                       t.gas3, t.pa4 + (size_t)b * 64);
    for (int row = b + B; row < t.P1; row += B)       // rows of pa4 no graph owns
      if (threadIdx.x < 64) t.pa4[(size_t)row * 64 + threadIdx.x] = 0.f;
  }
}

// =================================================================================================================
// Evaluation / inference of a SMALL batch in ONE launch (round 5): the forward half of the training kernel above -- graph-chain
// forward + readout forward of a graph on the same 1024 threads -- without any backward, partial rows or gradient hand-over.
// This is the reference's test() loop body (/root/reference/train.py:57-64) and plain Model.forward (model.py:26-45).
// Rider range as in training (the next batch's graph preparation, both phases when the grid is resident).
// Metrics (train.py:63-64: loss sum and #correct of the batch), when labels AND a counter are given (the pipeline object owns
// one: dgcnn_pipeline_eval_step): wave 0 of a graph's workgroup publishes its {logp[y], argmax == y} pair by an agent-coherent
// store and adds one to the counter -- monotonic over the pipeline's life, the host passes the value it reaches with this
// launch's last graph --; the LAST one to arrive sums the B pairs in k_eval_metrics' fixed order (tail.hip: 256 threads x
// stride 256, then a binary tree) and adds them to the device accumulator -- the same additions in the same order as the
// separate launch it replaces; nobody spins, nothing is fenced, no other wave waits.  (First form: an epoch-tagged 64-bit word
// in the workspace bumped by compare-and-swap -- 50 workgroups retrying on one address made the launch 34 us whatever the graphs.)
// =================================================================================================================
struct ChEval {
  unsigned int* err; unsigned int epoch; int insym;
  int C; TailW w; float* pooled; int* perm; float *a5g, *a6g, *a1dg; uint8_t* maskg; float* logp; int training; uint64_t seed;
  const int64_t* y; float* evl; unsigned int* ctr; unsigned int target; float* metrics; float scale;      // y == null: no metrics
};
// MAXN = CH_TRAIN_MAXN: one row tile per wave, the bitmap words of the tile in registers (the form every batch of graphs of <= 256
// nodes takes).  MAXN = CH_EVAL_MAXN (512, round 6): two row tiles per wave -- the chain body of k_chain_fwd_q<16, .., LOOP = false>,
// 149 KB of LDS, no static arrays here -- so that test() on PROTEINS / DD-like batches with a graph of 257..512 nodes stays ONE
// launch too; the readout ranks such a graph's keys by radix select (dg_select_topk above 256 keys).  No bitmap-symmetry hook
// in that form (api.hip leaves the reverse-edge check to the one-launch kernels only for batches of graphs of <= 256 nodes).
template <int XI, int W1S, bool BF = false, int MAXN = CH_TRAIN_MAXN>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4)))
k_chain_readout_eval(int N, int B, int F, const int* __restrict__ graph_ptr, const unsigned* __restrict__ bits,
                     const float* __restrict__ dinv, const float* __restrict__ xs, ChW gw, float* __restrict__ axg,
                     float* __restrict__ x1, float* __restrict__ x2, float* __restrict__ x3, float* __restrict__ x4, ChEval t,
                     unsigned long long* __restrict__ dbg, DgPrepRider rd) {
  if ((int)blockIdx.x >= B) { ch_rider_block((int)blockIdx.x - B, rd); return; }
  using C = ChQ<16, W1S, MAXN>;
  if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[15] = clock64();
  const int b = blockIdx.x;
  // (label and node range read once, up front: scalar / early loads, consumed behind the chain)
  const int yraw = (t.y && threadIdx.x < 64) ? (int)t.y[b] : 0;
  const int gn0 = graph_ptr[b], gn = graph_ptr[b + 1] - gn0;
  if (threadIdx.x == 0 && gn > MAXN) { t.err[1] = t.epoch; t.err[3] = ~t.epoch; }
  DG_DYN_SMEM(char, smem);
  float* keys_lds = reinterpret_cast<float*>(smem + C::OFF_DV) + C::ROWS;      // (second parity set of the dinv array, as in training)
  ch_chain_body<16, XI, W1S, false, MAXN, BF>(N, B, F, nullptr, nullptr, graph_ptr, bits, dinv, xs, gw, axg, x1, x2, x3, x4,
                                                       nullptr, keys_lds);
  __syncthreads();        // (full barrier, vmcnt(0): this graph's x1..x4 rows are written; the LDS images are dead)
  if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[14] = clock64();
  const RdSmem M = dg_rd_carve(smem, smem + RD_REGION0_BYTES);
  const int nn_ = min(gn, MAXN);
  const ChSymHook hook{reinterpret_cast<const unsigned int*>(smem + C::OFF_BL), nn_, 1 << dgd_class(max(nn_, 1)), (nn_ + 31) >> 5,
                       (MAXN <= 256 && t.insym) ? t.err : nullptr, t.epoch, nullptr, nullptr};
  dg_readout_fwd_body(M, b, gn0, gn, t.C, t.w, keys_lds, 0, x1, x2, x3, x4, t.pooled, t.perm, t.a5g, t.a6g, t.a1dg, t.maskg, t.logp,
                      t.training, t.seed, dbg, hook);
  if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[16] = clock64();
  if (!t.y || threadIdx.x >= 64) return;      // (waves 1..15 are done; wave 0 wrote the log-probabilities (M.lg) itself: program order)
  // ---- metrics: this graph's pair, then the last workgroup's fixed-order sum -- wave 0 only, no barrier -------------------------
  const int lane = threadIdx.x;
  {
    const float v = lane < t.C ? M.lg[lane] : -INFINITY;
    float mx = v;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const unsigned long long at = __builtin_amdgcn_ballot_w64(lane < t.C && v == mx);
    const int am = at ? (int)__builtin_ctzll(at) : 0;      // first index of the maximum, as k_eval_metrics' scan
    const bool ybad = (unsigned)yraw >= (unsigned)t.C;     // out-of-range label: NaN loss (sticky), no out-of-bounds read
    const int yb = ybad ? 0 : yraw;
    const float l = ybad ? __builtin_nanf("") : M.lg[yb];
    // ONE 8-byte agent-coherent store {loss term, launch tag | correct} and, WITHOUT waiting for it, the counter: a pair that is
    // still on its way when the last workgroup looks is recognised by its stale tag and polled (the store-then-wait-then-count
    // form put a write-through round trip, ~1.3 k cycles, in front of every workgroup's counter increment)
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(t.evl) + b,
                         ((unsigned long long)((t.target << 1) | ((am == yb) ? 1u : 0u)) << 32) | __float_as_uint(l),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned int old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(t.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (unsigned int)__builtin_amdgcn_readfirstlane((int)old);
  if (dbg && blockIdx.x == 0 && lane == 0) dbg[17] = clock64();
  if (old + 1u != t.target) return;
  // last workgroup: k_eval_metrics' sums (tail.hip: virtual thread v takes graphs v, v + 256, ... -- at most one here, B <= 256 --,
  // then the binary tree 128, 64, ..., 1) with virtual threads l, l + 64, l + 128, l + 192 in lane l: the same additions
  float sl[4], sc[4];
  {
    const unsigned long long* ev = reinterpret_cast<const unsigned long long*>(t.evl);
    unsigned long long pv[4];
    unsigned int spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int g = lane + 64 * q;
        pv[q] = __hip_atomic_load(ev + min(g, B - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && (g >= B || (unsigned int)(pv[q] >> 33) == (t.target & 0x7fffffffu));
      }
      if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
      if (++spins > (1u << 20)) { if (lane == 0) { t.err[4] = t.epoch; t.err[6] = ~t.epoch; } break; }      // (never seen)
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int g = lane + 64 * q;
      sl[q] = g < B ? 0.f - __uint_as_float((unsigned int)pv[q]) : 0.f;
      sc[q] = g < B ? 0.f + (float)((unsigned int)(pv[q] >> 32) & 1u) : 0.f;
    }
  }
  sl[0] += sl[2]; sl[1] += sl[3]; sc[0] += sc[2]; sc[1] += sc[3];      // st = 128
  sl[0] += sl[1]; sc[0] += sc[1];                                      // st = 64
  for (int st = 32; st >= 1; st >>= 1) { sl[0] += __shfl_down(sl[0], st); sc[0] += __shfl_down(sc[0], st); }
  if (lane == 0) { t.metrics[0] += sl[0] * t.scale; t.metrics[1] += sc[0]; }
}

// =================================================================================================================
// BACKWARD chain, conv4 and conv3 of a graph in one workgroup (replaces k_gcn_bwd1* and the layer-3 k_gcn_bwd32*):
//   gh4[j]  = dinv[j] * sum_{i in N(j)+j} gas4[i]                       block product, ONE column (three bf16 parts as three
//                                                                       columns of one B operand, as in the forward's conv4)
//   gx3     = gh4 * W4 + gp3 ; ga3 = gx3 (1 - x3^2) ; gas3 = dinv ga3   -> LDS image (three bf16 parts), never to HBM
//   dW4    += gh4 x3 ; db3 += ga3
//   gh3[j]  = dinv[j] * sum_i gas3[i]                                   32-wide block product, evaluated in BOTH orientations
//                                                                       from the same operand registers: lane = node (for
//                                                                       gx2 and tanh') and lane = column (for dW3)
//   dW3    += gh3^T x2                                                  fp32 MFMA, contraction over the tile's 16 nodes;
//                                                                       node (step s, lane group kq) = 4 kq + s on both
//                                                                       operands: exactly what the accumulator lanes hold
//   gx2     = gh3 W3 + gp2 ; ga2 = gx2 (1 - x2^2) ; gas2 = dinv ga2     -> global (conv2's backward kernel consumes it)
//   db2    += ga2
// Same static schedule, prefetch and two-tiles-per-wave structure as the forward.  dW3 accumulates in 16 registers per wave
// across all graphs of the workgroup; the three small vectors (dW4, db3, db2) accumulate in wave-private LDS slots (row sums
// over the 16 node lanes on the DPP path, then one read-modify-write by the owning lanes: no other wave touches the slot).
// The final fixed-order sum over the waves writes ONE partial row per workgroup in the layouts k_wgrad already reduces.
// =================================================================================================================
// a row of zeros: rows WITHOUT a flag in the sparse SortPooling-gradient slabs are read from it (an unconditional load on a
// redirected address: no select behind the load -- the select was scheduled right behind its load and waited for it there,
// an exposed L2 round trip per tile and layer in the ISA)
__device__ __attribute__((aligned(128))) float ch_zero_row[32];
template <int WAVES, int MAXN>
struct ChB {
  static constexpr int THREADS = 64 * WAVES, ROWS = 32 * WAVES, KW = MAXN / 32, PS = ROWS * 32, BUF = 6 * PS;
  static constexpr int PB = (MAXN * KW + THREADS - 1) / THREADS;
  static constexpr int OFF_WT = BUF;                     // W3^T in MFMA-operand order [2][8][64] fp32
  static constexpr int OFF_W4 = OFF_WT + 4096;           // W4 [32]
  static constexpr int OFF_DV = OFF_W4 + 128;            // two sets (graph parity): dinv [ROWS]
  static constexpr int OFF_G4 = OFF_DV + 2 * 4 * ROWS;   // two sets: gas4 as three bf16 parts [3][ROWS]
  static constexpr int OFF_BL = OFF_G4 + 2 * 6 * ROWS;   // bitmap rows
  static constexpr int OFF_TAB = OFF_BL + 4 * KW * MAXN;
  static constexpr int OFF_GT = OFF_TAB + 128;           // per wave: gh4 of its two tiles [2][16]
  static constexpr int OFF_SL = OFF_GT + WAVES * 128;    // per wave: dW4 [32] | db3 [32] | db2 [32]
  static constexpr int OFF_GS = OFF_SL + WAVES * 384;    // two sets: the graph's SortPooling-gradient row flags [ROWS] (sparse slabs)
  static constexpr int TOTAL = OFF_GS + 2 * 4 * ROWS;
};

// sum over the 16 lanes of each row (lanes sharing kq): every lane of the row receives the total (fixed order)
__device__ __forceinline__ float ch_row16_sum(float v) {
  v += DG_DPP(v, 0xB1, 0xf);    // quad_perm:[1,0,3,2]
  v += DG_DPP(v, 0x4E, 0xf);    // quad_perm:[2,3,0,1]
  v += DG_DPP(v, 0x124, 0xf);   // row_ror:4
  v += DG_DPP(v, 0x128, 0xf);   // row_ror:8
  return v;
}

template <int WAVES, bool LOOP, int MAXN>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4)))
k_chain_bwd_a(int N, int B, const int* __restrict__ sched, const int* __restrict__ nbig_p, const int* __restrict__ graph_ptr,
              const unsigned* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ gas4,
              const float* __restrict__ W4, const float* __restrict__ W3, const float* __restrict__ x3,
              const float* __restrict__ gp3, const float* __restrict__ x2, const float* __restrict__ gp2,
              float* __restrict__ gas2, float* __restrict__ pa4, int P1, float* __restrict__ pb3, int P32,
              const int* __restrict__ gpsel) {
  using C = ChB<WAVES, MAXN>;
  DG_DYN_SMEM(char, smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  char* H = smem;
  float* WTop = reinterpret_cast<float*>(smem + C::OFF_WT);
  float* w4s = reinterpret_cast<float*>(smem + C::OFF_W4);
  unsigned* bl = reinterpret_cast<unsigned*>(smem + C::OFF_BL);
  uint2* tab = reinterpret_cast<uint2*>(smem + C::OFF_TAB);
  float* g4t = reinterpret_cast<float*>(smem + C::OFF_GT) + wave * 32;
  float* slot = reinterpret_cast<float*>(smem + C::OFF_SL) + wave * 96;
  const int nbig = sched ? nbig_p[0] : 0, ns = B - nbig, G = (int)gridDim.x, w = (int)blockIdx.x;
  auto entry_of = [&](int r) {
    const int li = r * G + ((r & 1) ? G - 1 - w : w), lc = min(li, max(ns - 1, 0));
    int2 e;
    if (sched) e = *reinterpret_cast<const int2*>(sched + 2 * (nbig + lc));
    else { e.x = graph_ptr[lc]; e.y = graph_ptr[lc + 1] - e.x; }
    return make_int2(e.x, (li < ns && e.y <= MAXN) ? e.y : 0);
  };
  int2 eC = entry_of(0), eN = LOOP ? entry_of(1) : make_int2(0, 0);
  unsigned pbit[C::PB]; float pdv = 0.f, pg4 = 0.f; int pgs = 1;
  // (gpsel, the row flags of the sparse SortPooling-gradient slabs, travels with the graph's prefetch and is staged in LDS: read
  //  per tile from global memory it was a round trip of its own IN FRONT of the conditional row loads it guards -- two dependent
  //  global round trips per tile and layer, the first one waited for with vmcnt(0), i.e. together with the next graph's prefetch)
  const int* gsrc = gpsel ? gpsel : graph_ptr;      // (unconditional load; the value is ignored without gpsel)
  const int gmax = gpsel ? N - 1 : 0;
  auto prefetch = [&](int pn0, int pn) {
    const int pS = 1 << dgd_class(max(pn, 1));
    const unsigned* bp = bits + (size_t)N * (pS - 1) + (size_t)pn0 * pS;
    const int last = max(pn * pS - 1, 0), lr = pn0 + min(tid, max(pn - 1, 0));
#pragma unroll
    for (int j = 0; j < C::PB; ++j) pbit[j] = bp[min(tid + C::THREADS * j, last)];
    pdv = dinv[lr]; pg4 = gas4[lr]; pgs = gsrc[min(lr, gmax)];
  };
  int n0 = __builtin_amdgcn_readfirstlane(eC.x), n = __builtin_amdgcn_readfirstlane(eC.y);
  prefetch(n0, n);
  // ---- once per workgroup: W3^T in operand order, W4, nibble table, zeroed slots ---------------------------------------------
  {   // table [kb][s][lane] = W3[o = kappa(s, lane >> 4)][k = 16 kb + (lane & 15)]   (A operand of gx = gh W3, transposed form)
    for (int e = tid; e < 1024; e += C::THREADS) {
      const int o = e >> 5, k = e & 31;                  // coalesced: element e = W3[o][k]
      const int s_ = ((o >> 4) << 2) | (o & 3), l = (k & 15) | (((o >> 2) & 3) << 4);
      WTop[(((k >> 4) * 8 + s_) << 6) + l] = W3[e];
    }
    if (tid < 32) w4s[tid] = W4[tid];
    if (tid < 16) tab[tid] = make_uint2(((tid & 1) ? 0x3f80u : 0u) | ((tid & 2) ? 0x3f800000u : 0u),
                                        ((tid & 4) ? 0x3f80u : 0u) | ((tid & 8) ? 0x3f800000u : 0u));
    for (int e = tid; e < WAVES * 96; e += C::THREADS) reinterpret_cast<float*>(smem + C::OFF_SL)[e] = 0.f;
  }
  __syncthreads();
  const int rdoff = (4 * kq + (nl >> 2)) * 32 + 8 * ((nl & 3) ^ kq);
  const int mrow0 = 16 * wave + nl, mrow1 = mrow0 + 16 * WAVES;
  const int wsl = 8 * (kq ^ ((nl >> 2) & 3));
  f32x4 accW[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  const float4 w4a = *reinterpret_cast<const float4*>(w4s + 4 * kq), w4b = *reinterpret_cast<const float4*>(w4s + 16 + 4 * kq);
  int par = 0;
  for (int r = 0; LOOP ? r * G < ns : r < 1; ++r) {
    float* dv = reinterpret_cast<float*>(smem + C::OFF_DV) + par * C::ROWS;
    unsigned short* g4p = reinterpret_cast<unsigned short*>(smem + C::OFF_G4) + par * 3 * C::ROWS;
    const int K32 = (n + 31) >> 5, T = (n + 15) >> 4, RU = 32 * K32;
    const int S = 1 << dgd_class(max(n, 1));
    int tl = tid;
    DG_OPAQUE_V(tl);      // (per-iteration copy: no address piece of the staging is hoisted out of the walk and spilled)
    // ---- stage: bitmap rows, dinv, gas4 as three bf16 parts (zeros up to RU) ------------------------------------------------
#pragma unroll
    for (int j = 0; j < C::PB; ++j)
      if (tl + C::THREADS * j < n * S) bl[tl + C::THREADS * j] = pbit[j];
    int* gsl = reinterpret_cast<int*>(smem + C::OFF_GS) + par * C::ROWS;
    if (tl < C::ROWS) {
      dv[tl] = tl < n ? pdv : 0.f;
      gsl[tl] = (!gpsel || pgs != 0) ? 1 : 0;
      if (tl < RU) {
        unsigned q0, q1, q2;
        ch_split3(tl < n ? pg4 : 0.f, q0, q1, q2);
        g4p[tl] = (unsigned short)q0; g4p[C::ROWS + tl] = (unsigned short)q1; g4p[2 * C::ROWS + tl] = (unsigned short)q2;
      }
    }
    if (RU > 16 * T && tl < 192)      // rows 16T .. RU-1 of the gas3 image: written by no tile
      *reinterpret_cast<uint4*>(H + (tl >> 5) * C::PS + (16 * T + ((tl >> 1) & 15)) * 32 + 16 * (tl & 1)) = make_uint4(0u, 0u, 0u, 0u);
    dg_lds_barrier();
    int n0N = 0, nN = 0;
    if (LOOP) {
      n0N = __builtin_amdgcn_readfirstlane(eN.x); nN = __builtin_amdgcn_readfirstlane(eN.y);
      eN = entry_of(r + 2);
      prefetch(n0N, nN);
    }
    const unsigned* blr[2] = {bl + min(mrow0, max(n - 1, 0)) * S, bl + min(mrow1, max(n - 1, 0)) * S};
    const int mrow[2] = {mrow0, mrow1};
    const bool rv[2] = {mrow0 < n, mrow1 < n};
    const float dn[2] = {dv[mrow0], dv[mrow1]};
    const bool live[2] = {wave < T, wave + WAVES < T};

    // ======== conv4 backward + start of conv3's: per tile ========
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      if (live[ti]) {
        const int m = mrow[ti];
        const bool ok = rv[ti];
        // operands of the epilogue that do not depend on the product: requested first
        const size_t ro = (size_t)(n0 + min(m, n - 1)) * 32 + 4 * kq;
        const float4 xa = *reinterpret_cast<const float4*>(x3 + ro), xb = *reinterpret_cast<const float4*>(x3 + ro + 16);
        // (gpsel: the readout backward of a large batch writes the SortPooling-gradient rows of the SELECTED nodes only and
        //  a per-node flag -- no 57 MB of zero rows written there and read back here; a row without the flag is garbage)
        // (unconditional loads: lanes of rows without the flag read the row of zeros)
        const bool sel3 = gsl[min(m, n - 1)] != 0;
        const float* gq = sel3 ? gp3 + ro : ch_zero_row + 4 * kq;
        const float4 ga_ = *reinterpret_cast<const float4*>(gq), gb_ = *reinterpret_cast<const float4*>(gq + 16);
        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
        const unsigned short* hq = g4p + min(nl, 2) * C::ROWS + 4 * kq;
#pragma unroll
        for (int u = 0; u < C::KW; ++u) {
          if (u < K32) {
            uint2 lo = *reinterpret_cast<const uint2*>(hq + 32 * u), hi = *reinterpret_cast<const uint2*>(hq + 32 * u + 16);
            if (nl >= 3) { lo = make_uint2(0u, 0u); hi = make_uint2(0u, 0u); }
            bf16x8 bop;
            unsigned* bu = reinterpret_cast<unsigned*>(&bop);
            bu[0] = lo.x; bu[1] = lo.y; bu[2] = hi.x; bu[3] = hi.y;
            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch_bits_operand(ok ? blr[ti][u] : 0u, kq, tab), bop, a4, 0, 0, 0);
          }
        }
        // lane (j = nl, kq): part j of rows 4kq + r -> gh4 of those rows through the wave's LDS tile -> lane = node
        {
          float tot[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) tot[rr] = (a4[rr] + __shfl_xor(a4[rr], 1)) + __shfl_xor(a4[rr], 2);
          if (nl == 0) {
            const float4 dq = *reinterpret_cast<const float4*>(dv + 16 * (wave + WAVES * ti) + 4 * kq);
            *reinterpret_cast<float4*>(g4t + 16 * ti + 4 * kq) = make_float4(dq.x * tot[0], dq.y * tot[1], dq.z * tot[2], dq.w * tot[3]);
          }
        }
        DG_LOCKSTEP();
        const float gh = g4t[16 * ti + nl];              // (same wave wrote it: program order + lgkmcnt)
        const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        const float gv[8] = {ga_.x, ga_.y, ga_.z, ga_.w, gb_.x, gb_.y, gb_.z, gb_.w};
        const float wv[8] = {w4a.x, w4a.y, w4a.z, w4a.w, w4b.x, w4b.y, w4b.z, w4b.w};
        float gas3[8], s4[8], s3[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float ga = ok ? fmaf(gh, wv[c], gv[c]) * (1.f - xv[c] * xv[c]) : 0.f;
          gas3[c] = dn[ti] * ga;
          s4[c] = ch_row16_sum(ok ? gh * xv[c] : 0.f);
          s3[c] = ch_row16_sum(ga);
        }
        if (nl == 0) {             // this wave's slots: cols 4kq..4kq+3 and 16+4kq..
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int col = 16 * (c >> 2) + 4 * kq + (c & 3);
            slot[col] += s4[c]; slot[32 + col] += s3[c];
          }
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {   // gas3 -> the image (three bf16 parts, row-major, slot-swizzled)
          uint2 pk[3];
          ch_split3_pack4(gas3[4 * hb], gas3[4 * hb + 1], gas3[4 * hb + 2], gas3[4 * hb + 3], pk);
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(H + (p * 2 + hb) * C::PS + m * 32 + wsl) = pk[p];
        }
      }
    }
    dg_lds_barrier();
#ifdef CH_RACE_DELAY
    if (LOOP && wave % 3 == 0) { for (int q = 0; q < 4; ++q) __builtin_amdgcn_s_sleep(127); }      // (see ch_chain_body)
#endif

    // ======== conv3 backward: per tile ========
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      if (live[ti]) {
        const int m = mrow[ti];
        const bool ok = rv[ti];
        const size_t ro = (size_t)(n0 + min(m, n - 1)) * 32 + 4 * kq;
        const float4 xa = *reinterpret_cast<const float4*>(x2 + ro), xb = *reinterpret_cast<const float4*>(x2 + ro + 16);
        const bool sel2 = gsl[min(m, n - 1)] != 0;
        const float* gq = sel2 ? gp2 + ro : ch_zero_row + 4 * kq;
        const float4 ga_ = *reinterpret_cast<const float4*>(gq), gb_ = *reinterpret_cast<const float4*>(gq + 16);
        // x2 in the lane = column layout: rows 4kq + s of column 16nb + nl (B operand of dW3)
        const int mt = 16 * (wave + WAVES * ti);
        float xN[2][4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          const int mm = mt + 4 * kq + s_;
          const float* xr = x2 + (size_t)(n0 + min(mm, n - 1)) * 32 + nl;
          const float v0 = xr[0], v1 = xr[16];
          xN[0][s_] = mm < n ? v0 : 0.f; xN[1][s_] = mm < n ? v1 : 0.f;
        }
        f32x4 accT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accN[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const char* hp = H + rdoff;
#pragma unroll
        for (int u = 0; u < C::KW; ++u) {
          if (u < K32) {
            const bf16x8 bop = ch_bits_operand(ok ? blr[ti][u] : 0u, kq, tab);
            bf16x8 a[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int nb = 0; nb < 2; ++nb) a[p][nb] = ch_read_hsT(hp + (p * 2 + nb) * C::PS + u * 1024);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int nb = 0; nb < 2; ++nb) {
                accT[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][nb], bop, accT[nb], 0, 0, 0);   // lane = node
                accN[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bop, a[p][nb], accN[nb], 0, 0, 0);   // lane = column
              }
          }
        }
        // dW3 += gh^T x2 : A[o][node] from the lane = column product (rows 4kq + s), scaled by dinv of those rows
        {
          const float4 dq = *reinterpret_cast<const float4*>(dv + mt + 4 * kq);
          const float dd[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
              for (int nb = 0; nb < 2; ++nb)
                accW[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dd[s_] * accN[mb][s_], xN[nb][s_], accW[mb][nb], 0, 0, 0);
        }
        // gx2 = gh W3 (transposed form: A = W3^T table, B = gh in the lane = node layout)
        f32x4 gx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
          const float gv_ = dn[ti] * accT[s_ >> 2][s_ & 3];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) gx[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(WTop[(kb * 8 + s_) * 64 + lane], gv_, gx[kb], 0, 0, 0);
        }
        const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        const float gpv[8] = {ga_.x, ga_.y, ga_.z, ga_.w, gb_.x, gb_.y, gb_.z, gb_.w};
        float go[8], s2[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float ga = ok ? (gx[c >> 2][c & 3] + gpv[c]) * (1.f - xv[c] * xv[c]) : 0.f;
          go[c] = dn[ti] * ga;
          s2[c] = ch_row16_sum(ga);
        }
        if (ok) {
          float* dst = gas2 + (size_t)(n0 + m) * 32 + 4 * kq;
          *reinterpret_cast<float4*>(dst) = make_float4(go[0], go[1], go[2], go[3]);
          *reinterpret_cast<float4*>(dst + 16) = make_float4(go[4], go[5], go[6], go[7]);
        }
        if (nl == 0) {
#pragma unroll
          for (int c = 0; c < 8; ++c) slot[64 + 16 * (c >> 2) + 4 * kq + (c & 3)] += s2[c];
        }
      }
    }
#ifndef CH_NO_LOOP_END_BARRIER
    if (LOOP) dg_lds_barrier();      // (conv3's product reads `bl` word by word and the loop top re-stages `bl`: see ch_chain_body's loop end; k_chain_bwd_b has the same barrier)
#endif
    n0 = n0N; n = nN; par ^= 1;
  }
  // ---- this workgroup's partial rows: the waves' accumulators combined in a fixed order ---------------------------------------
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);             // [WAVES][1024] (the image is dead now)
  {
    float* my = red + wave * 1024;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) my[(mb * 16 + kq * 4 + rr) * 32 + nb * 16 + nl] = accW[mb][nb][rr];
  }
  __syncthreads();
  const float* slots = reinterpret_cast<const float*>(smem + C::OFF_SL);
  if ((int)blockIdx.x < P32) {
    float* dst = pb3 + (size_t)blockIdx.x * 1056;
    for (int t = tid; t < 1056; t += C::THREADS) {
      float a = 0.f;
      if (t < 1024) { for (int wv_ = 0; wv_ < WAVES; ++wv_) a += red[wv_ * 1024 + t]; }
      else { for (int wv_ = 0; wv_ < WAVES; ++wv_) a += slots[wv_ * 96 + 64 + (t - 1024)]; }       // db2
      dst[t] = a;
    }
  }
  if ((int)blockIdx.x < P1 && tid < 64) {
    float a = 0.f;
    for (int wv_ = 0; wv_ < WAVES; ++wv_) a += slots[wv_ * 96 + tid];                                  // dW4 | db3
    pa4[(size_t)blockIdx.x * 64 + tid] = a;
  }
  // partial rows no workgroup owns (the reductions of k_wgrad run over P32 / P1 rows): zero
  for (int row = (int)blockIdx.x + G; row < P32; row += G)
    for (int t = tid; t < 1056; t += C::THREADS) pb3[(size_t)row * 1056 + t] = 0.f;
  for (int row = (int)blockIdx.x + G; row < P1; row += G)
    if (tid < 64) pa4[(size_t)row * 64 + tid] = 0.f;
}

// =================================================================================================================
// BACKWARD chain, second kernel: conv2's backward carrying conv1's whole weight gradient (aggregate-first conv1, F <= 32;
// replaces the layer-2 k_gcn_bwd32*<AF>).  Input gas2 [N,32] (k_chain_bwd_a), per graph:
//   gh2 = dinv * (Adj gas2)        block product in both orientations (as in k_chain_bwd_a)
//   dW2 += gh2^T x1                lane = column operands, contraction over the tile's nodes
//   gx1 = gh2 W2 + gp1             evaluated with gh2 (lane = node) as the A operand: the RESULT is lane = column, the layout
//   ga1 = gx1 (1 - x1^2)           dW1's A operand needs -- conv1 has no further backward, so nothing is needed lane = node
//   dW1 += ga1^T ax ; db1 += ga1   (ax = A_hat x saved by the forward)
// Persistent registers: dW2 (16) + dW1 (8 per 16 raw features); db1 in the wave's LDS slot.
// =================================================================================================================
template <int WAVES, bool LOOP, int MAXN, int NBA>     // NBA = 16-column blocks of the raw features (F <= 16: 1, else 2)
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4)))
k_chain_bwd_b(int N, int B, int Fa, const int* __restrict__ sched, const int* __restrict__ nbig_p, const int* __restrict__ graph_ptr,
              const unsigned* __restrict__ bits, const float* __restrict__ dinv, const float* __restrict__ gas2,
              const float* __restrict__ W2, const float* __restrict__ x1, const float* __restrict__ gp1, const float* __restrict__ axg,
              float* __restrict__ pb2, float* __restrict__ pb1, int P32, const int* __restrict__ gpsel) {
  using C = ChB<WAVES, MAXN>;
  DG_DYN_SMEM(char, smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nl = lane & 15, kq = lane >> 4;
  char* H = smem;
  float* Wop = reinterpret_cast<float*>(smem + C::OFF_WT);       // B operand of gx = gh W2: [kb][s][lane] = W2[o = kappa(s, lane>>4)][16kb + (lane&15)]
  unsigned* bl = reinterpret_cast<unsigned*>(smem + C::OFF_BL);
  uint2* tab = reinterpret_cast<uint2*>(smem + C::OFF_TAB);
  float* slot = reinterpret_cast<float*>(smem + C::OFF_SL) + wave * 96;
  const int nbig = sched ? nbig_p[0] : 0, ns = B - nbig, G = (int)gridDim.x, w = (int)blockIdx.x;
  auto entry_of = [&](int r) {
    const int li = r * G + ((r & 1) ? G - 1 - w : w), lc = min(li, max(ns - 1, 0));
    int2 e;
    if (sched) e = *reinterpret_cast<const int2*>(sched + 2 * (nbig + lc));
    else { e.x = graph_ptr[lc]; e.y = graph_ptr[lc + 1] - e.x; }
    return make_int2(e.x, (li < ns && e.y <= MAXN) ? e.y : 0);
  };
  int2 eC = entry_of(0), eN = LOOP ? entry_of(1) : make_int2(0, 0);
  unsigned pbit[C::PB]; float pdv = 0.f; int pgs = 1;
  const int* gsrc = gpsel ? gpsel : graph_ptr;      // (row flags of the sparse slabs: with the prefetch, see k_chain_bwd_a)
  const int gmax = gpsel ? N - 1 : 0;
  auto prefetch = [&](int pn0, int pn) {
    const int pS = 1 << dgd_class(max(pn, 1));
    const unsigned* bp = bits + (size_t)N * (pS - 1) + (size_t)pn0 * pS;
    const int last = max(pn * pS - 1, 0), lr = pn0 + min(tid, max(pn - 1, 0));
#pragma unroll
    for (int j = 0; j < C::PB; ++j) pbit[j] = bp[min(tid + C::THREADS * j, last)];
    pdv = dinv[lr]; pgs = gsrc[min(lr, gmax)];
  };
  int n0 = __builtin_amdgcn_readfirstlane(eC.x), n = __builtin_amdgcn_readfirstlane(eC.y);
  prefetch(n0, n);
  {
    for (int e = tid; e < 1024; e += C::THREADS) {
      const int o = e >> 5, k = e & 31;
      const int s_ = ((o >> 4) << 2) | (o & 3), l = (k & 15) | (((o >> 2) & 3) << 4);
      Wop[(((k >> 4) * 8 + s_) << 6) + l] = W2[e];
    }
    if (tid < 16) tab[tid] = make_uint2(((tid & 1) ? 0x3f80u : 0u) | ((tid & 2) ? 0x3f800000u : 0u),
                                        ((tid & 4) ? 0x3f80u : 0u) | ((tid & 8) ? 0x3f800000u : 0u));
    for (int e = tid; e < WAVES * 96; e += C::THREADS) reinterpret_cast<float*>(smem + C::OFF_SL)[e] = 0.f;
  }
  __syncthreads();
  const int rdoff = (4 * kq + (nl >> 2)) * 32 + 8 * ((nl & 3) ^ kq);
  const int mrow0 = 16 * wave + nl, mrow1 = mrow0 + 16 * WAVES;
  f32x4 accW[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  f32x4 accA[2][NBA];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBA; ++nb) accA[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  int par = 0;
  for (int r = 0; LOOP ? r * G < ns : r < 1; ++r) {
    float* dv = reinterpret_cast<float*>(smem + C::OFF_DV) + par * C::ROWS;
    const int K32 = (n + 31) >> 5, T = (n + 15) >> 4, RU = 32 * K32;
    const int S = 1 << dgd_class(max(n, 1));
    int tl = tid;
    DG_OPAQUE_V(tl);      // (per-iteration copy: no address piece of the staging is hoisted out of the walk and spilled)
    // ---- stage: bitmap rows, dinv, and the graph's gas2 rows as three bf16 parts (item = row k, 4-column slot q of 8) -------
#pragma unroll
    for (int j = 0; j < C::PB; ++j)
      if (tl + C::THREADS * j < n * S) bl[tl + C::THREADS * j] = pbit[j];
    int* gsl = reinterpret_cast<int*>(smem + C::OFF_GS) + par * C::ROWS;
    if (tl < C::ROWS) { dv[tl] = tl < n ? pdv : 0.f; gsl[tl] = (!gpsel || pgs != 0) ? 1 : 0; }
    for (int it0 = 0; it0 < RU * 8; it0 += 2 * C::THREADS) {        // two items per thread in flight
      float4 v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int it = it0 + tl + C::THREADS * j, k = it >> 3, q = it & 7;
        v[j] = *reinterpret_cast<const float4*>(gas2 + (size_t)(n0 + min(k, max(n - 1, 0))) * 32 + 4 * q);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int it = it0 + tl + C::THREADS * j, k = it >> 3, q = it & 7;
        if (it < RU * 8) {
          const bool okk = k < n;
          const float f[4] = {okk ? v[j].x : 0.f, okk ? v[j].y : 0.f, okk ? v[j].z : 0.f, okk ? v[j].w : 0.f};
          uint2 pk[3];
          ch_split3_pack4(f[0], f[1], f[2], f[3], pk);
#pragma unroll
          for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint2*>(H + (p * 2 + (q >> 2)) * C::PS + k * 32 + 8 * ((q & 3) ^ ((k >> 2) & 3))) = pk[p];
        }
      }
    }
    dg_lds_barrier();
    int n0N = 0, nN = 0;
    if (LOOP) {
      n0N = __builtin_amdgcn_readfirstlane(eN.x); nN = __builtin_amdgcn_readfirstlane(eN.y);
      eN = entry_of(r + 2);
      prefetch(n0N, nN);
    }
    const unsigned* blr[2] = {bl + min(mrow0, max(n - 1, 0)) * S, bl + min(mrow1, max(n - 1, 0)) * S};
    const bool rv[2] = {mrow0 < n, mrow1 < n};
    const float dn[2] = {dv[mrow0], dv[mrow1]};
    const bool live[2] = {wave < T, wave + WAVES < T};
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      if (live[ti]) {
        const bool ok = rv[ti];
        const int mt = 16 * (wave + WAVES * ti);
        // operands in the lane = column layout: rows 4kq + s of columns 16nb + nl
        float xN[2][4], gN[2][4], aN[NBA][4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          const int mm = mt + 4 * kq + s_;
          const bool okr = mm < n;
          const size_t ro = (size_t)(n0 + min(mm, n - 1));
          const float* xr = x1 + ro * 32 + nl;
          const bool selr = gsl[min(mm, n - 1)] != 0;
          const float* gr = (selr ? gp1 + ro * 32 : ch_zero_row) + nl;      // (rows without the flag: the row of zeros)
          const float g0 = gr[0], g1 = gr[16];
          const float x0 = xr[0], x1v = xr[16];
          xN[0][s_] = okr ? x0 : 0.f; xN[1][s_] = okr ? x1v : 0.f;
          gN[0][s_] = okr ? g0 : 0.f; gN[1][s_] = okr ? g1 : 0.f;
#pragma unroll
          for (int nb = 0; nb < NBA; ++nb) {
            const int f = 16 * nb + nl;
            // (multiplied by a 0 / 1 mask, not selected: the compiler SINKS the load of `cond ? load : 0` into the branch, where
            //  it is waited for on the spot -- four exposed L2 round trips per tile in the ISA; the clamped element is real data)
            const float av = axg[ro * Fa + min(f, Fa - 1)];
            aN[nb][s_] = av * ((okr && f < Fa) ? 1.f : 0.f);
          }
        }
        f32x4 accT[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accN[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const char* hp = H + rdoff;
#pragma unroll
        for (int u = 0; u < C::KW; ++u) {
          if (u < K32) {
            const bf16x8 bop = ch_bits_operand(ok ? blr[ti][u] : 0u, kq, tab);
            bf16x8 a[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int nb = 0; nb < 2; ++nb) a[p][nb] = ch_read_hsT(hp + (p * 2 + nb) * C::PS + u * 1024);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int nb = 0; nb < 2; ++nb) {
                accT[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p][nb], bop, accT[nb], 0, 0, 0);
                accN[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bop, a[p][nb], accN[nb], 0, 0, 0);
              }
          }
        }
        const float4 dq = *reinterpret_cast<const float4*>(dv + mt + 4 * kq);
        const float dd[4] = {dq.x, dq.y, dq.z, dq.w};
        // dW2 += gh^T x1
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              accW[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dd[s_] * accN[mb][s_], xN[nb][s_], accW[mb][nb], 0, 0, 0);
        // gx1 (lane = column): D[node][k] = sum_o gh[node][o] W2[o][k], A = gh in the lane = node layout, B = W2 table
        f32x4 gx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
          const float gv_ = dn[ti] * accT[s_ >> 2][s_ & 3];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) gx[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv_, Wop[(kb * 8 + s_) * 64 + lane], gx[kb], 0, 0, 0);
        }
        // ga1 (rows 4kq + s of column 16kb + nl), db1, dW1 += ga1^T ax
        float gaN[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          float sb = 0.f;
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) {
            gaN[kb][s_] = (gx[kb][s_] + gN[kb][s_]) * (1.f - xN[kb][s_] * xN[kb][s_]);      // (rows >= n: gx = 0, gp = 0)
            sb += gaN[kb][s_];
          }
          sb += __shfl_xor(sb, 16);
          sb += __shfl_xor(sb, 32);
          if (kq == 0) slot[16 * kb + nl] += sb;
        }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb)
              accA[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gaN[mb][s_], aN[nb][s_], accA[mb][nb], 0, 0, 0);
      }
    }
    dg_lds_barrier();                 // (the image is rewritten by the next graph's staging)
    n0 = n0N; n = nN; par ^= 1;
  }
  // ---- partial rows ------------------------------------------------------------------------------------------------------
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);             // [WAVES][1024]
  {
    float* my = red + wave * 1024;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) my[(mb * 16 + kq * 4 + rr) * 32 + nb * 16 + nl] = accW[mb][nb][rr];
  }
  __syncthreads();
  const float* slots = reinterpret_cast<const float*>(smem + C::OFF_SL);
  if ((int)blockIdx.x < P32) {
    float* dst = pb2 + (size_t)blockIdx.x * 1056;
    for (int t = tid; t < 1056; t += C::THREADS) {
      float a = 0.f;
      if (t < 1024) { for (int wv_ = 0; wv_ < WAVES; ++wv_) a += red[wv_ * 1024 + t]; }
      else { for (int wv_ = 0; wv_ < WAVES; ++wv_) a += slots[wv_ * 96 + (t - 1024)]; }              // db1
      dst[t] = a;
    }
  }
  __syncthreads();
  {
    float* my = red + wave * 1024;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) my[(mb * 16 + kq * 4 + rr) * 32 + nb * 16 + nl] = accA[mb][nb][rr];
  }
  __syncthreads();
  if ((int)blockIdx.x < P32) {
    float* d1 = pb1 + (size_t)blockIdx.x * 32 * Fa;
    for (int t = tid; t < 32 * Fa; t += C::THREADS) {
      const int o = t / Fa, f = t - o * Fa;
      float a = 0.f;
      for (int wv_ = 0; wv_ < WAVES; ++wv_) a += red[wv_ * 1024 + o * 32 + f];
      d1[t] = a;                                                                     // W1's own [32,Fa] layout
    }
  }
  for (int row = (int)blockIdx.x + G; row < P32; row += G) {
    for (int t = tid; t < 1056; t += C::THREADS) pb2[(size_t)row * 1056 + t] = 0.f;
    for (int t = tid; t < 32 * Fa; t += C::THREADS) pb1[(size_t)row * 32 * Fa + t] = 0.f;
  }
}

// ---- host launcher ----------------------------------------------------------------------------------------------------
// size classes: graphs of <= 128 nodes (8 waves, one 16-row tile each, hs ping-pongs between two LDS images: 62 KB, two
// workgroups per CU) and 129..512 nodes (16 waves x two tiles, one LDS image: 111 KB).  Each launch walks all B graphs
// and leaves the other class' graphs alone; the second launch is skipped when the host's max_nodes hint rules it out.
#ifndef CH_QW
#define CH_QW 8                          // waves per workgroup of the persistent form: 8 -> graphs of <= 256 nodes, two workgroups per CU
#endif                                   // (4 -> <= 128 nodes, four per CU: measurement build; 2048 COLLAB graphs 37 + 19 us for the
                                         // larger graphs' launch against 46 us for everything in one launch at 8)
#ifndef CH_GRID_Q
#define CH_GRID_Q (CH_QW == 8 ? 512 : 1024)
#endif
#ifndef CH_ONESHOT_MAX_B
#define CH_ONESHOT_MAX_B 256             // up to this many graphs: one 16-wave workgroup per graph (<= 512 nodes, no schedule)
#endif
#define CH_SMALL_ROWS (32 * CH_QW)
int dg_chain_max_nodes() { return 512; }
// largest graph the single-launch forms take (above it the size-class kernel runs as a second launch)
int dg_chain_small_rows(int B) { return B <= CH_ONESHOT_MAX_B ? 512 : CH_SMALL_ROWS; }
// batches of at most one graph per persistent workgroup need no schedule (and graph preparation no planning pass)
int dg_chain_needs_schedule(int B) { return B > CH_GRID_Q ? 1 : 0; }

int dg_launch_chain_fwd(int N, int B, int F, int max_nodes, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv,
                        const float* xs, const float* params, const DgParams* pl, float* ax, float* x1, float* x2, float* x3,
                        float* x4, int32_t* dmap, int bf16, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  // dmap == null: no schedule was built (dg_chain_needs_schedule(B) == 0): one graph per workgroup straight from graph_ptr
  if (N <= 0 || B <= 0 || F < 1 || F > DG_AF_MAX_F || !graph_ptr || !bits || !dinv || !xs) return DGCNN_EINVAL;
  if (!dmap && dg_chain_needs_schedule(B)) return DGCNN_EINVAL;
  ChW gw;
  gw.W1 = params + pl->off[0]; gw.b1 = params + pl->off[1]; gw.W2 = params + pl->off[2]; gw.b2 = params + pl->off[3];
  gw.W3 = params + pl->off[4]; gw.b3 = params + pl->off[5]; gw.W4 = params + pl->off[6]; gw.b4 = params + pl->off[7];
  using CL = ChCfg<16, 2, false>;
  static DgPerDeviceOnce attr_once;
  if (attr_once.needed()) {
#define CH_ATTR(W, XI, WS, LP) (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fwd_q<W, XI, WS, LP, false>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, ChQ<W, WS>::TOTAL) != hipSuccess || \
                                hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fwd_q<W, XI, WS, LP, true>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, ChQ<W, WS>::TOTAL) != hipSuccess)
    if (CH_ATTR(CH_QW, 1, 4, true) || CH_ATTR(CH_QW, 2, 4, true) || CH_ATTR(CH_QW, 4, 8, true) ||
        CH_ATTR(16, 1, 4, false) || CH_ATTR(16, 2, 4, false) || CH_ATTR(16, 4, 8, false) ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_fwd<16, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            CL::TOTAL) != hipSuccess)
      return DGCNN_ELAUNCH;
    attr_once.done();
  }
  const int* sched = dmap ? dmap + dgd_sched0(N, B) : nullptr;
  const int* nbig = dmap ? dmap + DGD_NBIG + (CH_SMALL_ROWS == 256 ? 1 : 0) : nullptr;      // graphs above the size class = first entry of the class
  // XI = xs items per thread: rows x 4-column slots with data / threads = 2^lg / 2 (at least 1)
#define CH_LQ1(W, XI, WS, LP, BFV, GRID, SCH) hipExtLaunchKernelGGL((k_chain_fwd_q<W, XI, WS, LP, BFV>), dim3(GRID), dim3(64 * W), ChQ<W, WS>::TOTAL, s, \
    ev_start, ev_stop, 0, N, B, F, SCH, nbig, graph_ptr, bits, dinv, xs, gw, ax, x1, x2, x3, x4, dg_debug_buffer())
#define CH_LQ(W, XI, WS, LP, GRID, SCH) do { if (bf16) CH_LQ1(W, XI, WS, LP, true, GRID, SCH); else CH_LQ1(W, XI, WS, LP, false, GRID, SCH); } while (0)
  if (B <= CH_ONESHOT_MAX_B) {           // one 16-wave workgroup per graph, straight from graph_ptr (graphs of <= 512 nodes)
    const int* none = nullptr;
    if (F <= 8) CH_LQ(16, 1, 4, false, B, none); else if (F <= 16) CH_LQ(16, 2, 4, false, B, none); else CH_LQ(16, 4, 8, false, B, none);
    DG_CHECK_LAUNCH();
    return DGCNN_OK;
  }
  const int grid = B < CH_GRID_Q ? B : CH_GRID_Q;
  if (F <= 8) CH_LQ(CH_QW, 1, 4, true, grid, sched); else if (F <= 16) CH_LQ(CH_QW, 2, 4, true, grid, sched); else CH_LQ(CH_QW, 4, 8, true, grid, sched);
#undef CH_LQ
#undef CH_LQ1
  DG_CHECK_LAUNCH();
  if (bf16 && (max_nodes <= 0 || max_nodes > CH_SMALL_ROWS)) return DGCNN_EUNSUPPORTED;      // (the size-class kernel has no bf16 form)
  if (max_nodes <= 0 || max_nodes > CH_SMALL_ROWS) {
    hipLaunchKernelGGL((k_chain_fwd<16, 2, false>), dim3(B), dim3(CL::THREADS), CL::TOTAL, s, N, F, graph_ptr, bits, dinv, xs, gw,
                       ax, x1, x2, x3, x4, CH_SMALL_ROWS, sched, nbig);
    DG_CHECK_LAUNCH();
  }
  return DGCNN_OK;
}

// compute units of the calling thread's device (asked once per device; the query is idempotent)
static int dg_device_cus() {
  static std::atomic<int> cus[64];
  const int dev = DgPerDeviceOnce::current();
  int ncu = cus[dev].load(std::memory_order_relaxed);
  if (ncu == 0) {
    int v = 0;
    ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 1;
    cus[dev].store(ncu, std::memory_order_relaxed);
  }
  return ncu;
}
int dg_chain_train_max_b() { return CH_ONESHOT_MAX_B; }
int dg_chain_train_max_nodes() { return CH_TRAIN_MAXN; }
int dg_chain_eval_max_nodes() { return CH_EVAL_MAXN; }

// chain forward + readout forward + readout backward of a small batch in one launch (training step with labels)
int dg_launch_chain_readout_tail(int N, int B, int F, int C, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv,
                                 const float* xs, const float* params, const DgParams* pl, float* ax, float* x1, float* x2, float* x3,
                                 float* x4, float* pooled, int32_t* perm, float* a5, float* a6, float* a1d, uint8_t* drop_mask,
                                 float* logp, int training, uint64_t seed, const int64_t* y, float loss_scale, float* dlogit, float* gz1,
                                 float* gz6, float* gz5, float* gp1, float* gp2, float* gp3, float* gas4, float* gb4p, float* lossv,
                                 float* ptail, int32_t* err, uint32_t epoch, float* gas3, float* pa4, int P1, hipStream_t s,
                                 const DgPrepRider* rider, hipEvent_t ev_start, hipEvent_t ev_stop, float* pb3, float* pb2, float* pb1,
                                 int bf16, int* fused_b_out, int insym) {
  if (fused_b_out) *fused_b_out = 0;
  if (N <= 0 || B <= 0 || B > CH_ONESHOT_MAX_B || !err || F < 1 || F > DG_AF_MAX_F || C < 1 || C > DGCNN_MAX_C || !graph_ptr || !bits ||
      !dinv || !xs || !y)
    return DGCNN_EINVAL;
  ChW gw;
  gw.W1 = params + pl->off[0]; gw.b1 = params + pl->off[1]; gw.W2 = params + pl->off[2]; gw.b2 = params + pl->off[3];
  gw.W3 = params + pl->off[4]; gw.b3 = params + pl->off[5]; gw.W4 = params + pl->off[6]; gw.b4 = params + pl->off[7];
  ChTail t;
  t.err = reinterpret_cast<unsigned int*>(err); t.epoch = epoch; t.insym = insym;
  t.W4 = gw.W4; t.gas3 = gas3; t.pa4 = (pa4 && gas3 && P1 >= B) ? pa4 : nullptr; t.P1 = P1;
  // pb3 / pb2 / pb1 (B rows each): the WHOLE GCN backward of a graph follows its conv4 backward in the same workgroup
  const bool full = t.pa4 && pb3 && pb2 && pb1;
  t.pb3 = full ? pb3 : nullptr; t.pb2 = full ? pb2 : nullptr; t.pb1 = full ? pb1 : nullptr;
  t.C = C; t.w = dg_tail_w(params, pl); t.pooled = pooled; t.perm = perm; t.a5g = a5; t.a6g = a6; t.a1dg = a1d; t.maskg = drop_mask;
  t.logp = logp; t.training = training; t.seed = seed; t.y = y; t.loss_scale = loss_scale; t.dlogit = dlogit; t.gz1g = gz1;
  t.gz6g = gz6; t.gz5g = gz5; t.gp1 = gp1; t.gp2 = gp2; t.gp3 = gp3; t.gas4 = gas4; t.gb4p = gb4p; t.lossv = lossv; t.ptail = ptail;
  DgPrepRider rd{};
  if (rider) rd = *rider;
  rd.fused_b = 0;
  // phase B of the next batch's preparation in THIS launch, behind phase A (dg_prep.h): the caller asked for it by handing in the
  // counter; taken when both phases exist for this rider
  // ... and when EVERY workgroup of the launch is resident from its start (one 1024-thread workgroup with this much LDS per CU): the
  // phase-B workgroups then wait beside running phase-A workgroups and both phases are over long before the graph workgroups are.
  // With more workgroups than CUs the riders only start when graph workgroups end, and two dependent phases at the launch's tail
  // cost more than phase B costs k_wgrad (measured at 128 / 256 graphs: 55 / 101 us per step against 44 / 56).
  const int ncu = dg_device_cus();
  const int na2 = (rd.nblk + CH_RIDER_ITEMS - 1) / CH_RIDER_ITEMS, nb2 = (rd.nblk_b + CH_RIDER_ITEMS - 1) / CH_RIDER_ITEMS;
  unsigned int sync_prev = 0; bool sync_moved = false;
  if (fused_b_out && rd.mode == 0 && rd.nblk > 0 && rd.nblk_b > 0 && rd.sync_ctr && rd.sync_host && !rd.dmap &&
      B + na2 + nb2 <= ncu - 8) {
    rd.nblk = na2;                       // (every rider thread takes CH_RIDER_ITEMS items of its phase)
    rd.fused_b = nb2;
    // the host mirror of the phase-A counter moves with the launch: rolled back below if the launch fails (it would otherwise
    // stay ahead of the device counter for the pipeline's life and every later phase-B workgroup would spin out its bound)
    sync_prev = *rd.sync_host; sync_moved = true;
    *rd.sync_host = sync_prev + (unsigned int)rd.nblk;
    rd.sync_target = *rd.sync_host;
    *fused_b_out = rd.fused_b;
  }
  static_assert(ChQ<16, 4, CH_TRAIN_MAXN>::TOTAL >= RD_REGION0_BYTES + RD_SMALL_BYTES, "the readout's LDS plan aliases the chain's images");
  static_assert(ChTrainLds<4>::TOTAL >= RD_THREADS * 8, "a rider workgroup's row buffer inside the launch's dynamic LDS");
  static DgPerDeviceOnce attr_once;
  if (attr_once.needed()) {
#define CH_ATTR2(XI, WS, BFV) (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_readout_tail<XI, WS, BFV>), \
                               hipFuncAttributeMaxDynamicSharedMemorySize, ChTrainLds<WS>::TOTAL) != hipSuccess)
    if (CH_ATTR2(1, 4, false) || CH_ATTR2(2, 4, false) || CH_ATTR2(4, 8, false) || CH_ATTR2(1, 4, true) || CH_ATTR2(2, 4, true) ||
        CH_ATTR2(4, 8, true))
      return DGCNN_ELAUNCH;
    attr_once.done();
  }
#define CH_LT(XI, WS, BFV) hipExtLaunchKernelGGL((k_chain_readout_tail<XI, WS, BFV>), dim3(B + rd.nblk + rd.fused_b), dim3(1024), ChTrainLds<WS>::TOTAL, s, \
                                                 ev_start, ev_stop, 0, N, B, F, graph_ptr, bits, dinv, xs, gw, ax, x1, x2, x3, x4, t,        \
                                                 dg_debug_buffer(), rd)
  if (bf16) { if (F <= 8) CH_LT(1, 4, true); else if (F <= 16) CH_LT(2, 4, true); else CH_LT(4, 8, true); }
  else { if (F <= 8) CH_LT(1, 4, false); else if (F <= 16) CH_LT(2, 4, false); else CH_LT(4, 8, false); }
#undef CH_LT
  if (hipGetLastError() != hipSuccess) {
    if (sync_moved) { *rd.sync_host = sync_prev; if (fused_b_out) *fused_b_out = 0; }
    return DGCNN_ELAUNCH;
  }
  return DGCNN_OK;
}

// chain forward + readout forward (+ metrics) of a small batch in one launch: evaluation / inference (round 5).  `rider` as in
// dg_launch_chain_readout_tail (*fused_b_out > 0: phase B of the rider joined the launch).  y == null: no metrics.
// evl [B][2] floats: workspace scratch of the metrics hand-over; ev_ctr / ev_host: the pipeline's device counter and its host mirror.
int dg_launch_chain_readout_eval(int N, int B, int F, int C, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv,
                                 const float* xs, const float* params, const DgParams* pl, float* ax, float* x1, float* x2, float* x3,
                                 float* x4, float* pooled, int32_t* perm, float* a5, float* a6, float* a1d, uint8_t* drop_mask,
                                 float* logp, int training, uint64_t seed, const int64_t* y, float loss_scale, float* evl,
                                 unsigned int* ev_ctr, unsigned int* ev_host, float* metrics, int32_t* err, uint32_t epoch,
                                 hipStream_t s, const DgPrepRider* rider, hipEvent_t ev_start, hipEvent_t ev_stop, int bf16,
                                 int* fused_b_out, int insym, int max_nodes) {
  if (fused_b_out) *fused_b_out = 0;
  if (N <= 0 || B <= 0 || B > CH_ONESHOT_MAX_B || !err || F < 1 || F > DG_AF_MAX_F || C < 1 || C > DGCNN_MAX_C || !graph_ptr || !bits ||
      !dinv || !xs || (y && (!evl || !ev_ctr || !ev_host || !metrics)) || max_nodes <= 0 || max_nodes > CH_EVAL_MAXN)
    return DGCNN_EINVAL;
  const bool wide = max_nodes > CH_TRAIN_MAXN;      // a graph of 257..512 nodes: the two-tiles-per-wave instantiation
  if (wide) insym = 0;
  ChW gw;
  gw.W1 = params + pl->off[0]; gw.b1 = params + pl->off[1]; gw.W2 = params + pl->off[2]; gw.b2 = params + pl->off[3];
  gw.W3 = params + pl->off[4]; gw.b3 = params + pl->off[5]; gw.W4 = params + pl->off[6]; gw.b4 = params + pl->off[7];
  ChEval t;
  t.err = reinterpret_cast<unsigned int*>(err); t.epoch = epoch; t.insym = insym;
  t.C = C; t.w = dg_tail_w(params, pl); t.pooled = pooled; t.perm = perm; t.a5g = a5; t.a6g = a6; t.a1dg = a1d; t.maskg = drop_mask;
  t.logp = logp; t.training = training; t.seed = seed; t.y = y; t.evl = evl; t.ctr = ev_ctr; t.metrics = metrics;
  t.scale = loss_scale != 0.f ? loss_scale : 1.0f / (float)B;          // (as dg_launch_eval_metrics)
  // the host mirror of the metrics counter moves with the launch (rolled back below if the launch fails)
  const unsigned int ev_prev = y ? *ev_host : 0u;
  if (y) { *ev_host = ev_prev + (unsigned int)B; t.target = *ev_host; } else t.target = 0u;
  DgPrepRider rd{};
  if (rider) rd = *rider;
  rd.fused_b = 0;
  const int ncu = dg_device_cus();
  const int na2 = (rd.nblk + CH_RIDER_ITEMS - 1) / CH_RIDER_ITEMS, nb2 = (rd.nblk_b + CH_RIDER_ITEMS - 1) / CH_RIDER_ITEMS;
  unsigned int sync_prev = 0; bool sync_moved = false;
  if (fused_b_out && rd.mode == 0 && rd.nblk > 0 && rd.nblk_b > 0 && rd.sync_ctr && rd.sync_host && !rd.dmap &&
      B + na2 + nb2 <= ncu - 8) {      // (the whole grid resident from the start: see dg_launch_chain_readout_tail)
    rd.nblk = na2;
    rd.fused_b = nb2;
    sync_prev = *rd.sync_host; sync_moved = true;
    *rd.sync_host = sync_prev + (unsigned int)rd.nblk;
    rd.sync_target = *rd.sync_host;
    *fused_b_out = rd.fused_b;
  }
  static_assert(ChQ<16, 4, CH_TRAIN_MAXN>::TOTAL >= RD_THREADS * 8, "a rider workgroup's row buffer inside the launch's dynamic LDS");
  static DgPerDeviceOnce attr_once;
  if (attr_once.needed()) {
#define CH_ATTR3(XI, WS, BFV) (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_readout_eval<XI, WS, BFV>), \
                               hipFuncAttributeMaxDynamicSharedMemorySize, ChQ<16, WS, CH_TRAIN_MAXN>::TOTAL) != hipSuccess)
#define CH_ATTR3W(XI, WS, BFV) (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_readout_eval<XI, WS, BFV, CH_EVAL_MAXN>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, ChQ<16, WS, CH_EVAL_MAXN>::TOTAL) != hipSuccess)
    if (CH_ATTR3(1, 4, false) || CH_ATTR3(2, 4, false) || CH_ATTR3(4, 8, false) || CH_ATTR3(1, 4, true) || CH_ATTR3(2, 4, true) ||
        CH_ATTR3(4, 8, true) || CH_ATTR3W(1, 4, false) || CH_ATTR3W(2, 4, false) || CH_ATTR3W(4, 8, false) || CH_ATTR3W(1, 4, true) ||
        CH_ATTR3W(2, 4, true) || CH_ATTR3W(4, 8, true))
      return DGCNN_ELAUNCH;
    attr_once.done();
  }
  static_assert(ChQ<16, 8, CH_EVAL_MAXN>::TOTAL <= 160 * 1024, "the two-tiles-per-wave form fits one CU's LDS (no static arrays in this kernel)");
  static_assert(ChQ<16, 4, CH_EVAL_MAXN>::OFF_DV >= RD_REGION0_BYTES + RD_SMALL_BYTES,
                "the readout's LDS plan stays below the key copy (second dinv set) it ranks");
#define CH_LE(XI, WS, BFV, MX) hipExtLaunchKernelGGL((k_chain_readout_eval<XI, WS, BFV, MX>), dim3(B + rd.nblk + rd.fused_b), dim3(1024),  \
                                                 (ChQ<16, WS, MX>::TOTAL), s, ev_start, ev_stop, 0, N, B, F, graph_ptr, bits, dinv, \
                                                 xs, gw, ax, x1, x2, x3, x4, t, dg_debug_buffer(), rd)
#define CH_LE2(XI, WS, BFV) do { if (wide) CH_LE(XI, WS, BFV, CH_EVAL_MAXN); else CH_LE(XI, WS, BFV, CH_TRAIN_MAXN); } while (0)
  if (bf16) { if (F <= 8) CH_LE2(1, 4, true); else if (F <= 16) CH_LE2(2, 4, true); else CH_LE2(4, 8, true); }
  else { if (F <= 8) CH_LE2(1, 4, false); else if (F <= 16) CH_LE2(2, 4, false); else CH_LE2(4, 8, false); }
#undef CH_LE2
#undef CH_LE
  if (hipGetLastError() != hipSuccess) {
    if (sync_moved) { *rd.sync_host = sync_prev; if (fused_b_out) *fused_b_out = 0; }
    if (y) *ev_host = ev_prev;
    return DGCNN_ELAUNCH;
  }
  return DGCNN_OK;
}

// ---- backward chain (conv4 + conv3) -------------------------------------------------------------------------------------
#define CH_BWD_MAXN 256
int dg_chain_bwd_max_nodes() { return CH_BWD_MAXN; }
int dg_launch_chain_bwd_a(int N, int B, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv, const float* gas4,
                          const float* W4, const float* W3, const float* x3, const float* gp3, const float* x2, const float* gp2,
                          float* gas2, float* pa4, int P1, float* pb3, int P32, int32_t* dmap, hipStream_t s, const int32_t* gpsel) {
  if (N <= 0 || B <= 0 || !graph_ptr || !bits || !dinv || !gas4 || !W4 || !W3 || !x3 || !gp3 || !x2 || !gp2 || !gas2 || !pa4 || !pb3 ||
      P1 <= 0 || P32 <= 0)
    return DGCNN_EINVAL;
  if (!dmap && dg_chain_needs_schedule(B)) return DGCNN_EINVAL;
  using CB8 = ChB<8, CH_BWD_MAXN>;
  using CB16 = ChB<16, CH_BWD_MAXN>;
  static DgPerDeviceOnce attr_once;
  if (attr_once.needed()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_bwd_a<8, true, CH_BWD_MAXN>), hipFuncAttributeMaxDynamicSharedMemorySize, CB8::TOTAL) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_bwd_a<16, false, CH_BWD_MAXN>), hipFuncAttributeMaxDynamicSharedMemorySize, CB16::TOTAL) != hipSuccess)
      return DGCNN_ELAUNCH;
    attr_once.done();
  }
  static_assert(CB8::BUF >= 8 * 4096 && CB16::BUF >= 16 * 4096, "the final reduction aliases the image");
  const int* sched = dmap ? dmap + dgd_sched0(N, B) : nullptr;
  const int* nbig = dmap ? dmap + DGD_NBIG + 1 : nullptr;      // graphs above 256 nodes come first in the schedule (none here: checked by the caller)
  // every workgroup writes ONE partial row: the grid cannot exceed the rows k_wgrad reduces (P32 / P1 = node tiles of the
  // batch, capped): batches of very small graphs (more graphs than tiles) take the looping form with fewer workgroups
  int grid = B < 512 ? B : 512;
  if (grid > P32) grid = P32;
  if (grid > P1) grid = P1;
  if (grid == B && B <= CH_ONESHOT_MAX_B) {
    hipLaunchKernelGGL((k_chain_bwd_a<16, false, CH_BWD_MAXN>), dim3(B), dim3(1024), CB16::TOTAL, s, N, B, (const int*)nullptr, (const int*)nullptr,
                       graph_ptr, bits, dinv, gas4, W4, W3, x3, gp3, x2, gp2, gas2, pa4, P1, pb3, P32, gpsel);
  } else {
    hipLaunchKernelGGL((k_chain_bwd_a<8, true, CH_BWD_MAXN>), dim3(grid), dim3(512), CB8::TOTAL, s, N, B, sched, nbig, graph_ptr, bits, dinv,
                       gas4, W4, W3, x3, gp3, x2, gp2, gas2, pa4, P1, pb3, P32, gpsel);
  }
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

int dg_launch_chain_bwd_b(int N, int B, int Fa, const int32_t* graph_ptr, const uint32_t* bits, const float* dinv, const float* gas2,
                          const float* W2, const float* x1, const float* gp1, const float* ax, float* pb2, float* pb1, int P32,
                          int32_t* dmap, hipStream_t s, const int32_t* gpsel) {
  if (N <= 0 || B <= 0 || Fa < 1 || Fa > DG_AF_MAX_F || !graph_ptr || !bits || !dinv || !gas2 || !W2 || !x1 || !gp1 || !ax || !pb2 ||
      !pb1 || P32 <= 0)
    return DGCNN_EINVAL;
  if (!dmap && dg_chain_needs_schedule(B)) return DGCNN_EINVAL;
  using CB8 = ChB<8, CH_BWD_MAXN>;
  using CB16 = ChB<16, CH_BWD_MAXN>;
  static DgPerDeviceOnce attr_once;
  if (attr_once.needed()) {
#define CH_ATTRB(W, LP, NB, TOT) (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_bwd_b<W, LP, CH_BWD_MAXN, NB>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, TOT) != hipSuccess)
    if (CH_ATTRB(8, true, 1, CB8::TOTAL) || CH_ATTRB(8, true, 2, CB8::TOTAL) || CH_ATTRB(16, false, 1, CB16::TOTAL) ||
        CH_ATTRB(16, false, 2, CB16::TOTAL))
      return DGCNN_ELAUNCH;
    attr_once.done();
  }
  const int* sched = dmap ? dmap + dgd_sched0(N, B) : nullptr;
  const int* nbig = dmap ? dmap + DGD_NBIG + 1 : nullptr;
  int grid = B < 512 ? B : 512;
  if (grid > P32) grid = P32;
#define CH_LB(W, LP, NB, GRID, TOT, SCH, NBG) hipLaunchKernelGGL((k_chain_bwd_b<W, LP, CH_BWD_MAXN, NB>), dim3(GRID), dim3(64 * W), TOT, s, N, B, Fa, \
    SCH, NBG, graph_ptr, bits, dinv, gas2, W2, x1, gp1, ax, pb2, pb1, P32, gpsel)
  const int* none = nullptr;
  if (grid == B && B <= CH_ONESHOT_MAX_B) { if (Fa <= 16) CH_LB(16, false, 1, B, CB16::TOTAL, none, none); else CH_LB(16, false, 2, B, CB16::TOTAL, none, none); }
  else { if (Fa <= 16) CH_LB(8, true, 1, grid, CB8::TOTAL, sched, nbig); else CH_LB(8, true, 2, grid, CB8::TOTAL, sched, nbig); }
#undef CH_LB
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}
