// gcn_dense.hip -- graph-convolution aggregation as DENSE PER-GRAPH BLOCK PRODUCTS on the matrix cores (gfx950).
//
// Same layer arithmetic as gcn.hip (PyG GCNConv + tanh, /root/reference/model.py:13-16,30-33):
//     out[i] = tanh( dinv[i] * sum_{j in N(i) + {i}} hs[j] + b ),      hs[j] = dinv[j] * (x W^T)[j]
// but the sum over neighbours is evaluated per graph as the product (A+I)_g . HS_g of the graph's 0/1 adjacency block
// with the graph's rows of hs.  A batch is a disjoint union of SMALL graphs (COLLAB: 75 nodes, mean degree 37, i.e.
// density 0.5): the CSR gather of gcn.hip issues one 128-B row load per EDGE through the texture path (11.6x the
// compulsory bytes, profiles/r01: TA busy 47 %, L1 hit 95 %), while the block product stages every row of hs ONCE in
// LDS and spends n_g^2/2 fp32-MFMA flops per node row -- 0.74 GFLOP for 2048 graphs, 5 us at the fp32 matrix peak.
//
// Structures (built by graph preparation, dg_prep.h): a bit-packed adjacency row per node (self bit included;
// 4*ceil(n_g/32) bytes instead of 4*deg bytes of column indices) and a work-item -> graph map.  Only for batches
// whose edge list is promised (and verified) coalesced + undirected, so the one bitmap serves forward (A) and
// backward (A^T = A), and whose graphs have at most DGD_MAXN = 512 nodes.
//
// Work item = (graph g, group of 64 rows): 4 waves, one 16-row MFMA tile per wave.  Its block product runs as a
// sequence of STAGES of 64 k-rows:
//     acc[16 x 32] += bits[16 x 64] (0/1, two bitmap words per row, expanded in registers) . HS[64 x 32] (LDS)
// on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain: the sum runs over the graph's nodes in ascending order,
// zeros included -- deterministic, independent of batch composition); 32-column words that are zero for all 16 rows
// are skipped.  Workgroups are PERSISTENT: each walks a contiguous range of items, and the stages of all its items
// form one software pipeline -- stage s+1's rows of hs and bitmap words are in flight (8 + 2 registers per thread)
// while stage s multiplies out of the other LDS buffer, one barrier per stage.  (The first, one-item-per-workgroup
// form of this kernel ran at 1.6 TB/s: each workgroup's life was a serial chain of record -> rows -> LDS -> MFMA ->
// stores with nothing overlapping it.)  Epilogue per tile, wave-private (no workgroup barrier): dst scale, bias, tanh,
// coalesced row store, next layer's X.W^T on MFMA (fp32 16x16x4, or bf16 16x16x32 for the bf16 leg), stored pre-scaled.
#include "dg_common.h"
#include "dg_prep.h"
#include "dg_readout.h"
#include <hip/hip_ext.h>

#ifndef DGD_SK
#define DGD_SK 64                        // k rows per pipeline stage (64 or 128)
#endif
#define DGD_SW (DGD_SK / 32)             // bitmap words per stage
#define DGD_KPT (DGD_SK * 32 / (64 * DGD_WAVES))   // consecutive k rows per thread in the staging (32 columns x 16 k-blocks = 512 threads)
#define DGD_HT_ROW (DGD_SK + 8)          // bf16 per column of a staged part: SK k + 8 pad (144 / 272 B: 16-B aligned, and the
                                         // 16 columns a b128 read group touches fall on distinct banks)
#define DGD_HT_PART (32 * DGD_HT_ROW)    // one part: 32 columns
#define DGD_BUF (3 * DGD_HT_PART)        // one stage buffer, in bf16 units: three parts (13.5 KiB)
#define DGD_XT 36                        // row stride (floats) of the wave-private 16x32 tiles
#ifndef DGD_WAVES
#define DGD_WAVES 8                      // one 16-row tile per wave: an item is 128 rows (DGD_ROWS, dg_prep.h).  4 waves x 64-row
#endif                                   // items (-DDGD_WAVES=4 -DDGD_ROWS=64 -DDGD_MAX_GRID=1024) measured: forward 25.6 -> 26.2,
                                         // aggregate-first conv1 32 -> 43, backward 46.5 -> 44.0, conv4 backward 27.2 -> 23.2 us;
                                         // 2048-graph step 430 -> 436 us
#define DGD_THREADS (64 * DGD_WAVES)
#ifndef DGD_MAX_GRID
#define DGD_MAX_GRID 512                 // persistent forward grid: 2 workgroups of 8 waves per CU (must divide DGD_SPLITS; 768 = 3 per
                                         // CU measured: k_gcn_fwd32d 25.0 -> 25.4 us, k_gcn_fwd_af_d 33 -> 43 us)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short dgd_f2bf(float f) {      // round to nearest even (finite inputs)
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned dgd_pack2(float a, float b) { return (unsigned)dgd_f2bf(a) | ((unsigned)dgd_f2bf(b) << 16); }

// ---- the stage stream of one workgroup ---------------------------------------------------------------------------
struct DgdRec { int n0, n, r0; };                        // item record (dg_prep.h): first node, node count, first row
struct DgdStageDesc { int n0, n, r0, c, nst; };          // stage c of nst of the item (n0, n, r0); all wave-uniform

#define DGD_REC_CACHE 128                 // item records of the workgroup held in LDS (a window, refilled when exhausted)
struct DgdGen {          // walks this workgroup's items [0, w1) of the global table grec (stride 1), skipping none
  const int* grec;       // global: record i at grec[3*i]
  int* lrec;             // LDS window [DGD_REC_CACHE][3] holding items [cbase, cbase + DGD_REC_CACHE)
  int w, w1, c, cbase;
  DgdRec cur;
  __device__ __forceinline__ void fill(int from) {       // all threads of the workgroup; uniform control flow
    cbase = from;
    __syncthreads();                                     // nobody still reads the old window
    const int cnt = min(DGD_REC_CACHE, w1 - from);
    for (int t = threadIdx.x; t < 3 * cnt; t += DGD_THREADS) lrec[t] = grec[3 * from + t];
    __syncthreads();
  }
  __device__ __forceinline__ DgdRec ld(int i) {
    DgdRec r = {0, 0, 0};
    if (i < w1) {
      if (i >= cbase + DGD_REC_CACHE) fill(i);
      const int* p = lrec + 3 * (i - cbase);
      r.n0 = __builtin_amdgcn_readfirstlane(p[0]);
      r.n = __builtin_amdgcn_readfirstlane(p[1]);
      r.r0 = __builtin_amdgcn_readfirstlane(p[2]);
    }
    return r;
  }
  __device__ __forceinline__ void init(const int* __restrict__ grec_, int* lrec_, int count) {
    grec = grec_; lrec = lrec_; w = 0; w1 = count; c = 0;
    fill(0);
    cur = ld(0);
  }
  __device__ __forceinline__ bool valid() const { return w < w1; }
  __device__ __forceinline__ DgdStageDesc get() const {
    DgdStageDesc d = {cur.n0, cur.n, cur.r0, c, (cur.n + DGD_SK - 1) / DGD_SK};
    if (w >= w1) d.n = 0;              // past the end: a descriptor whose loads are all predicated off
    return d;
  }
  __device__ __forceinline__ void advance() {
    if ((c + 1) * DGD_SK < cur.n) { ++c; return; }
    c = 0; ++w;
    cur = ld(w);
  }
};

// ---- stage loaders: 64 rows of the B operand -> registers -> LDS planes Hs[nb][k][16], plus this lane's two bitmap words
struct DgdBits {
  unsigned pend[DGD_SW], cur[DGD_SW];
  __device__ __forceinline__ void load(const DgDense& G, const DgdStageDesc& d, int wave, int lane) {
    const int m = d.r0 + wave * 16 + (lane & 15);
    const int K32 = (d.n + 31) >> 5, S = 1 << dgd_class(d.n);
    const unsigned* bp = G.bits + (size_t)G.N * (S - 1) + (size_t)(d.n0 + m) * S + DGD_SW * d.c;
    const bool ok = m < d.n;
#pragma unroll
    for (int j = 0; j < DGD_SW; ++j) pend[j] = (ok && DGD_SW * d.c + j < K32) ? bp[j] : 0u;
  }
  __device__ __forceinline__ void commit() {
#pragma unroll
    for (int j = 0; j < DGD_SW; ++j) cur[j] = pend[j];
  }
};
// The block product runs on the BF16 matrix cores and is nevertheless exact in fp32: the adjacency operand is 0/1
// (exact in bf16) and every fp32 value h is split, at staging time, into THREE bf16 parts h = h0 + h1 + h2 -- the top
// 8, middle 8 and low 8 bits of its 24-bit significand (two masks and two exact subtractions) -- so
//     A.H = A.H0 + A.H1 + A.H2     with exact products and fp32 accumulation inside v_mfma_f32_16x16x32_bf16,
// 3 MFMAs of K = 32 (~17 cycles each per SIMD) instead of 8 fp32 MFMAs of K = 4 (32 cycles each): 5x less
// matrix-pipe time for the same bits.  (bf16 leg: hs is stored in bf16, one part.)
// Staged layout Ht[part][column][k] (k contiguous): thread (column c = t & 31, k block kb = t >> 5) loads 8 consecutive
// rows of its column (a wave-instruction reads two 128-B row segments) and writes 16 B per part; the B operand of the
// MFMA (lane (n, kg): 8 consecutive k of column n) is then one ds_read_b128.
__device__ __forceinline__ void dgd_split3(float h, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned u0 = __float_as_uint(h) & 0xffff0000u;
  const float r1 = h - __uint_as_float(u0);                  // exact: <= 16 significant bits
  const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(u1);                 // exact: <= 8 significant bits, i.e. a bf16 value
  p0 = u0 >> 16; p1 = u1 >> 16; p2 = __float_as_uint(r2) >> 16;
}
// Loads are issued UNCONDITIONALLY (no exec-mask branch per row) from one base address per stage; rows at or beyond the
// graph's end are zeroed by a select, and the row offset is clamped to the slab's last row (N - 1) so that the over-read
// of up to 3 rows past the LAST graph of the batch never leaves the caller's buffer.
template <int PARTS>
__device__ __forceinline__ void dgd_store_col4(unsigned short* Ht, int c, int kb, const float (&v)[DGD_KPT]) {
  unsigned q[3][DGD_KPT];
#pragma unroll
  for (int i = 0; i < DGD_KPT; ++i) {
    if (PARTS == 3) dgd_split3(v[i], q[0][i], q[1][i], q[2][i]);
    else q[0][i] = __float_as_uint(v[i]) >> 16;               // (bf16 leg: the value IS a bf16)
  }
#pragma unroll
  for (int p = 0; p < PARTS; ++p) {
    unsigned short* dp = Ht + p * DGD_HT_PART + c * DGD_HT_ROW + DGD_KPT * kb;
    if (DGD_KPT == 8)
      *reinterpret_cast<uint4*>(dp) = make_uint4(q[p][0] | (q[p][1] << 16), q[p][2] | (q[p][3] << 16),
                                                 q[p][DGD_KPT - 4] | (q[p][DGD_KPT - 3] << 16), q[p][DGD_KPT - 2] | (q[p][DGD_KPT - 1] << 16));
    else
      *reinterpret_cast<uint2*>(dp) = make_uint2(q[p][0] | (q[p][1] << 16), q[p][2] | (q[p][3] << 16));
  }
}
struct DgdStage32 {          // hs [N,32] fp32: thread (column c = t & 31, k block kb = t >> 5 of 4 rows)
  static constexpr int PARTS = 3;
  float v[DGD_KPT];
  int N;
  __device__ __forceinline__ void load(const float* __restrict__ hs, const DgdStageDesc& d, int t) {
    const int c = t & 31, k0 = d.c * DGD_SK + DGD_KPT * (t >> 5);
    const int r0 = min(d.n0 + k0, N - 1), lim = N - 1 - r0;
    const float* bp = hs + (size_t)r0 * 32 + c;
#pragma unroll
    for (int i = 0; i < DGD_KPT; ++i) { const float x = bp[min(i, lim) * 32]; v[i] = k0 + i < d.n ? x : 0.f; }
  }
  __device__ __forceinline__ void store(unsigned short* Ht, int t) const { dgd_store_col4<3>(Ht, t & 31, t >> 5, v); }
};
struct DgdStage32bf {        // hs [N,32] bf16 (64-B rows)
  static constexpr int PARTS = 1;
  float v[DGD_KPT];
  int N;
  __device__ __forceinline__ void load(const unsigned short* __restrict__ hs, const DgdStageDesc& d, int t) {
    const int c = t & 31, k0 = d.c * DGD_SK + DGD_KPT * (t >> 5);
    const int r0 = min(d.n0 + k0, N - 1), lim = N - 1 - r0;
    const unsigned short* bp = hs + (size_t)r0 * 32 + c;
#pragma unroll
    for (int i = 0; i < DGD_KPT; ++i) { const unsigned x = bp[min(i, lim) * 32]; v[i] = k0 + i < d.n ? __uint_as_float(x << 16) : 0.f; }
  }
  __device__ __forceinline__ void store(unsigned short* Ht, int t) const { dgd_store_col4<1>(Ht, t & 31, t >> 5, v); }
};
struct DgdStageF {           // src [N,F] fp32, F <= 32 (raw features; F == 1: a scalar per node); columns >= F stay zero
  static constexpr int PARTS = 3;
  float v[DGD_KPT];
  int F;
  __device__ __forceinline__ void load(const float* __restrict__ src, const DgdStageDesc& d, int t) {
    const int c = t & 31, k0 = d.c * DGD_SK + DGD_KPT * (t >> 5);
#pragma unroll
    for (int i = 0; i < DGD_KPT; ++i) v[i] = (c < F && k0 + i < d.n) ? src[(size_t)(d.n0 + k0 + i) * F + c] : 0.f;
  }
  __device__ __forceinline__ void store(unsigned short* Ht, int t) const {
    if ((t & 31) < F) dgd_store_col4<3>(Ht, t & 31, t >> 5, v);
  }
};

// ---- the block product of one 16-row tile with 32 staged rows (one bitmap word) -------------------------------------
// A operand (16 x 32 bf16): lane (m = lane & 15, kg = lane >> 4) holds bits 8kg .. 8kg+7 of row m's word as bf16 0 / 1,
// expanded through a 16-entry nibble table in LDS (tab[nib] = four bf16).  B operand: 16 B of Ht per (part, plane).
template <int NB, int PARTS, int ROW = DGD_HT_ROW>
__device__ __forceinline__ void dgd_mma_word(unsigned w, const unsigned short* __restrict__ Ht, int krow0, int lane,
                                             const uint2* __restrict__ tab, f32x4 (&acc)[NB]) {
  constexpr int PART = 32 * ROW;
  const int kg = lane >> 4;
  const unsigned byte = (w >> (8 * kg)) & 0xffu;
  const uint2 lo = tab[byte & 15u], hi = tab[byte >> 4];
  bf16x8 a;
  unsigned* au = reinterpret_cast<unsigned*>(&a);
  au[0] = lo.x; au[1] = lo.y; au[2] = hi.x; au[3] = hi.y;
  const unsigned short* hp = Ht + (lane & 15) * ROW + krow0 + 8 * kg;
  bf16x8 b[PARTS][NB];
#pragma unroll
  for (int p = 0; p < PARTS; ++p)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      b[p][nb] = *reinterpret_cast<const bf16x8*>(hp + p * PART + nb * 16 * ROW);
#pragma unroll
  for (int p = 0; p < PARTS; ++p)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[p][nb], acc[nb], 0, 0, 0);
}

// ---- the pipeline: BODY supplies acc[NB], begin_item(desc) (zero acc, request the item's own operands) and
// end_item(desc) (the tile epilogue); both are called by every wave (they test their own tile's liveness).
// Hs: two stage buffers of DGD_BUF bf16.  Returns after the last stage of the workgroup's item range.
#ifdef DGD_TIMING       // measurement builds (tools/build_variant.sh): per-workgroup phase clocks of the forward kernel
#define DGD_T(k) do { if (dbg && lane == 0 && wave == 0) { const unsigned long long now_ = clock64(); dbg[blockIdx.x * 8 + (k)] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define DGD_T(k) do { } while (0)
#endif
template <int NB, typename STAGE, typename SRC, typename BODY>
__device__ __forceinline__ void dgd_pipeline(const DgDense& G, const SRC* __restrict__ src, unsigned short* Hs, STAGE& st,
                                             BODY& body, int lane, int wave, unsigned long long* dbg = nullptr) {
  const int t = threadIdx.x;
  __shared__ uint2 tab[16];                // nibble -> four bf16 (0 / 1.0): the A operand of the block product
  if (t < 16) tab[t] = make_uint2(((t & 1) ? 0x3f80u : 0u) | ((t & 2) ? 0x3f800000u : 0u),
                                  ((t & 4) ? 0x3f80u : 0u) | ((t & 8) ? 0x3f800000u : 0u));
#ifdef DGD_TIMING
  unsigned long long tprev_ = clock64();
  if (dbg && lane == 0 && wave == 0) { for (int k = 0; k < 8; ++k) dbg[blockIdx.x * 8 + k] = 0; dbg[blockIdx.x * 8 + 7] = tprev_; }
#endif
  __shared__ int lrec[3 * DGD_REC_CACHE];
  DgdGen gen;
  {   // this workgroup's contiguous range of equal-cost shares of the item table (dg_prep.h), laid out XCD-contiguously
      // (workgroup b runs on XCD b % 8): a graph's rows stay in one L2, and workgroups finish together
    const int Gd = (int)gridDim.x;
    const int wg = dg_xcd_tile((int)blockIdx.x, Gd);
    const int s0 = (int)(((long long)wg * DGD_SPLITS) / Gd), s1 = (int)(((long long)(wg + 1) * DGD_SPLITS) / Gd);
    const int w0 = __builtin_amdgcn_readfirstlane(G.dmap[s0]), w1 = __builtin_amdgcn_readfirstlane(G.dmap[s1]);
    gen.init(G.dmap + DGD_REC0 + 3 * w0, lrec, max(0, w1 - w0));
  }
  if (!gen.valid()) return;
  DgdBits bits;
  DgdStageDesc cur = gen.get();
  st.load(src, cur, t); bits.load(G, cur, wave, lane);
  gen.advance();
  DgdStageDesc nxt = gen.get();
  bool nv = gen.valid();
  st.store(Hs, t); bits.commit();
  // (the loads of a stage past the end are all predicated off by its n = 0: issuing them UNCONDITIONALLY lets the
  // compiler load straight into the loop-carried registers -- behind an `if` it loaded into temporaries and had to
  // wait for the data at the loop latch just to move it, i.e. right after issuing it)
  st.load(src, nxt, t); bits.load(G, nxt, wave, lane);
  int p = 0;
  DGD_T(0);                                // 0: prologue (records, first stage)
  while (true) {
    dg_lds_barrier();                      // buffer p is complete, buffer p^1 is free (LDS-only barrier: the epilogue's
                                           // global stores are NOT drained here)
    DGD_T(1);                              // 1: barrier
    if (cur.c == 0) body.begin_item(cur);
    if (cur.r0 + wave * 16 < cur.n) {
      const unsigned short* hb = Hs + p * DGD_BUF;
      // (words beyond the graph's last one were loaded as 0 for every row: the emptiness test covers them.  Fetching both
      // words' operands before the first MFMA -- 48 more registers -- was measured: block-product phase 9.8 k -> 9.0 k
      // cycles per workgroup, kernel time unchanged, one workgroup per CU less for the backward kernel; not kept)
#pragma unroll
      for (int j = 0; j < DGD_SW; ++j) {
        const unsigned w = bits.cur[j];
        if (__builtin_amdgcn_ballot_w64(w != 0u) != 0ull) dgd_mma_word<NB, STAGE::PARTS>(w, hb, 32 * j, lane, tab, body.acc);
      }
    }
    DGD_T(2);                              // 2: block product
    st.store(Hs + (p ^ 1) * DGD_BUF, t); bits.commit();             // (waits for the rows requested one stage ago)
    DGD_T(3);                              // 3: wait for the prefetched rows + LDS store
    if (nv) gen.advance();
    const DgdStageDesc nn = gen.get();
    const bool nnv = gen.valid();
    st.load(src, nn, t); bits.load(G, nn, wave, lane);
    DGD_T(4);                              // 4: issue of the next loads
    if (cur.c == cur.nst - 1) body.end_item(cur);                   // epilogue under the loads just requested
    DGD_T(5);                              // 5: epilogue
#ifdef DGD_TIMING
    if (dbg && lane == 0 && wave == 0) dbg[blockIdx.x * 8 + 6] += 1;   // 6: stages
#endif
    if (!nv) break;
    cur = nxt; nxt = nn; nv = nnv; p ^= 1;
  }
}

__device__ __forceinline__ void dgd_wave_sync() {      // orders this wave's LDS traffic (wave-private tiles: no s_barrier)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- tile epilogue shared by the 32-wide forward kernels -------------------------------------------------------------
// val[nb][r] = activated output of row kq*4+r, column nb*16 + (lane & 15) (MFMA accumulator layout), already written
// to the wave-private tile xt[16][DGD_XT].  Stores the rows coalesced, then the next layer's linear step:
//   MODE 0: hs_next[row] = dinv[row] * (x_row . Wn^T) on the matrix cores (fp32 16x16x4, or bf16 16x16x32 when BF16)
//   MODE 1: hs_next[row] = dinv[row] * (x_row . w4)   (32 -> 1)
//   MODE 2: nothing
template <int MODE, bool BF16>
__device__ __forceinline__ void dgd_tile_epilogue(float* xt, int node0, int rows_live, int lane, const float (&dpre)[4],
                                                  const float* __restrict__ dinv, float* __restrict__ xout,
                                                  const float (&wreg)[2][8], const bf16x8 (&wbf)[2], const float (&w4)[8],
                                                  void* __restrict__ hs_next) {
  dgd_wave_sync();
#pragma unroll
  for (int p = 0; p < 2; ++p) {          // 8 rows per pass, 8 lanes x 16 B = one 128-B line per row
    const int row = p * 8 + (lane >> 3), q = lane & 7;
    const float4 v = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 4 * q);
    if (row < rows_live) *reinterpret_cast<float4*>(xout + (size_t)(node0 + row) * 32 + 4 * q) = v;
  }
  if (MODE == 2) return;
  if (MODE == 1) {
    const int row = lane >> 2, seg = lane & 3;
    const float4 a = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 8 * seg);
    const float4 b = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 8 * seg + 4);
    float p = a.x * w4[0];
    p = fmaf(a.y, w4[1], p); p = fmaf(a.z, w4[2], p); p = fmaf(a.w, w4[3], p);
    p = fmaf(b.x, w4[4], p); p = fmaf(b.y, w4[5], p); p = fmaf(b.z, w4[6], p); p = fmaf(b.w, w4[7], p);
    p += __shfl_xor(p, 1);
    p += __shfl_xor(p, 2);
    if (seg == 0 && row < rows_live) reinterpret_cast<float*>(hs_next)[node0 + row] = dinv[node0 + row] * p;
    return;
  }
  f32x4 d2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (BF16) {
    // A = x tile [16 x 32] in bf16: lane (m = lane & 15, kg = lane >> 4) holds x[m][8kg .. 8kg+7]; one MFMA per 16x16 block
    const float* xr = xt + (lane & 15) * DGD_XT + 8 * (lane >> 4);
    const float4 a0 = *reinterpret_cast<const float4*>(xr), a1 = *reinterpret_cast<const float4*>(xr + 4);
    bf16x8 a;
    unsigned* au = reinterpret_cast<unsigned*>(&a);
    au[0] = dgd_pack2(a0.x, a0.y); au[1] = dgd_pack2(a0.z, a0.w); au[2] = dgd_pack2(a1.x, a1.y); au[3] = dgd_pack2(a1.z, a1.w);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) d2[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wbf[nb], d2[nb], 0, 0, 0);
  } else {
    float a[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a[kk] = xt[(lane & 15) * DGD_XT + 4 * kk + (lane >> 4)];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) d2[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], wreg[nb][kk], d2[nb], 0, 0, 0);
  }
  dgd_wave_sync();                        // every lane has read its A operands: the tile can be overwritten
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) xt[((lane >> 4) * 4 + r) * DGD_XT + nb * 16 + (lane & 15)] = dpre[r] * d2[nb][r];
  dgd_wave_sync();
  if (BF16) {                             // 64-B rows: 4 lanes x 16 B
    const int row = lane >> 2, seg = lane & 3;
    const float4 a = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 8 * seg);
    const float4 b = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 8 * seg + 4);
    const uint4 o = make_uint4(dgd_pack2(a.x, a.y), dgd_pack2(a.z, a.w), dgd_pack2(b.x, b.y), dgd_pack2(b.z, b.w));
    if (row < rows_live)
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(hs_next) + (size_t)(node0 + row) * 32 + 8 * seg) = o;
  } else {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = p * 8 + (lane >> 3), q = lane & 7;
      const float4 v = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 4 * q);
      if (row < rows_live) *reinterpret_cast<float4*>(reinterpret_cast<float*>(hs_next) + (size_t)(node0 + row) * 32 + 4 * q) = v;
    }
  }
}

// operands of the fused next-layer linear step, loaded once per workgroup lifetime
template <int MODE, bool BF16>
__device__ __forceinline__ void dgd_load_wnext(const float* __restrict__ Wn, int lane, float (&wreg)[2][8], bf16x8 (&wbf)[2],
                                               float (&w4)[8]) {
  if (MODE == 0 && !BF16) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) wreg[nb][kk] = Wn[(nb * 16 + (lane & 15)) * 32 + 4 * kk + (lane >> 4)];     // B[k][n] = Wn[n][k]
  }
  if (MODE == 0 && BF16) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float* wr = Wn + (nb * 16 + (lane & 15)) * 32 + 8 * (lane >> 4);
      const float4 a0 = *reinterpret_cast<const float4*>(wr), a1 = *reinterpret_cast<const float4*>(wr + 4);
      unsigned* u = reinterpret_cast<unsigned*>(&wbf[nb]);
      u[0] = dgd_pack2(a0.x, a0.y); u[1] = dgd_pack2(a0.z, a0.w); u[2] = dgd_pack2(a1.x, a1.y); u[3] = dgd_pack2(a1.z, a1.w);
    }
  }
  if (MODE == 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) w4[k] = Wn[8 * (lane & 3) + k];
  }
}

// =================================================================================================================
// forward, 32-wide layer (conv2, conv3; conv1 when F > 32 after its stand-alone linear)
// =================================================================================================================
template <int MODE, bool BF16>
struct DgdFwd32Body {
  f32x4 acc[2];
  float dpre[4];
  float wreg[2][8]; bf16x8 wbf[2]; float w4[8];
  float bc0, bc1;
  const float* dinv; float* xout; void* hs_next; float* xt;
  int lane, wave;
  __device__ __forceinline__ void begin_item(const DgdStageDesc& d) {
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = d.r0 + wave * 16 + (lane >> 4) * 4 + r;
      dpre[r] = m < d.n ? dinv[d.n0 + m] : 0.f;
    }
  }
  __device__ __forceinline__ void end_item(const DgdStageDesc& d) {
    const int m0 = d.r0 + wave * 16;
    if (m0 >= d.n) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (lane >> 4) * 4 + r;
      xt[row * DGD_XT + (lane & 15)] = dg_tanh(fmaf(dpre[r], acc[0][r], bc0));
      xt[row * DGD_XT + 16 + (lane & 15)] = dg_tanh(fmaf(dpre[r], acc[1][r], bc1));
    }
    dgd_tile_epilogue<MODE, BF16>(xt, d.n0 + m0, min(16, d.n - m0), lane, dpre, dinv, xout, wreg, wbf, w4, hs_next);
  }
};

template <int MODE, bool BFIN, bool BF16>     // BFIN: hs is bf16; BF16: hs_next is bf16 and X.W runs on the bf16 matrix cores
__global__ void __launch_bounds__(DGD_THREADS)
k_gcn_fwd32d(DgDense G, const float* __restrict__ dinv, const void* __restrict__ hs, const float* __restrict__ bias,
             float* __restrict__ xout, const float* __restrict__ Wn, void* __restrict__ hs_next, unsigned long long* dbg) {
  __shared__ __attribute__((aligned(16))) unsigned short Hs[2 * DGD_BUF];
  __shared__ __attribute__((aligned(16))) float xts[DGD_WAVES][16 * DGD_XT];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  DgdFwd32Body<MODE, BF16> body;
  dgd_load_wnext<MODE, BF16>(Wn, lane, body.wreg, body.wbf, body.w4);
  body.bc0 = bias[lane & 15]; body.bc1 = bias[16 + (lane & 15)];
  body.dinv = dinv; body.xout = xout; body.hs_next = hs_next; body.xt = xts[wave]; body.lane = lane; body.wave = wave;
  if (BFIN) {
    DgdStage32bf st; st.N = G.N;
    dgd_pipeline<2>(G, reinterpret_cast<const unsigned short*>(hs), Hs, st, body, lane, wave, dbg);
  } else {
    DgdStage32 st; st.N = G.N;
    dgd_pipeline<2>(G, reinterpret_cast<const float*>(hs), Hs, st, body, lane, wave, dbg);
  }
}

#ifdef DGD_WAVE_FWD32     // measurement builds only (tools/build_variant.sh wave "-DDGD_WAVE_FWD32"): a NEGATIVE result, kept
                          // for the record -- 34.4 us against the staged pipeline's 25.7 us at 2048 COLLAB-shape graphs.
                          // Wave-life clocks (tools/wave_timing.py): 16.9 k cycles per 16-row tile -- 1.7 k record fetch,
                          // 4.7 k to ISSUE the first 56 loads (84 cycles each: 16 waves per CU queue at the address unit),
                          // 2.7 k latency, 5.5 k block product (the per-tile split is ~450 VALU instructions), 2.4 k
                          // epilogue; 103 VGPRs = 4 waves per SIMD and 8-wave workgroups leave 46 % of the wave slots idle.
// ---- the same layer with INDEPENDENT WAVES (no staging, no workgroup barrier) --------------------------------------
// One workgroup per item, launched for every item (the hardware dispatcher balances them); wave w owns the 16-row tile
// r0 + 16w and leaves at once when the graph has no such rows.  Each wave fetches the B operand of its block product
// straight from global memory in the matrix-core layout -- lane (n = lane & 15, kg = lane >> 4) reads rows 8kg..8kg+7 of
// columns n and 16 + n of the 32-row k block: a wave instruction covers four 64-B row segments -- splits it into the
// three bf16 parts in registers and multiplies.  The rows of a graph are read once per tile (n/16 times, from L1/L2:
// ~100 MB of L1 traffic per launch at 2048 COLLAB-shape graphs against the 750 MB of the CSR gather), the split work
// is repeated per tile (VALU is idle anyway), and in exchange nothing ever waits for another wave: the staged
// pipeline above spends ~60 % of a workgroup's life in barriers, prologue and exposed load latency with only two
// 8-wave workgroups resident per CU and ~4 items per workgroup.
template <int PARTS>
__device__ __forceinline__ void dgw_consume(const float (&raw)[2][8], unsigned w, int lane, const uint2* __restrict__ tab,
                                            f32x4 (&acc)[2]) {
  if (__builtin_amdgcn_ballot_w64(w != 0u) == 0ull) return;
  const unsigned byte = (w >> (8 * (lane >> 4))) & 0xffu;
  const uint2 lo = tab[byte & 15u], hi = tab[byte >> 4];
  bf16x8 a;
  unsigned* au = reinterpret_cast<unsigned*>(&a);
  au[0] = lo.x; au[1] = lo.y; au[2] = hi.x; au[3] = hi.y;
  bf16x8 b[PARTS][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    unsigned q[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PARTS == 3) dgd_split3(raw[nb][i], q[0][i], q[1][i], q[2][i]);
      else q[0][i] = __float_as_uint(raw[nb][i]) >> 16;
    }
#pragma unroll
    for (int p = 0; p < PARTS; ++p) {
      unsigned* bu = reinterpret_cast<unsigned*>(&b[p][nb]);
#pragma unroll
      for (int j = 0; j < 4; ++j) bu[j] = q[p][2 * j] | (q[p][2 * j + 1] << 16);
    }
  }
#pragma unroll
  for (int p = 0; p < PARTS; ++p)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[p][nb], acc[nb], 0, 0, 0);
}

template <bool BFIN>
__device__ __forceinline__ void dgw_load(const void* __restrict__ hs, const unsigned* __restrict__ brow, int n0, int n, int kb,
                                         int K32, bool rowok, int lane, float (&raw)[2][8], unsigned& w) {
  // (every load is issued unconditionally on a clamped address: see dgd_pipeline)
  const int k0 = 32 * kb + 8 * (lane >> 4), c = lane & 15;
  const int r0 = min(k0, n - 1), lim = n - 1 - r0;
  w = brow[min(kb, K32 - 1)];
  w = (rowok && kb < K32) ? w : 0u;
  if (BFIN) {
    const unsigned short* bp = reinterpret_cast<const unsigned short*>(hs) + (size_t)(n0 + r0) * 32 + c;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int o = min(i, lim) * 32;
      const unsigned x0 = bp[o], x1 = bp[o + 16];
      raw[0][i] = k0 + i < n ? __uint_as_float(x0 << 16) : 0.f;
      raw[1][i] = k0 + i < n ? __uint_as_float(x1 << 16) : 0.f;
    }
  } else {
    const float* bp = reinterpret_cast<const float*>(hs) + (size_t)(n0 + r0) * 32 + c;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int o = min(i, lim) * 32;
      const float x0 = bp[o], x1 = bp[o + 16];
      raw[0][i] = k0 + i < n ? x0 : 0.f;
      raw[1][i] = k0 + i < n ? x1 : 0.f;
    }
  }
}

template <int MODE, bool BFIN, bool BF16>
__global__ void __launch_bounds__(DGD_THREADS)
k_gcn_fwd32w(DgDense G, const float* __restrict__ dinv, const void* __restrict__ hs, const float* __restrict__ bias,
             float* __restrict__ xout, const float* __restrict__ Wn, void* __restrict__ hs_next, unsigned long long* dbg) {
  __shared__ __attribute__((aligned(16))) float xts[DGD_WAVES][16 * DGD_XT];
  __shared__ uint2 tabs[DGD_WAVES][16];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef DGD_TIMING      // measurement builds: wave-life phase totals (tools/wave_timing.py)
  unsigned long long tw_ = clock64();
  const unsigned long long tw0_ = tw_;
#define DGW_T(k) do { if (dbg && lane == 0) { const unsigned long long now_ = clock64(); dbg[65536 + 8 * (blockIdx.x * 8 + wave) + (k)] += now_ - tw_; tw_ = now_; } } while (0)
#else
#define DGW_T(k) do { } while (0)
#endif
  if ((int)blockIdx.x >= G.dmap[DGD_SPLITS + 1]) { DGW_T(6); return; }
  const int* rec = G.dmap + DGD_REC0 + 3 * blockIdx.x;
  const int n0 = rec[0], n = rec[1], m0 = rec[2] + wave * 16;
  if (m0 >= n) { DGW_T(5); return; }
  DGW_T(0);                              // 0: record
  uint2* tab = tabs[wave];
  if (lane < 16) tab[lane] = make_uint2(((lane & 1) ? 0x3f80u : 0u) | ((lane & 2) ? 0x3f800000u : 0u),
                                        ((lane & 4) ? 0x3f80u : 0u) | ((lane & 8) ? 0x3f800000u : 0u));
  const int K32 = (n + 31) >> 5, S = 1 << dgd_class(n);
  const int mrow = m0 + (lane & 15);
  const bool rowok = mrow < n;
  const unsigned* brow = G.bits + (size_t)G.N * (S - 1) + (size_t)(n0 + min(mrow, n - 1)) * S;
  float ra[2][8], rb[2][8];
  unsigned wa, wb;
  dgw_load<BFIN>(hs, brow, n0, n, 0, K32, rowok, lane, ra, wa);
  dgw_load<BFIN>(hs, brow, n0, n, 1, K32, rowok, lane, rb, wb);
  // operands of the epilogue, requested behind the first rows
  float wreg[2][8]; bf16x8 wbf[2]; float w4[8];
  dgd_load_wnext<MODE, BF16>(Wn, lane, wreg, wbf, w4);
  const float bc0 = bias[lane & 15], bc1 = bias[16 + (lane & 15)];
  float dpre[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + (lane >> 4) * 4 + r;
    dpre[r] = m < n ? dinv[n0 + m] : 0.f;
  }
  dgd_wave_sync();                       // the table
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  constexpr int PARTS = BFIN ? 1 : 3;
  DGW_T(1);                              // 1: issue of the first loads
#ifdef DGD_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  DGW_T(2);                              // 2: their latency
#endif
  for (int kb = 0; ; kb += 2) {
    dgw_consume<PARTS>(ra, wa, lane, tab, acc);
    if (kb + 1 >= K32) break;
    dgw_load<BFIN>(hs, brow, n0, n, kb + 2, K32, rowok, lane, ra, wa);
    dgw_consume<PARTS>(rb, wb, lane, tab, acc);
    if (kb + 2 >= K32) break;
    dgw_load<BFIN>(hs, brow, n0, n, kb + 3, K32, rowok, lane, rb, wb);
  }
  DGW_T(3);                              // 3: block product (+ the later loads)
  float* xt = xts[wave];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = (lane >> 4) * 4 + r;
    xt[row * DGD_XT + (lane & 15)] = dg_tanh(fmaf(dpre[r], acc[0][r], bc0));
    xt[row * DGD_XT + 16 + (lane & 15)] = dg_tanh(fmaf(dpre[r], acc[1][r], bc1));
  }
  dgd_tile_epilogue<MODE, BF16>(xt, n0 + m0, min(16, n - m0), lane, dpre, dinv, xout, wreg, wbf, w4, hs_next);
  DGW_T(4);                              // 4: epilogue
#ifdef DGD_TIMING
  if (dbg && lane == 0) dbg[65536 + 8 * (blockIdx.x * 8 + wave) + 7] += 1;
#endif
}

#endif   // DGD_WAVE_FWD32

// =================================================================================================================
// forward of conv1, aggregate-first (raw feature width F <= 32):  ax = A_hat x (saved), x1 = tanh(ax W1^T + b1),
// hs2 = dinv * (x1 W2^T).  xs = dinv * x [N,F] comes from graph preparation.  NB = number of 16-column planes (F > 16: 2).
// =================================================================================================================
template <int NB, bool BF16>
struct DgdAfBody {
  f32x4 acc[NB];
  float dpre[4];
  float wreg[2][8]; bf16x8 wbf[2]; float w4[8];
  float w1r[2][NB * 4];                  // B operand of ax . W1^T : B[k][n] = W1[n][k], K = F padded to a multiple of 4
  float bc0, bc1;
  const float* dinv; float* xout; float* axout; void* hs_next; float* xt;
  int lane, wave, F, F4;
  __device__ __forceinline__ void begin_item(const DgdStageDesc& d) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = d.r0 + wave * 16 + (lane >> 4) * 4 + r;
      dpre[r] = m < d.n ? dinv[d.n0 + m] : 0.f;
    }
  }
  __device__ __forceinline__ void end_item(const DgdStageDesc& d) {
    const int m0 = d.r0 + wave * 16;
    if (m0 >= d.n) return;
    // ax tile (columns >= F are exact zeros: their B columns are zero) -> LDS + the saved slab
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) * 4 + r, col = nb * 16 + (lane & 15);
        const float ax = dpre[r] * acc[nb][r];
        xt[row * DGD_XT + col] = ax;
        if (col < F && m0 + row < d.n) axout[(size_t)(d.n0 + m0 + row) * F + col] = ax;
      }
    dgd_wave_sync();
    float a[NB * 4];
#pragma unroll
    for (int kk = 0; kk < NB * 4; ++kk) a[kk] = kk < F4 ? xt[(lane & 15) * DGD_XT + 4 * kk + (lane >> 4)] : 0.f;
    f32x4 d1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < NB * 4; ++kk)
      if (kk < F4) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) d1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], w1r[nb][kk], d1[nb], 0, 0, 0);
      }
    dgd_wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (lane >> 4) * 4 + r;
      xt[row * DGD_XT + (lane & 15)] = dg_tanh(d1[0][r] + bc0);
      xt[row * DGD_XT + 16 + (lane & 15)] = dg_tanh(d1[1][r] + bc1);
    }
    dgd_tile_epilogue<0, BF16>(xt, d.n0 + m0, min(16, d.n - m0), lane, dpre, dinv, xout, wreg, wbf, w4, hs_next);
  }
};

template <int NB, bool BF16>
__global__ void __launch_bounds__(DGD_THREADS)
k_gcn_fwd_af_d(DgDense G, int F, const float* __restrict__ dinv, const float* __restrict__ xs, const float* __restrict__ W1,
               const float* __restrict__ bias, float* __restrict__ axout, float* __restrict__ xout,
               const float* __restrict__ Wn, void* __restrict__ hs_next) {
  __shared__ __attribute__((aligned(16))) unsigned short Hs[2 * DGD_BUF];
  __shared__ __attribute__((aligned(16))) float xts[DGD_WAVES][16 * DGD_XT];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  DgdAfBody<NB, BF16> body;
  dgd_load_wnext<0, BF16>(Wn, lane, body.wreg, body.wbf, body.w4);
  body.bc0 = bias[lane & 15]; body.bc1 = bias[16 + (lane & 15)];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int kk = 0; kk < NB * 4; ++kk) {
      const int k = 4 * kk + (lane >> 4);
      body.w1r[nb][kk] = k < F ? W1[(nb * 16 + (lane & 15)) * F + k] : 0.f;
    }
  body.dinv = dinv; body.xout = xout; body.axout = axout; body.hs_next = hs_next; body.xt = xts[wave];
  body.lane = lane; body.wave = wave; body.F = F; body.F4 = (F + 3) >> 2;
  // columns >= F of the staged parts are never written: clear them once (0 * garbage could be NaN)
  for (int t = threadIdx.x; t < DGD_BUF; t += DGD_THREADS) reinterpret_cast<unsigned*>(Hs)[t] = 0u;
  __syncthreads();
  DgdStageF st; st.F = F;
  dgd_pipeline<NB>(G, xs, Hs, st, body, lane, wave);
}

// =================================================================================================================
// forward of conv4 (32 -> 1): x4[i] = tanh( dinv[i] * sum_{j in N(i)+{i}} h4s[j] + b )
// =================================================================================================================
struct DgdFwd1Body {
  f32x4 acc[1];
  float dpre[4];
  float b;
  const float* dinv; float* x4;
  int lane, wave;
  __device__ __forceinline__ void begin_item(const DgdStageDesc& d) {
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = d.r0 + wave * 16 + (lane >> 4) * 4 + r;
      dpre[r] = m < d.n ? dinv[d.n0 + m] : 0.f;
    }
  }
  __device__ __forceinline__ void end_item(const DgdStageDesc& d) {
    if ((lane & 15) != 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = d.r0 + wave * 16 + (lane >> 4) * 4 + r;
      if (m < d.n) x4[d.n0 + m] = dg_tanh(fmaf(dpre[r], acc[0][r], b));
    }
  }
};

__global__ void __launch_bounds__(DGD_THREADS)
k_gcn_fwd1d(DgDense G, const float* __restrict__ dinv, const float* __restrict__ h4s, const float* __restrict__ bias,
            float* __restrict__ x4) {
  __shared__ __attribute__((aligned(16))) unsigned short Hs[2 * DGD_BUF];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  DgdFwd1Body body;
  body.b = bias[0]; body.dinv = dinv; body.x4 = x4; body.lane = lane; body.wave = wave;
  for (int t = threadIdx.x; t < DGD_BUF; t += DGD_THREADS) reinterpret_cast<unsigned*>(Hs)[t] = 0u;
  __syncthreads();
  DgdStageF st; st.F = 1;
  dgd_pipeline<1>(G, h4s, Hs, st, body, lane, wave);
}

// =================================================================================================================
// backward of a 32-wide layer l (l = 3, 2), dense block form of k_gcn_bwd32 (gcn.hip): with gas_l = dinv * dL/d(pre-act_l)
//   gh[j]      = dinv[j] * sum_{i in N(j)+{j}} gas_l[i]          (block product, A^T = A)
//   dW_l      += gh^T . x_{l-1}                                   (MFMA, K = the tile's 16 nodes; per-wave accumulators)
//   gx_{l-1}   = gh . W_l + gp_{l-1}                              (MFMA)
//   ga_{l-1}   = gx_{l-1} * (1 - x_{l-1}^2) ; gas_{l-1} = dinv * ga_{l-1} ; db_{l-1} += ga_{l-1}
//   AF (layer 2 when conv1 ran aggregate-first): dW_1 += ga_1^T . ax  from the saved A_hat X slab, no gas_1 output
// Grid = the P32 partial slots of the workspace; every workgroup writes one partial row {dW_l [32x32], db_{l-1} [32]}
// (+ dW_1 [32 x Fa]), reduced later in a fixed order.
// =================================================================================================================
#define DGD_BW_HF DGD_BUF                 // the two stage buffers (2 * DGD_BUF bf16) measured in floats
#define DGD_BW_SMEM (DGD_BW_HF + DGD_WAVES * 2 * 16 * DGD_XT)      // stage buffers + per wave {ght, xt} tiles

template <bool AF>
struct DgdBwd32Body {
  f32x4 acc[2];
  float dpre[4], gpp[2][4];
  float4 xr[2];
  const float* Wlds;                       // W_l [32][33] in LDS: B operand of gx = gh . W_l : B[k][n] = W_l[k][nb*16+n]
  f32x4 accW[2][2], accA[2][2];
  float pb[2];
  const float *dinv, *xprev, *gpprev, *axin;
  float* gas_prev;
  float *ght, *xt, *aux;
  int lane, wave, Fa, nbA;
  __device__ __forceinline__ void begin_item(const DgdStageDesc& d) {
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m0 = d.r0 + wave * 16, kq = lane >> 4, nl = lane & 15;
    // everything of the epilogue that does not depend on the block product is requested now
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + kq * 4 + r;
      const bool ok = m < d.n;
      dpre[r] = ok ? dinv[d.n0 + m] : 0.f;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) gpp[nb][r] = ok ? gpprev[(size_t)(d.n0 + m) * 32 + nb * 16 + nl] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int m = m0 + p * 8 + (lane >> 3);
      xr[p] = m < d.n ? *reinterpret_cast<const float4*>(xprev + (size_t)(d.n0 + m) * 32 + 4 * (lane & 7))
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void end_item(const DgdStageDesc& d) {
    const int m0 = d.r0 + wave * 16, kq = lane >> 4, nl = lane & 15;
    if (m0 >= d.n) return;
    float axv[AF ? 8 : 1];
    if (AF) {      // (requested here, not at the item's start: 8 registers less over the block product -> 2 workgroups per CU)
#pragma unroll
      for (int u = 0; u < 8; ++u) {          // ax tile [16][Fa]: element idx = lane + 64u -> (row idx / 32, col idx % 32)
        const int idx = lane + 64 * u, row = idx >> 5, col = idx & 31;
        axv[u] = (col < Fa && m0 + row < d.n) ? axin[(size_t)(d.n0 + m0 + row) * Fa + col] : 0.f;
      }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ght[(kq * 4 + r) * DGD_XT + nb * 16 + nl] = dpre[r] * acc[nb][r];
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<float4*>(xt + (p * 8 + (lane >> 3)) * DGD_XT + 4 * (lane & 7)) = xr[p];
    dgd_wave_sync();
    // gx = gh . W_l
    f32x4 gx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    {
      float a[8], wv[2][8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        a[kk] = ght[nl * DGD_XT + 4 * kk + kq];
        wv[0][kk] = Wlds[(4 * kk + kq) * 33 + nl]; wv[1][kk] = Wlds[(4 * kk + kq) * 33 + 16 + nl];
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) gx[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], wv[nb][kk], gx[nb], 0, 0, 0);
    }
    // dW_l += gh^T . x_prev : A[m][k] = ght[k][mb*16+m], B[k][n] = xt[k][nb*16+n], K = 16 nodes
    {
      float a[2][4], b[2][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          a[h][kk] = ght[(4 * kk + kq) * DGD_XT + h * 16 + nl];
          b[h][kk] = xt[(4 * kk + kq) * DGD_XT + h * 16 + nl];
        }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            accW[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][kk], b[nb][kk], accW[mb][nb], 0, 0, 0);
    }
    // tanh' of the previous layer, its bias gradient, and the propagated gradient
    float ga[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xv = xt[(kq * 4 + r) * DGD_XT + nb * 16 + nl];
        ga[nb][r] = (gx[nb][r] + gpp[nb][r]) * (1.f - xv * xv);
        pb[nb] += ga[nb][r];
      }
    dgd_wave_sync();                      // all reads of ght done: it now carries ga (AF) or dinv*ga (the output rows)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ght[(kq * 4 + r) * DGD_XT + nb * 16 + nl] = AF ? ga[nb][r] : dpre[r] * ga[nb][r];
    if (AF) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int idx = lane + 64 * u; aux[(idx >> 5) * DGD_XT + (idx & 31)] = axv[u]; }
    }
    dgd_wave_sync();
    if (AF) {
      // dW_1 += ga_1^T . ax : A[m][k] = gat[k][mb*16+m], B[k][n] = ax[k][nq*16+n]
      float a[2][4], b[2][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          a[h][kk] = ght[(4 * kk + kq) * DGD_XT + h * 16 + nl];
          b[h][kk] = aux[(4 * kk + kq) * DGD_XT + h * 16 + nl];
        }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nq = 0; nq < 2; ++nq)
            if (nq < nbA) accA[mb][nq] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][kk], b[nq][kk], accA[mb][nq], 0, 0, 0);
    } else {
      const int rows_live = min(16, d.n - m0);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = p * 8 + (lane >> 3), q = lane & 7;
        const float4 v = *reinterpret_cast<const float4*>(ght + row * DGD_XT + 4 * q);
        if (row < rows_live) *reinterpret_cast<float4*>(gas_prev + (size_t)(d.n0 + m0 + row) * 32 + 4 * q) = v;
      }
    }
    dgd_wave_sync();                      // the tiles are rewritten by this wave's next item
  }
};

template <bool AF>
__global__ void __launch_bounds__(DGD_THREADS) __attribute__((amdgpu_waves_per_eu(AF ? 2 : 4)))      // plain form: <= 128 registers, 2 workgroups per CU (forcing the AF form into 128 spills 35 registers: 51 -> ~95 us)
k_gcn_bwd32d(DgDense G, const float* __restrict__ dinv, const float* __restrict__ gas, const float* __restrict__ Wl,
             const float* __restrict__ xprev, const float* __restrict__ gpprev, float* __restrict__ gas_prev,
             float* __restrict__ part, const float* __restrict__ axin, int Fa, float* __restrict__ part1) {
  __shared__ __attribute__((aligned(16))) float smem[DGD_BW_SMEM];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kq = lane >> 4, nl = lane & 15;
  DgdBwd32Body<AF> body;
  body.ght = smem + DGD_BW_HF + wave * 2 * 16 * DGD_XT;
  body.xt = body.ght + 16 * DGD_XT;
  body.aux = body.xt;                         // AF: the ax tile [16][<=32] replaces x_prev once tanh' has read it
  __shared__ float Wlds[32 * 33];         // (registers: 16 fewer per lane -> two workgroups per CU)
  for (int t = threadIdx.x; t < 1024; t += DGD_THREADS) Wlds[(t >> 5) * 33 + (t & 31)] = Wl[t];
  body.Wlds = Wlds;
  __syncthreads();
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) { body.accW[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; body.accA[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  body.pb[0] = 0.f; body.pb[1] = 0.f;
  body.dinv = dinv; body.xprev = xprev; body.gpprev = gpprev; body.axin = axin; body.gas_prev = gas_prev;
  body.lane = lane; body.wave = wave; body.Fa = Fa; body.nbA = AF ? ((Fa + 15) >> 4) : 0;
  DgdStage32 st; st.N = G.N;
  dgd_pipeline<2>(G, gas, reinterpret_cast<unsigned short*>(smem), st, body, lane, wave);

  // ---- this workgroup's partial row: the four waves' accumulators combined in a fixed order -----------------------
  __syncthreads();
  float* red = smem;                         // [DGD_WAVES][1056] (the stage buffers and tiles are dead now)
  {
    float* my = red + wave * 1056;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) my[(mb * 16 + kq * 4 + r) * 32 + nb * 16 + nl] = body.accW[mb][nb][r];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float p = body.pb[nb];
      p += __shfl_xor(p, 16);
      p += __shfl_xor(p, 32);
      if (lane < 16) my[1024 + nb * 16 + lane] = p;
    }
  }
  __syncthreads();
  float* dst = part + (size_t)blockIdx.x * 1056;
  for (int t = threadIdx.x; t < 1056; t += DGD_THREADS) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < DGD_WAVES; ++w) a += red[w * 1056 + t];        // fixed order
    dst[t] = a;
  }
  if (AF) {
    __syncthreads();
    float* my = red + wave * 1024;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int r = 0; r < 4; ++r) my[(mb * 16 + kq * 4 + r) * 32 + nq * 16 + nl] = body.accA[mb][nq][r];
    __syncthreads();
    float* d1 = part1 + (size_t)blockIdx.x * 32 * Fa;
    for (int t = threadIdx.x; t < 32 * Fa; t += DGD_THREADS) {
      const int c = t / Fa, k = t - c * Fa;
      const int o = c * 32 + k;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < DGD_WAVES; ++w) a += red[w * 1024 + o];
      d1[t] = a;                                                                     // W1's own [32,Fa] layout
    }
  }
}

// =================================================================================================================
// backward of conv4 (F_out = 1) fused with the start of conv3's backward, dense block form of k_gcn_bwd1 (gcn.hip):
//   gh4[j] = dinv[j] * sum_{i in N(j)+{j}} gas4[i] ;  gx3 = gh4 * W4 + gp3 ;  ga3 = gx3 * (1 - x3^2) ; gas3 = dinv * ga3
//   partials: dW4 += gh4 * x3 (32), db3 += ga3 (32)    ->  pa4[P1][64]
// =================================================================================================================
struct DgdBwd1Body {
  f32x4 acc[1];
  float dpre[4];
  float4 xr[2], gr[2], w4q, pW, pB;
  float dj[2];
  const float *dinv, *x3, *gp3;
  float* gas3; float* gh4s;
  int lane, wave;
  __device__ __forceinline__ void begin_item(const DgdStageDesc& d) {
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m0 = d.r0 + wave * 16, q = lane & 7;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + (lane >> 4) * 4 + r;
      dpre[r] = m < d.n ? dinv[d.n0 + m] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int m = m0 + p * 8 + (lane >> 3);
      const bool ok = m < d.n;
      xr[p] = ok ? *reinterpret_cast<const float4*>(x3 + (size_t)(d.n0 + m) * 32 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      gr[p] = ok ? *reinterpret_cast<const float4*>(gp3 + (size_t)(d.n0 + m) * 32 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      dj[p] = ok ? dinv[d.n0 + m] : 0.f;
    }
  }
  __device__ __forceinline__ void end_item(const DgdStageDesc& d) {
    const int m0 = d.r0 + wave * 16, q = lane & 7;
    if (m0 >= d.n) return;
    if ((lane & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gh4s[(lane >> 4) * 4 + r] = dpre[r] * acc[0][r];
    }
    dgd_wave_sync();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = p * 8 + (lane >> 3), m = m0 + row;
      if (m < d.n) {
        const float gh = gh4s[row];
        float4 ga;
        ga.x = fmaf(gh, w4q.x, gr[p].x) * (1.f - xr[p].x * xr[p].x);
        ga.y = fmaf(gh, w4q.y, gr[p].y) * (1.f - xr[p].y * xr[p].y);
        ga.z = fmaf(gh, w4q.z, gr[p].z) * (1.f - xr[p].z * xr[p].z);
        ga.w = fmaf(gh, w4q.w, gr[p].w) * (1.f - xr[p].w * xr[p].w);
        *reinterpret_cast<float4*>(gas3 + (size_t)(d.n0 + m) * 32 + 4 * q) =
            make_float4(dj[p] * ga.x, dj[p] * ga.y, dj[p] * ga.z, dj[p] * ga.w);
        pW.x = fmaf(gh, xr[p].x, pW.x); pW.y = fmaf(gh, xr[p].y, pW.y);
        pW.z = fmaf(gh, xr[p].z, pW.z); pW.w = fmaf(gh, xr[p].w, pW.w);
        pB.x += ga.x; pB.y += ga.y; pB.z += ga.z; pB.w += ga.w;
      }
    }
    dgd_wave_sync();
  }
};

__global__ void __launch_bounds__(DGD_THREADS)
k_gcn_bwd1d(DgDense G, const float* __restrict__ dinv, const float* __restrict__ gas4, const float* __restrict__ W4,
            const float* __restrict__ x3, const float* __restrict__ gp3, float* __restrict__ gas3,
            float* __restrict__ pa4) {
  __shared__ __attribute__((aligned(16))) unsigned short Hs[2 * DGD_BUF];
  __shared__ float gh4s[DGD_WAVES][16];
  __shared__ float red[DGD_WAVES][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  DgdBwd1Body body;
  body.w4q = *reinterpret_cast<const float4*>(W4 + 4 * (lane & 7));
  body.pW = make_float4(0.f, 0.f, 0.f, 0.f); body.pB = make_float4(0.f, 0.f, 0.f, 0.f);
  body.dinv = dinv; body.x3 = x3; body.gp3 = gp3; body.gas3 = gas3; body.gh4s = gh4s[wave];
  body.lane = lane; body.wave = wave;
  for (int t = threadIdx.x; t < DGD_BUF; t += DGD_THREADS) reinterpret_cast<unsigned*>(Hs)[t] = 0u;
  __syncthreads();
  DgdStageF st; st.F = 1;
  dgd_pipeline<1>(G, gas4, Hs, st, body, lane, wave);
  // lanes with equal q (8 row groups) -> wave totals -> fixed-order sum over the 4 waves
  float4 pW = body.pW, pB = body.pB;
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    pW = dg_add4(pW, dg_shfl_xor4(pW, o));
    pB = dg_add4(pB, dg_shfl_xor4(pB, o));
  }
  if (lane < 8) {
    *reinterpret_cast<float4*>(&red[wave][4 * lane]) = pW;
    *reinterpret_cast<float4*>(&red[wave][32 + 4 * lane]) = pB;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < DGD_WAVES; ++w) a += red[w][threadIdx.x];        // fixed order
    pa4[(size_t)blockIdx.x * 64 + threadIdx.x] = a;
  }
}


// =================================================================================================================
// FUSED forward, dense block form: ONE workgroup per graph runs conv1..conv4 and the SortPooling readout + dense tail
// (/root/reference/model.py:26-45) in a single launch -- for small batches of small graphs (the reference's batch of 50),
// where the per-layer launches are bound by ~5 us of dispatch + cold-read latency each, not by work.
//   * the graph's adjacency bitmap is built in LDS from its CSR slice (LDS integer atomics; no graph-preparation change)
//   * hs of the current layer lives in LDS as three bf16 parts, k-contiguous (Ht[part][col][k]); every layer is
//     acc = bits . Ht on v_mfma_f32_16x16x32_bf16 (exact, see dgd_split3), one 16-row tile per wave, then the tile epilogue
//     (dst scale, bias, tanh, x_l row store for backward / SortPooling) and the next layer's X.W^T on the fp32 matrix
//     cores, whose output is split and written STRAIGHT into the other Ht buffer (each lane owns 4 consecutive k of its
//     column: one 8-byte LDS store per part) -- between layers there is one barrier and no global traffic
//   * conv4 (32 -> 1) is the same product with a single column; its output feeds the LDS sort of dg_readout.h directly
// Limits: n_g <= FD_NMAX (192: 12 tiles), F <= 32 (aggregate-first conv1).  Same sums as the per-layer dense kernels.
// =================================================================================================================
#define FD_NMAX 192
#define FD_THREADS 1024
#define FD_ROW (FD_NMAX + 8)             // bf16 per staged column: 400 B (16-B aligned; 100 dwords = 36 mod 64: the 16
                                         // columns of a b128 read group fall on distinct banks)
#define FD_PART (32 * FD_ROW)
#define FD_HT (3 * FD_PART)              // one hs buffer (bf16 units): 37.5 KiB
#define FD_KW (FD_NMAX / 32)             // bitmap words per row
#define FD_WS 33                         // row stride of the transposed weight tiles in LDS

__device__ __forceinline__ void fd_store_hs4(unsigned short* Ht, int col, int row0, const float (&v)[4]) {
  unsigned q[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dgd_split3(v[i], q[0][i], q[1][i], q[2][i]);
#pragma unroll
  for (int p = 0; p < 3; ++p)
    *reinterpret_cast<uint2*>(Ht + p * FD_PART + col * FD_ROW + row0) = make_uint2(q[p][0] | (q[p][1] << 16), q[p][2] | (q[p][3] << 16));
}

struct FdW { const float *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4; };

__global__ void __launch_bounds__(FD_THREADS)
k_fused_fwd_d(int F, int C, FdW gw, TailW tw, const float* __restrict__ xin, const int* __restrict__ rowptr,
              const int* __restrict__ colidx, const float* __restrict__ dinv, const int* __restrict__ graph_ptr,
              float* __restrict__ axg, float* __restrict__ x1, float* __restrict__ x2, float* __restrict__ x3,
              float* __restrict__ x4, float* __restrict__ pooled, int* __restrict__ perm, float* __restrict__ a5g,
              float* __restrict__ a6g, float* __restrict__ a1dg, uint8_t* __restrict__ maskg, float* __restrict__ logp,
              int training, uint64_t seed, unsigned int* __restrict__ err, unsigned int epoch, unsigned long long* dbg,
              int B, DgPrepRider rd) {
  if ((int)blockIdx.x >= B) {    // rider range: phase A of the NEXT batch's graph preparation (dg_prep.h), as on k_readout_fwd
    dg_rider_phase_a(((int)blockIdx.x - B) * FD_THREADS + (int)threadIdx.x, rd);
    return;
  }
#define FD_MARK(k) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[k] = clock64(); } while (0)
#ifdef DGD_TIMING
  const unsigned long long tstart_ = clock64();
#endif
  FD_MARK(0);
  DG_DYN_SMEM(char, smem);
  // LDS plan: [Ht0 | Ht1] (2 x 37.5 KiB; the readout's 32-KiB region aliases it afterwards) | xt tiles 12 x 2304 B |
  //           bits 192 x 6 x 4 | dv, x4s (192 floats each) | tab | readout small
  unsigned short* Ht0 = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Ht1 = Ht0 + FD_HT;
  float* xts = reinterpret_cast<float*>(Ht1 + FD_HT);
  unsigned* bits = reinterpret_cast<unsigned*>(xts + (FD_NMAX / 16) * 16 * DGD_XT);
  float* dv = reinterpret_cast<float*>(bits + FD_NMAX * FD_KW);
  float* x4s = dv + FD_NMAX;
  uint2* tab = reinterpret_cast<uint2*>(x4s + FD_NMAX);
  float* W1t = reinterpret_cast<float*>(tab + 16);         // [k][33]: W^T of conv1 (rows k >= F zero), conv2, conv3 --
  float* W2t = W1t + 32 * FD_WS;                           // loaded ONCE per workgroup with coalesced reads; a per-lane
  float* W3t = W2t + 32 * FD_WS;                           // register copy costs 48 scattered loads x 16 waves (13 us!)
  char* small = reinterpret_cast<char*>(W3t + 32 * FD_WS);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, nl = lane & 15;
  const int n0 = graph_ptr[b], n = graph_ptr[b + 1] - n0;
  if (n > FD_NMAX) {                     // host hint (max_nodes) violated: flag, produce nothing
    if (tid == 0) { err[1] = epoch; err[3] = ~epoch; }
    return;
  }
  FD_MARK(14);
  const int K32 = (n + 31) >> 5;
  const int m0 = wave * 16;              // this wave's tile (live iff m0 < n)
  const bool live = m0 < n;
  // ---- prologue: everything in flight at once -------------------------------------------------------------------
  if (tid < 16) tab[tid] = make_uint2(((tid & 1) ? 0x3f80u : 0u) | ((tid & 2) ? 0x3f800000u : 0u),
                                      ((tid & 4) ? 0x3f80u : 0u) | ((tid & 8) ? 0x3f800000u : 0u));
  for (int t = tid; t < FD_NMAX * FD_KW; t += FD_THREADS) bits[t] = 0u;
  // both hs buffers: k < 32*K32 of every (part, column) -- the block products read exactly that range; rows >= n and
  // columns >= F must be 0.  16-B stores: (2 buffers x 96 columns) x 4*K32 pieces
  for (int t = tid; t < 2 * 96 * 4 * K32; t += FD_THREADS) {
    const int pc = t / (4 * K32), piece = t - pc * (4 * K32);
    *reinterpret_cast<uint4*>(Ht0 + pc * FD_ROW + 8 * piece) = make_uint4(0u, 0u, 0u, 0u);
  }
  FD_MARK(15);
  {   // transposed weights -> LDS (thread t: element (row cc = t >> 5, column k = t & 31) of the [32,32] matrices)
    const int cc = tid >> 5, k = tid & 31;
    W2t[k * FD_WS + cc] = gw.W2[tid];
    W3t[k * FD_WS + cc] = gw.W3[tid];
    W1t[k * FD_WS + cc] = 0.f;
  }
  float w4v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w4v[k] = gw.W4[8 * (lane & 3) + k];
  const float b1c0 = gw.b1[nl], b1c1 = gw.b1[16 + nl], b2c0 = gw.b2[nl], b2c1 = gw.b2[16 + nl];
  const float b3c0 = gw.b3[nl], b3c1 = gw.b3[16 + nl], b4s = gw.b4[0];
  float dpre[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int m = m0 + kq * 4 + r; dpre[r] = m < n ? dinv[n0 + m] : 0.f; }
  for (int t = tid; t < n; t += FD_THREADS) dv[t] = dinv[n0 + t];
  __syncthreads();                       // zeroed LDS visible
  for (int t = tid; t < 32 * F; t += FD_THREADS) { const int cc = t / F, k = t - cc * F; W1t[k * FD_WS + cc] = gw.W1[t]; }
  FD_MARK(1);
  // adjacency bitmap: wave per row, lanes over the row's neighbours; self bit by lane 0
  {
    bool bad = false;
    for (int i = wave; i < n; i += FD_THREADS / 64) {
      const int s = rowptr[n0 + i], e = rowptr[n0 + i + 1];
      if (lane == 0) atomicOr(&bits[i * FD_KW + (i >> 5)], 1u << (i & 31));
      for (int q = s + lane; q < e; q += 64) {
        const int j = colidx[q] - n0;
        if ((unsigned)j < (unsigned)n) atomicOr(&bits[i * FD_KW + (j >> 5)], 1u << (j & 31)); else bad = true;
      }
    }
    if (bad) { err[1] = epoch; err[3] = ~epoch; }           // an edge left its graph: the batch is not block-diagonal
  }
  // conv1 operand: xs = dinv * x -> Ht0 (columns < F), thread (column c, 4 consecutive rows)
  for (int t = tid; t < 32 * (FD_NMAX / 4); t += FD_THREADS) {
    const int c = t & 31, r0 = 4 * (t >> 5);
    if (c < F && r0 < n) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float xv = 0.f;
        if (r0 + i < n) { xv = dinv[n0 + r0 + i] * xin[(size_t)(n0 + r0 + i) * F + c]; DG_OPAQUE_V(xv); }
        v[i] = xv;
      }
      fd_store_hs4(Ht0, c, r0, v);
    }
  }
  __syncthreads();
  FD_MARK(2);
  unsigned wb[FD_KW];
#pragma unroll
  for (int u = 0; u < FD_KW; ++u) wb[u] = (live && u < K32) ? bits[(m0 + nl) * FD_KW + u] : 0u;   // rows >= n are empty
  float* xt = xts + wave * 16 * DGD_XT;
  const int rows_live = min(16, n - m0);

  // ---- conv1 (aggregate-first): ax = dv * (bits . xs), x1 = tanh(ax W1^T + b1), hs2 = dv * (x1 W2^T) -> Ht1 --------
  if (live) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int u = 0; u < FD_KW; ++u)
      if (u < K32 && __builtin_amdgcn_ballot_w64(wb[u] != 0u) != 0ull) {
        if (F > 16) dgd_mma_word<2, 3, FD_ROW>(wb[u], Ht0, 32 * u, lane, tab, acc);
        else { f32x4 a1[1] = {acc[0]}; dgd_mma_word<1, 3, FD_ROW>(wb[u], Ht0, 32 * u, lane, tab, a1); acc[0] = a1[0]; }
      }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r, col = nb * 16 + nl;
        const float ax = dpre[r] * acc[nb][r];
        xt[row * DGD_XT + col] = ax;
        if (col < F && row < rows_live) axg[(size_t)(n0 + m0 + row) * F + col] = ax;
      }
    dgd_wave_sync();
    const int F4 = (F + 3) >> 2;
    float a[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a[kk] = kk < F4 ? xt[nl * DGD_XT + 4 * kk + kq] : 0.f;
    f32x4 d1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      if (kk < F4) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          d1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], W1t[(4 * kk + kq) * FD_WS + nb * 16 + nl], d1[nb], 0, 0, 0);
      }
    dgd_wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xt[(kq * 4 + r) * DGD_XT + nl] = dg_tanh(d1[0][r] + b1c0);
      xt[(kq * 4 + r) * DGD_XT + 16 + nl] = dg_tanh(d1[1][r] + b1c1);
    }
  }
  // shared tail of a 32-wide layer: xt holds the activated tile; store its rows, next X.W^T, split into `Hn`
  auto finish32 = [&](float* __restrict__ xout, const float* __restrict__ Wt, unsigned short* Hn) {
    dgd_wave_sync();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = p * 8 + (lane >> 3), q = lane & 7;
      const float4 v = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 4 * q);
      if (row < rows_live) *reinterpret_cast<float4*>(xout + (size_t)(n0 + m0 + row) * 32 + 4 * q) = v;
    }
    float a[8], wv[2][8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      a[kk] = xt[nl * DGD_XT + 4 * kk + kq];
      wv[0][kk] = Wt[(4 * kk + kq) * FD_WS + nl]; wv[1][kk] = Wt[(4 * kk + kq) * FD_WS + 16 + nl];      // B[k][n] = W[n][k]
    }
    f32x4 d2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) d2[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], wv[nb][kk], d2[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dpre[r] * d2[nb][r];        // rows >= n: dpre = 0 -> zeros
      fd_store_hs4(Hn, nb * 16 + nl, m0 + kq * 4, v);
    }
  };
  if (live) finish32(x1, W2t, Ht1);
  dg_lds_barrier();
  FD_MARK(3);

  // ---- conv2: from Ht1, hs3 -> Ht0 ----------------------------------------------------------------------------------
  auto aggregate32 = [&](const unsigned short* Hc, float bc0, float bc1) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int u = 0; u < FD_KW; ++u)
      if (u < K32 && __builtin_amdgcn_ballot_w64(wb[u] != 0u) != 0ull) dgd_mma_word<2, 3, FD_ROW>(wb[u], Hc, 32 * u, lane, tab, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xt[(kq * 4 + r) * DGD_XT + nl] = dg_tanh(fmaf(dpre[r], acc[0][r], bc0));
      xt[(kq * 4 + r) * DGD_XT + 16 + nl] = dg_tanh(fmaf(dpre[r], acc[1][r], bc1));
    }
  };
  if (live) { aggregate32(Ht1, b2c0, b2c1); finish32(x2, W3t, Ht0); }
  dg_lds_barrier();
  FD_MARK(4);

  // ---- conv3: from Ht0; next linear is 32 -> 1: h4s = dv * (x3 . w4) -> column 0 of Ht1 -------------------------------
  for (int t = tid; t < 96 * 4 * K32; t += FD_THREADS) {        // (only column 0 is rewritten below)
    const int pc = t / (4 * K32), piece = t - pc * (4 * K32);
    *reinterpret_cast<uint4*>(Ht1 + pc * FD_ROW + 8 * piece) = make_uint4(0u, 0u, 0u, 0u);
  }
  if (live) {
    aggregate32(Ht0, b3c0, b3c1);
    dgd_wave_sync();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = p * 8 + (lane >> 3), q = lane & 7;
      const float4 v = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 4 * q);
      if (row < rows_live) *reinterpret_cast<float4*>(x3 + (size_t)(n0 + m0 + row) * 32 + 4 * q) = v;
    }
    const int row = lane >> 2, seg = lane & 3;
    const float4 va = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 8 * seg);
    const float4 vb = *reinterpret_cast<const float4*>(xt + row * DGD_XT + 8 * seg + 4);
    float p = va.x * w4v[0];
    p = fmaf(va.y, w4v[1], p); p = fmaf(va.z, w4v[2], p); p = fmaf(va.w, w4v[3], p);
    p = fmaf(vb.x, w4v[4], p); p = fmaf(vb.y, w4v[5], p); p = fmaf(vb.z, w4v[6], p); p = fmaf(vb.w, w4v[7], p);
    p += __shfl_xor(p, 1);
    p += __shfl_xor(p, 2);
    dgd_wave_sync();
    if (seg == 0) xt[row] = row < rows_live ? dv[m0 + row] * p : 0.f;      // h4s of the tile's 16 rows
  }
  dg_lds_barrier();                      // Ht1 cleared by everyone, h4s tiles written
  if (live && lane < 4) {                // lane l: rows 4l .. 4l+3 of the tile -> column 0, three parts
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = xt[4 * lane + i];
    fd_store_hs4(Ht1, 0, m0 + 4 * lane, v);
  }
  dg_lds_barrier();
  FD_MARK(5);

  // ---- conv4 (32 -> 1): x4 = tanh(dv * (bits . h4s) + b4): sort keys in LDS + the saved slab --------------------------
  if (live) {
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int u = 0; u < FD_KW; ++u)
      if (u < K32 && __builtin_amdgcn_ballot_w64(wb[u] != 0u) != 0ull) dgd_mma_word<1, 3, FD_ROW>(wb[u], Ht1, 32 * u, lane, tab, acc);
    if (nl == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + kq * 4 + r;
        if (m < n) { const float v4 = dg_tanh(fmaf(dpre[r], acc[0][r], b4s)); x4s[m] = v4; x4[n0 + m] = v4; }
      }
    }
  }
  __syncthreads();      // full barrier: x1..x4 of this graph are complete and visible to this workgroup
  FD_MARK(6);
  if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[7] = dbg[6];

  // ---- SortPooling + dense tail (keys from LDS, rows from the slabs just written); region0 aliases the hs buffers ----
  const RdSmem M = dg_rd_carve(smem, small);
  dg_readout_fwd_body(M, b, n0, n, C, tw, x4s, 0, x1, x2, x3, x4, pooled, perm, a5g, a6g, a1dg, maskg, logp, training, seed, dbg);
#ifdef DGD_TIMING
  if (dbg && tid == 0) { dbg[16 + 2 * b] = clock64() - tstart_; dbg[17 + 2 * b] = n; }
#endif
}

#define FD_LDS_BYTES (2 * FD_HT * 2 + (FD_NMAX / 16) * 16 * DGD_XT * 4 + FD_NMAX * FD_KW * 4 + 2 * FD_NMAX * 4 + 16 * 8 + 3 * 32 * FD_WS * 4 + RD_SMALL_BYTES + 64)

int dg_fused_d_max_nodes() { return FD_NMAX; }

int dg_launch_fused_fwd_d(int N, int B, int F, int C, const float* params, const DgParams* pl, const float* x,
                          const int32_t* rowptr, const int32_t* colidx, const float* dinv, const int32_t* graph_ptr, float* ax,
                          float* x1, float* x2, float* x3, float* x4, float* pooled, int32_t* perm, float* a5, float* a6,
                          float* a1d, uint8_t* drop_mask, float* logp, int training, uint64_t seed, int32_t* err, uint32_t epoch,
                          hipStream_t s, const DgPrepRider* rider, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (N <= 0 || B <= 0 || F < 1 || F > DG_AF_MAX_F) return DGCNN_EINVAL;
  DgPrepRider rd{};
  if (rider) rd = *rider;
  static_assert(2 * FD_HT * 2 >= RD_REGION0_BYTES, "the readout's region aliases the hs buffers");
  static DgPerDeviceOnce attr_once;
  if (attr_once.needed()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_fwd_d), hipFuncAttributeMaxDynamicSharedMemorySize,
                            FD_LDS_BYTES) != hipSuccess)
      return DGCNN_ELAUNCH;
    attr_once.done();
  }
  FdW gw;
  gw.W1 = params + pl->off[0]; gw.b1 = params + pl->off[1]; gw.W2 = params + pl->off[2]; gw.b2 = params + pl->off[3];
  gw.W3 = params + pl->off[4]; gw.b3 = params + pl->off[5]; gw.W4 = params + pl->off[6]; gw.b4 = params + pl->off[7];
  hipExtLaunchKernelGGL(k_fused_fwd_d, dim3(B + rd.nblk), dim3(FD_THREADS), FD_LDS_BYTES, s, ev_start, ev_stop, 0, F, C, gw,
                        dg_tail_w(params, pl), x, rowptr, colidx, dinv, graph_ptr, ax, x1, x2, x3, x4, pooled, perm, a5, a6, a1d,
                        drop_mask, logp, training, seed, reinterpret_cast<unsigned int*>(err), epoch, dg_debug_buffer(), B, rd);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// =================================================================================================================
// host launchers
// =================================================================================================================
static inline int dgd_grid(const DgDense* G) {      // persistent: every workgroup takes DGD_SPLITS / grid equal-cost shares
  return G->NW < DGD_MAX_GRID ? G->NW : DGD_MAX_GRID;       // (NW = upper bound of the item count; <= 100000, api.hip)
}

int dg_launch_gcn_fwd32d(int mode, int bf16_in, int bf16_out, const DgDense* G, const float* dinv, const void* hs,
                         const float* bias, float* xout, const float* Wnext, void* hs_next, hipStream_t s,
                         hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (!G || G->NW <= 0) return DGCNN_EINVAL;
  if (mode != 0) bf16_out = 0;            // only the 32x32 linear step has a bf16 form (the 32->1 output is a fp32 scalar)
#ifdef DGD_WAVE_FWD32
#define DGD_L(M, BI, BO) hipExtLaunchKernelGGL((k_gcn_fwd32w<M, BI, BO>), dim3(G->NW), dim3(DGD_THREADS), 0, s, ev_start, \
                                               ev_stop, 0, *G, dinv, hs, bias, xout, Wnext, hs_next, dg_debug_buffer())
#else
#define DGD_L(M, BI, BO) hipExtLaunchKernelGGL((k_gcn_fwd32d<M, BI, BO>), dim3(dgd_grid(G)), dim3(DGD_THREADS), 0, s, ev_start, \
                                               ev_stop, 0, *G, dinv, hs, bias, xout, Wnext, hs_next, dg_debug_buffer())
#endif
  if (mode == 0) {
    if (bf16_in && bf16_out) DGD_L(0, true, true); else if (bf16_in) DGD_L(0, true, false);
    else if (bf16_out) DGD_L(0, false, true); else DGD_L(0, false, false);
  } else if (mode == 1) { if (bf16_in) DGD_L(1, true, false); else DGD_L(1, false, false); }
  else { if (bf16_in) DGD_L(2, true, false); else DGD_L(2, false, false); }
#undef DGD_L
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

int dg_launch_gcn_fwd_af_d(int bf16_out, const DgDense* G, int F, const float* dinv, const float* xs, const float* W1,
                           const float* bias, float* ax, float* xout, const float* Wnext, void* hs_next, hipStream_t s,
                           hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (!G || G->NW <= 0 || F < 1 || F > DG_AF_MAX_F) return DGCNN_EINVAL;
#define DGD_L(NB, BF) hipExtLaunchKernelGGL((k_gcn_fwd_af_d<NB, BF>), dim3(dgd_grid(G)), dim3(DGD_THREADS), 0, s, ev_start, \
                                            ev_stop, 0, *G, F, dinv, xs, W1, bias, ax, xout, Wnext, hs_next)
  if (F <= 16) { if (bf16_out) DGD_L(1, true); else DGD_L(1, false); }
  else { if (bf16_out) DGD_L(2, true); else DGD_L(2, false); }
#undef DGD_L
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

int dg_launch_gcn_fwd1d(const DgDense* G, const float* dinv, const float* h4s, const float* bias, float* x4, hipStream_t s) {
  if (!G || G->NW <= 0) return DGCNN_EINVAL;
  hipLaunchKernelGGL(k_gcn_fwd1d, dim3(dgd_grid(G)), dim3(DGD_THREADS), 0, s, *G, dinv, h4s, bias, x4);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

int dg_launch_gcn_bwd32d(const DgDense* G, const float* dinv, const float* gas, const float* Wl, const float* xprev,
                         const float* gpprev, float* gas_prev, float* part, int P32, hipStream_t s, const float* ax, int Fa,
                         float* part1) {
  if (!G || G->NW <= 0 || P32 <= 0) return DGCNN_EINVAL;
  if (ax) {
    if (Fa < 1 || Fa > DG_AF_MAX_F || !part1) return DGCNN_EINVAL;
    hipLaunchKernelGGL((k_gcn_bwd32d<true>), dim3(P32), dim3(DGD_THREADS), 0, s, *G, dinv, gas, Wl, xprev, gpprev, gas_prev,
                       part, ax, Fa, part1);
  } else {
    hipLaunchKernelGGL((k_gcn_bwd32d<false>), dim3(P32), dim3(DGD_THREADS), 0, s, *G, dinv, gas, Wl, xprev, gpprev, gas_prev,
                       part, nullptr, 0, nullptr);
  }
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

int dg_launch_gcn_bwd1d(const DgDense* G, const float* dinv, const float* gas4, const float* W4, const float* x3,
                        const float* gp3, float* gas3, float* pa4, int P1, hipStream_t s) {
  if (!G || G->NW <= 0 || P1 <= 0) return DGCNN_EINVAL;
  hipLaunchKernelGGL(k_gcn_bwd1d, dim3(P1), dim3(DGD_THREADS), 0, s, *G, dinv, gas4, W4, x3, gp3, gas3, pa4);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}
