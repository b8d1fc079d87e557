// dg_readout.h -- device code shared by the per-op kernels (tail.hip) and the fused per-graph
// kernels (fused.hip): SortPooling selection and the dense tail forward.
#pragma once
#include "dg_common.h"

#define SP_THREADS 256
#define SP_LDS_KEYS 4096
#define KCAT (DGCNN_K * DGCNN_CAT)   // 2910

__device__ __forceinline__ unsigned long long dg_pack_key(float key, int idx) {
  key = key + 0.0f;                       // -0.0 -> +0.0 so signed zeros tie like in torch.sort
  unsigned int u = __float_as_uint(key);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-order-preserving map
  u = ~u;                                            // descending key
  return ((unsigned long long)u << 32) | (unsigned int)idx;
}

// selects the first min(n,K) nodes of graph [n0, n0+n) into sel[0..K) (local indices, -1 = none).
// Works for any workgroup size that is a multiple of 64 (<= 1024).
__device__ __forceinline__ void dg_select_topk(const float* __restrict__ x4, int n0, int n, unsigned long long* keys,
                               unsigned long long* red, int* sel, unsigned long long* dbg = nullptr) {
#ifdef RD_FINE      // measurement build (tools/build_variant.sh rdfine "-DRD_FINE"; tools/phase_step_kernel.py prints the stamps)
#define SK_MARK(k) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[60 + (k)] = clock64(); } while (0)
#else
#define SK_MARK(k) do { } while (0)
#endif
  const int tid = threadIdx.x, T = blockDim.x;
  SK_MARK(0);
  const int m = n < DGCNN_K ? n : DGCNN_K;
  if (tid < DGCNN_K) sel[tid] = -1;
  // (LDS-only barriers in this path: nothing here depends on an outstanding global store -- a caller whose workgroup wrote
  //  the keys itself hands in an LDS copy -- so a workgroup's pending row stores drain while it ranks, up to the barrier at the
  //  end, which the gather of those rows needs anyway)
  if (n <= 256) {
    // rank by counting, P lanes per key (P = the largest power of two with P * n <= T, at most 16): lane `part` of key i counts
    // the keys j = part, part + P, ... below it; the P counts are added by xor-shuffles inside the aligned lane group.
    // (one thread per key walked all n keys: 126 dependent LDS reads = 3.8 k of a 126-node graph's 6.1 k cycles here.)
    // The key area is padded to a multiple of 8 P entries with a sentinel no key is above, so that the counting loop -- eight
    // keys per trip, all eight reads in flight -- has no bounds checks: 3 instructions per key instead of 8 (the loop is
    // executed by all 16 waves, 4 per SIMD: its instruction count times ~18 cycles is its duration).
    int lp = 0;
    while (lp < 4 && (n << (lp + 1)) <= T) ++lp;
    const int P = 1 << lp, i = tid >> lp, part = tid & (P - 1);
    const int npad = (n + 8 * P - 1) & ~(8 * P - 1);          // <= 256 + 127: inside the caller's SP_LDS_KEYS slots
    // (this thread's first key is requested BEFORE the barrier, unconditionally on a clamped node: behind it the load was a round
    //  trip of its own in front of the LDS store)
    const float k0 = x4[n0 + max(min(tid, n - 1), 0)];
#ifndef RD_TOPK_NO_ENTRY_BARRIER      // (measurement: callers that placed a full barrier right in front of this function do not need it)
    dg_lds_barrier();
#endif
    SK_MARK(1);
    if (tid < npad) keys[tid] = tid < n ? dg_pack_key(k0, tid) : ~0ull;
    for (int t = tid + T; t < npad; t += T) keys[t] = t < n ? dg_pack_key(x4[n0 + t], t) : ~0ull;
    dg_lds_barrier();
    SK_MARK(2);
    const bool on = i < n;
    const unsigned long long my = keys[on ? i : 0];
    int rank = 0;
    const unsigned long long* kp = keys + part;
    for (int j0 = 0; j0 < npad; j0 += 8 * P) {
      unsigned long long kj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) kj[u] = kp[j0 + u * P];
#pragma unroll
      for (int u = 0; u < 8; ++u) rank += kj[u] < my ? 1 : 0;
    }
    SK_MARK(3);
    for (int o = 1; o < P; o <<= 1) rank += __shfl_xor(rank, o);
    if (on && part == 0 && rank < DGCNN_K) sel[rank] = i;
    SK_MARK(4);
  } else {
    __syncthreads();
    // (graphs of up to SP_LDS_KEYS / 2 nodes keep their packed keys in LDS; larger ones -- DD's 5748-node graph -- re-pack them
    //  from the L2-resident key column in every pass: 4 + 1 scans instead of the 30 scans of a k-round selection, which took
    //  88 us for that graph, or the ~78 barrier-separated passes of a 4096-key bitonic sort)
    const bool in_lds = n <= SP_LDS_KEYS / 2;
    auto key_at = [&](int t) { return in_lds ? keys[t] : dg_pack_key(x4[n0 + t], t); };
    // Only the first K of the order are needed: RADIX SELECT of the K-th key's 32-bit value (four passes over 8-bit
    // digits, an LDS histogram per pass), then the <= K + ties candidates at or below it are ranked among themselves.
    // ~10 barriers in all, against the ~55 barrier-separated passes of a bitonic sort of 1024 keys (the largest graph of
    // a DD batch, 661 nodes, spent 41.8 k of its 73 k readout cycles sorting).  Integer atomics only: deterministic.
    // (scratch in the upper half of the key area -- every caller provides SP_LDS_KEYS slots: no static LDS of its own,
    //  the graph-per-workgroup kernels have none to spare)
    unsigned long long* cand = keys + SP_LDS_KEYS / 2;                               // [256]
    unsigned int* hist = reinterpret_cast<unsigned int*>(cand + 256);                // [2048] + 8 words of state
    unsigned int& s_prefix = hist[2048]; unsigned int& s_mask = hist[2049];
    unsigned int& s_need = hist[2050]; unsigned int& s_cnt = hist[2051]; unsigned int& s_stop = hist[2052];
    static_assert(256 * 8 + (2048 + 8) * 4 <= (SP_LDS_KEYS / 2) * 8, "candidates + histogram inside the upper half of the key area");
    if (in_lds) for (int t = tid; t < n; t += T) keys[t] = dg_pack_key(x4[n0 + t], t);
    if (tid == 0) { s_prefix = 0u; s_mask = 0u; s_need = (unsigned)m; s_cnt = 0u; s_stop = 0u; }
    // Digits of 11 + 11 + 10 bits (2048-bin LDS histogram), EARLY EXIT once the keys below the chosen bin plus the bin itself are
    // at most 256: they are ranked directly (below).  The keys are tanh outputs: sign + exponent hardly separate them, so with
    // 8-bit digits the first pass resolved almost nothing and all four passes -- three barriers and a scan of the keys each -- ran
    // (661-node DD graph: 11.6 k of the readout's 40 k cycles); the first 11-bit digit reaches two mantissa bits, the second
    // leaves a handful of keys per bin: two passes.  Same selection: the candidates are exactly the keys <= the bin's upper end,
    // ranked by their full 64-bit keys.
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0), wbits = pass == 2 ? 10 : 11;
      const unsigned int dmask = (1u << wbits) - 1u;
      for (int b = tid; b < 2048; b += T) hist[b] = 0u;
      __syncthreads();
      const unsigned int prefix = s_prefix, mask = s_mask, need = s_need;
      for (int t = tid; t < n; t += T) {
        const unsigned int u = (unsigned int)(key_at(t) >> 32);
        if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & dmask], 1u);
      }
      __syncthreads();
      if (tid < 64) {            // wave 0: lane l owns bins 32 l .. 32 l + 31 (ascending digit = ascending key)
        unsigned int c[32];
        const uint4* hp = reinterpret_cast<const uint4*>(hist + 32 * tid);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const uint4 v = hp[q]; c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; }
        unsigned int sum = 0u;
#pragma unroll
        for (int q = 0; q < 32; ++q) sum += c[q];
        unsigned int incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const unsigned int a = __shfl_up(incl, o); if (tid >= o) incl += a; }
        const unsigned int excl = incl - sum;
        if (excl < need && need <= incl) {             // exactly one lane
          unsigned int before = excl, bin = 32 * tid, cbin = 0u;
          bool found = false;
#pragma unroll
          for (int q = 0; q < 32; ++q) {
            if (!found) {
              if (before + c[q] < need) { before += c[q]; ++bin; }
              else { found = true; cbin = c[q]; }
            }
          }
          s_prefix = prefix | (bin << shift);
          s_mask = mask | (dmask << shift);
          s_need = need - before;
          if (((unsigned)m - (need - before)) + cbin <= 256u) s_stop = 1u;      // (keys below the bin) + (keys in it)
        }
      }
      __syncthreads();
      if (s_stop) break;         // (uniform: read behind the barrier)
    }
    const unsigned int ustar = s_prefix, umask = s_mask;      // the K-th key's value in the digits resolved so far
    for (int t = tid; t < n; t += T) {
      const unsigned long long k = key_at(t);
      if (((unsigned int)(k >> 32) & umask) <= ustar) {
        const unsigned int pos = atomicAdd(&s_cnt, 1u);
        if (pos < 256u) cand[pos] = k;
      }
    }
    __syncthreads();
    const int cnt = (int)s_cnt;
    if (cnt <= 256) {
      if (tid < cnt) {
        const unsigned long long my = cand[tid];
        int rank = 0;
        for (int j = 0; j < cnt; ++j) rank += cand[j] < my ? 1 : 0;
        if (rank < DGCNN_K) sel[rank] = (int)(my & 0xffffffffull);
      }
    } else if (n <= SP_LDS_KEYS) {      // more than 256 - K keys tie with the K-th one: full sort
      __syncthreads();
      if (!in_lds) { for (int t = tid; t < n; t += T) keys[t] = dg_pack_key(x4[n0 + t], t); __syncthreads(); }
      dg_block_bitonic<unsigned long long>(keys, n);
      if (tid < m) sel[tid] = (int)(keys[tid] & 0xffffffffull);
    } else {                     // ... in a graph too large for the LDS sort: one scan per selected node
      __syncthreads();
      unsigned long long prev = 0ull;
      for (int r = 0; r < m; ++r) {
        unsigned long long best = ~0ull;
        for (int t = tid; t < n; t += T) {
          const unsigned long long p = dg_pack_key(x4[n0 + t], t);
          if ((r == 0 || p > prev) && p < best) best = p;
        }
        for (int o = 32; o > 0; o >>= 1) {       // workgroup min: wave shuffle then LDS
          const unsigned long long other = __shfl_xor(best, o);
          best = other < best ? other : best;
        }
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        unsigned long long b0 = red[0];
        for (int w = 1; w < T / 64; ++w) b0 = red[w] < b0 ? red[w] : b0;
        prev = b0;
        if (tid == 0) sel[r] = (int)(b0 & 0xffffffffull);
        __syncthreads();
      }
    }
  }
  // (LDS only -- `sel`: a full barrier here would also wait, with vmcnt(0), for every global load the CALLER has in flight, i.e.
  //  the readout's cold parameter prefetch.  Every caller places a full barrier between its own stores of the rows / keys and
  //  this function.)
  dg_lds_barrier();
  SK_MARK(5);
#undef SK_MARK
}

__device__ __forceinline__ float dg_cat_load(const float* __restrict__ x1, const float* __restrict__ x2,
                                             const float* __restrict__ x3, const float* __restrict__ x4,
                                             int node, int c) {
  if (c < 32) return x1[(size_t)node * 32 + c];
  if (c < 64) return x2[(size_t)node * 32 + c - 32];
  if (c < 96) return x3[(size_t)node * 32 + c - 64];
  return x4[node];
}

#define RD_THREADS 1024
struct TailW {   // device pointers into the flat parameter buffer
  const float *W5, *b5, *W6, *b6, *Wf1, *bf1, *Wf2, *bf2;
};
static inline TailW dg_tail_w(const float* params, const DgParams* pl) {
  TailW w;
  w.W5 = params + pl->off[8];  w.b5 = params + pl->off[9];
  w.W6 = params + pl->off[10]; w.b6 = params + pl->off[11];
  w.Wf1 = params + pl->off[12]; w.bf1 = params + pl->off[13];
  w.Wf2 = params + pl->off[14]; w.bf2 = params + pl->off[15];
  return w;
}
#define NW5 (DGCNN_C5 * DGCNN_CAT)                 // 1552
#define NW6 (DGCNN_C6 * DGCNN_C5 * DGCNN_KW6)      // 2560


// LDS carve of the readout: region0 (>= RD_REGION0_BYTES, 16-B aligned) + small activations
#define RD_REGION0_BYTES 32768
struct RdSmem {
  unsigned long long* region0;   // sort keys, afterwards pooled rows + conv5/conv6 weights
  unsigned long long* red;       // [16]
  int* sel;                      // [K]
  float* a5s;                    // [16*30]
  float* p5;                     // [16*15]
  float* flat;                   // [352]
  float* a1s;                    // [128]
  float* lg;                     // [MAX_C]
  float* cpart;                  // [4][32][16] conv5 K-split partial tiles
  float* bias;                   // b5[16] | b6[32] | bf1[128] | bf2[MAX_C]
};
#define RD_SMALL_BYTES (16 * 8 + 32 * 4 + (DGCNN_C5 * DGCNN_K + DGCNN_C5 * DGCNN_T5 + DGCNN_FLAT + DGCNN_HID1 + DGCNN_MAX_C + 4 * 32 * 16 + 16 + 32 + 128 + DGCNN_MAX_C) * 4)
__device__ __forceinline__ RdSmem dg_rd_carve(void* region0, void* small) {
  RdSmem m;
  m.region0 = reinterpret_cast<unsigned long long*>(region0);
  char* p = reinterpret_cast<char*>(small);
  m.red = reinterpret_cast<unsigned long long*>(p); p += 16 * 8;
  m.sel = reinterpret_cast<int*>(p); p += 32 * 4;
  m.a5s = reinterpret_cast<float*>(p); p += DGCNN_C5 * DGCNN_K * 4;
  m.p5 = reinterpret_cast<float*>(p); p += DGCNN_C5 * DGCNN_T5 * 4;
  m.flat = reinterpret_cast<float*>(p); p += DGCNN_FLAT * 4;
  m.a1s = reinterpret_cast<float*>(p); p += DGCNN_HID1 * 4;
  m.lg = reinterpret_cast<float*>(p); p += DGCNN_MAX_C * 4;
  m.cpart = reinterpret_cast<float*>(p); p += 4 * 32 * 16 * 4;
  m.bias = reinterpret_cast<float*>(p);
  return m;
}

// SortPooling + dense tail for graph b, executed by one workgroup of RD_THREADS threads.
//   keys/key_n0: where the sort keys (channel 96) of this graph's nodes are read from (global x4 with
//   key_n0 = n0, or an LDS copy with key_n0 = 0).  Rows are gathered from the global slabs x1..x4.
// BIG (thousands of graphs per launch, registers capped at 64 for two workgroups per CU): nothing is held in registers
// over the sort or over conv5/conv6 -- the conv weights are loaded after the sort and classifier_1's rows at their
// use (every weight is L2-resident when thousands of workgroups read the same 209 KB; the early prefetch only bought
// 7 spilled registers = 58 MB of scratch traffic per launch at 2048 graphs).  Same arithmetic order: bit-identical.
// HEAD = false (large batches): stops at conv6's output; classifier_1/2 and log_softmax run batched over graphs
// (classifier.hip) instead of once per graph.
// IDLE (optional): work of the caller's that the 14 waves without a tile of conv5 do while waves 0 and 1 run it -- called as
// idle(wave, lane) by every wave; must not touch the readout's LDS plan, must not contain a barrier
// ... and at_fc2(tid): called by every thread at the start of the classifier_2 phase, where 13 of the 16 waves have nothing to do and
// classifier_1's forward rows have just left their registers (the one-launch training kernel requests classifier_1's rows for the
// BACKWARD there: 176 wave-level loads, ~2.5 k cycles of the CU's address path, that used to stand in front of the backward's first barrier)
struct RdNoIdle {
  __device__ __forceinline__ void operator()(int, int) const {}
  __device__ __forceinline__ void at_fc2(int) const {}
};
template <bool BIG = false, bool HEAD = true, class IDLE = RdNoIdle>
__device__ __forceinline__ void dg_readout_fwd_body(
    const RdSmem& M, int b, int n0, int n, int C, const TailW& w, const float* keys, int key_n0,
    const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ x3,
    const float* __restrict__ x4, float* __restrict__ pooled, int* __restrict__ perm, float* __restrict__ a5g,
    float* __restrict__ a6g, float* __restrict__ a1dg, uint8_t* __restrict__ maskg, float* __restrict__ logp,
    int training, uint64_t seed, unsigned long long* dbg = nullptr, const IDLE idle = IDLE{}) {
#define RD_MARK(k) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[k] = clock64(); } while (0)
  float* sp = reinterpret_cast<float*>(M.region0);        // [2910]
  float* W5s = sp + 2912;                                 // [1552]
  float* W6s = W5s + NW5;                                 // [2560]   (2912+1552+2560)*4 = 28096 <= 32768
  int* sel = M.sel;
  float *a5s = M.a5s, *p5 = M.p5, *flat = M.flat, *a1s = M.a1s, *lg = M.lg;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

  // conv5/conv6 weights -> LDS first: their global loads fly while the keys are sorted (the key area is
  // the first n*8 <= 4096*8 bytes of region0 only when n > 1456; W5s starts at byte 11648)
  float* bs = M.bias;        // every bias of the tail in LDS: no global load between the prefetch and its use
  // (ONE unconditional load through a selected pointer, stored to LDS behind the selection: as four `if (..) bs[tid] = w.b[..]`
  //  branches each was load -> s_waitcnt vmcnt(0) -> ds_write on the spot, a cold round trip -- the optimizer rewrites the biases
  //  every step -- in front of the sort)
  const float* bsrc = tid < 16 ? w.b5 + tid : (tid < 48 ? w.b6 + (tid - 16) : (tid < 176 ? w.bf1 + (tid - 48) : (tid < 176 + C ? w.bf2 + (tid - 176) : w.b5)));
  const float bval = *bsrc;
  // conv5 / conv6 weights: loads issued now, held in registers over the sort (the key area may overlap W5s/W6s),
  // stored to LDS afterwards -- no memory round trip of theirs is left on the critical path
  DgStage4<NW5, RD_THREADS> st5;
  DgStage4<NW6, RD_THREADS> st6;
  if (!BIG) { st5.load(w.W5, tid); st6.load(w.W6, tid); }
  float wf2a = 0.f, wf2b = 0.f;                   // classifier_2 row of class `wv` (used at the very end: no cold load there)
  if (!BIG && wv < C) { wf2a = w.Wf2[wv * DGCNN_HID1 + lane]; wf2b = w.Wf2[wv * DGCNN_HID1 + lane + 64]; }
  dg_select_topk(keys, key_n0, n, M.region0, M.red, sel, dbg);   // ends with a barrier
  if (tid < 176 + C) bs[tid] = bval;                              // (read behind the barrier that follows the gather)
  if (BIG) { st5.load(w.W5, tid); st6.load(w.W6, tid); }
  RD_MARK(8);
  if (tid < DGCNN_K) perm[b * DGCNN_K + tid] = sel[tid] >= 0 ? n0 + sel[tid] : -1;
  {   // SortPooling gather, ROW-WISE: 32 lanes per selected node copy its x1 | x2 | x3 rows (three coalesced 128-byte loads on a
      // clamped row, selected afterwards) and lane 0 its x4 -- no division by 97, no four-way slab branch per element (the
      // element-wise form was ~75 instructions per thread, executed by all 16 waves)
    static_assert(RD_THREADS == 32 * 32 && DGCNN_K <= 32 && DGCNN_CAT == 97, "one 32-lane group per SortPooling slot");
    const int s = tid >> 5, c = tid & 31;
    const int ln = s < DGCNN_K ? sel[s] : -1;
    const size_t node = (size_t)(n0 + (ln > 0 ? ln : 0));
    float v1 = x1[node * 32 + c], v2 = x2[node * 32 + c], v3 = x3[node * 32 + c], v4 = x4[node];
    if (ln < 0) { v1 = 0.f; v2 = 0.f; v3 = 0.f; v4 = 0.f; }
    if (s < DGCNN_K) {
      float* sr = sp + s * DGCNN_CAT;
      float* pr = pooled + (size_t)b * KCAT + s * DGCNN_CAT;
      sr[c] = v1; sr[32 + c] = v2; sr[64 + c] = v3;
      pr[c] = v1; pr[32 + c] = v2; pr[64 + c] = v3;
      if (c == 0) { sr[96] = v4; pr[96] = v4; }
    }
  }
  st5.store(W5s, tid); st6.store(W6s, tid);
  dg_lds_barrier();      // (LDS only: nothing below reads what this workgroup stored to global memory -- the pooled rows, perm)
  RD_MARK(9);
  // classifier_1's weights (180 KB, rewritten by the optimizer every step, so never cache-warm) are this
  // kernel's longest memory wait.  VMEM loads complete in order, so they are issued only NOW -- after every
  // other global load of the kernel has been consumed -- into registers; until their use in classifier_1
  // there is no further global load and only LDS-only barriers, so they land while conv5 / pool / conv6 run.
  // Row blocks are rotated per graph so that the ~50 concurrent workgroups do not request the same lines.
  // Mapping: a wave owns 8 output rows; lane = (row r = lane >> 3, part p = lane & 7); a 352-float row is 8 parts
  // x 11 float4, so the whole 8-row block is 11 x 16-byte loads per lane (vs 48 dword loads with lanes along
  // the row): 4.4x fewer vector-memory instructions through the CU's address unit for the same 180 KB.
  // The 180 KB stream keeps the CU's address unit busy for ~2900 cycles (64 B/clk); a barrier cannot complete before
  // every wave has ISSUED the loads that precede it in its instruction stream, so the prefetch is cut in three and
  // spread over conv5's MFMA, conv5's combine and pool/conv6 instead of standing in front of the first barrier.
  float4 wf[11];
  const float* wr = w.Wf1 + (size_t)(((wv + b) & 15) * 8 + (lane >> 3)) * DGCNN_FLAT + 4 * (lane & 7);
  if (!BIG) {
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) wf[jq] = *reinterpret_cast<const float4*>(wr + 32 * jq);
  }
  // conv5 + bias + ReLU + MaxPool1d(2,2) as ONE phase on two waves (round 4; was: K split over 4 waves per tile -> barrier ->
  // combine on 480 threads -> barrier -> pool on 240 threads -> barrier).  z5[s][o] = sum_m sp[s][m] W5[o][m]:
  // [32(30) x 16] = [32 x 100(97)] . [100 x 16], wave mt owns slots s = 16 mt .. 16 mt + 15, 25 k-steps in two accumulator
  // chains.  The accumulator lane (o = lane & 15, kq) holds s = 16 mt + 4 kq + 0..3 -- two pooling pairs: bias, ReLU and the
  // pool happen in registers.  (A phase executed by 2 waves costs their instruction count once; the three phases it replaces
  // were executed by 8, 8 and 4 waves and separated by barriers.)  output index o*30+s ([B,16,30])
  if (wv < 2) {
    const int mt = wv, mi = lane & 15, kq = lane >> 4;
    const int srow = 16 * mt + mi;
    const bool sok = srow < DGCNN_K;
    const float* ap = sp + (sok ? srow : 0) * DGCNN_CAT + kq;
    const float* bp = W5s + mi * DGCNN_CAT + kq;
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u0 = 0; u0 < 25; u0 += 9) {           // operands of nine steps at a time (all 25 held 50 registers beside the prefetch)
      float av[9], bv[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) {                // m = 4 u + kq; the last step holds m = 96 only (kq = 0)
        const int u = u0 + q;
        if (u < 25) {
          const bool mok = u < 24 || kq == 0;
          const float a_ = ap[mok ? 4 * u : 0], b_ = bp[mok ? 4 * u : 0];
          av[q] = (sok && mok) ? a_ : 0.f;
          bv[q] = mok ? b_ : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int u = u0 + q;
        if (u < 25) {
          if (u & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], d1, 0, 0, 0);
          else d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], d0, 0, 0, 0);
        }
      }
    }
    const float bias = bs[mi];
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf((d0[r] + d1[r]) + bias, 0.f);
    const int s0 = 16 * mt + 4 * kq;              // this lane's four slots s0 .. s0 + 3 of channel o = mi
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (s0 + r < DGCNN_K) {
        a5s[mi * DGCNN_K + s0 + r] = v[r];
        a5g[(size_t)b * (DGCNN_C5 * DGCNN_K) + mi * DGCNN_K + s0 + r] = v[r];
      }
    if (s0 < DGCNN_K) p5[mi * DGCNN_T5 + (s0 >> 1)] = fmaxf(v[0], v[1]);           // MaxPool1d(2,2): [16,30] -> [16,15]
    if (s0 + 2 < DGCNN_K) p5[mi * DGCNN_T5 + (s0 >> 1) + 1] = fmaxf(v[2], v[3]);
  }
  if (!BIG) {
#pragma unroll
    for (int jq = 4; jq < 8; ++jq) wf[jq] = *reinterpret_cast<const float4*>(wr + 32 * jq);
  }
  idle(wv, lane);
  dg_lds_barrier();
  RD_MARK(10);
  if (!BIG) {
#pragma unroll
    for (int jq = 8; jq < 11; ++jq) wf[jq] = *reinterpret_cast<const float4*>(wr + 32 * jq);
  }
  // conv6 + bias + ReLU as ONE phase on two waves: z6[oc][t] = sum_{c,d} W6[oc][c][d] p5[c][t+d] -> [32 x 16(11)] = [32 x 80] . [80 x 16],
  // wave mt owns oc = 16 mt .. 16 mt + 15; k-slot of (step u, lane group kq) = 20 kq + u, so that c = 4 kq + u / 5 and d = u % 5
  // are affine in kq (no division); flat index oc*11+t (x.view(B,-1), model.py:40)
  if (wv < 2) {
    const int mt = wv, mi = lane & 15, kq = lane >> 4;
    const float* ap = W6s + (16 * mt + mi) * (DGCNN_C5 * DGCNN_KW6) + 20 * kq;
    const bool tok = mi < DGCNN_T6;
    const float* bp = p5 + (4 * kq) * DGCNN_T5 + (tok ? mi : 0);
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u0 = 0; u0 < 20; u0 += 10) {
      float av[10], bv[10];
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        const int u = u0 + q;
        av[q] = ap[u];
        const float b_ = bp[(u / DGCNN_KW6) * DGCNN_T5 + (u % DGCNN_KW6)];
        bv[q] = tok ? b_ : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        if ((u0 + q) & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], d1, 0, 0, 0);
        else d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], d0, 0, 0, 0);
      }
    }
    if (tok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oc = 16 * mt + 4 * kq + r;
        const float acc = fmaxf((d0[r] + d1[r]) + bs[16 + oc], 0.f);
        flat[oc * DGCNN_T6 + mi] = acc;
        a6g[(size_t)b * DGCNN_FLAT + oc * DGCNN_T6 + mi] = acc;
      }
    }
  }
  if (!HEAD) return;
  dg_lds_barrier();
  RD_MARK(11);
  // classifier_1: 352 -> 128, ReLU, Dropout(0.5).  16 waves x 8 rows; the weights were prefetched before conv5;
  // 44 fmas per lane, then the 8 parts of a row are combined by a fixed xor butterfly (1, 2, 4).
  {
    const int p8 = lane & 7;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (BIG) {
#pragma unroll
      for (int jq = 0; jq < 11; ++jq) wf[jq] = *reinterpret_cast<const float4*>(wr + 32 * jq);
    }
#pragma unroll
    for (int jq = 0; jq < 11; ++jq) {
      const float4 f = *reinterpret_cast<const float4*>(flat + 32 * jq + 4 * p8);
      a0 = fmaf(wf[jq].x, f.x, a0); a1 = fmaf(wf[jq].y, f.y, a1);
      a2 = fmaf(wf[jq].z, f.z, a2); a3 = fmaf(wf[jq].w, f.w, a3);
    }
    float tot = (a0 + a1) + (a2 + a3);
    tot += __shfl_xor(tot, 1);
    tot += __shfl_xor(tot, 2);
    tot += __shfl_xor(tot, 4);
    if (p8 == 0) {
      const int j = ((wv + b) & 15) * 8 + (lane >> 3);    // 50 CUs do not hit the same Wf1 lines in lockstep
      float av = fmaxf(tot + bs[48 + j], 0.f);
      uint8_t keep = 1;
      if (training) {
        keep = dg_keep(seed, (uint64_t)b * DGCNN_HID1 + j) ? 1 : 0;
        av = keep ? av * 2.0f : 0.f;     // p = 0.5 -> scale 1/(1-p) = 2
      }
      a1s[j] = av;
      a1dg[(size_t)b * DGCNN_HID1 + j] = av;
      maskg[(size_t)b * DGCNN_HID1 + j] = keep;
    }
  }
  dg_lds_barrier();
  RD_MARK(12);
  idle.at_fc2(tid);
  // classifier_2: 128 -> C, wave per class
  for (int c = wv; c < C; c += RD_THREADS / 64) {
    const float* wr = w.Wf2 + c * DGCNN_HID1;
    const bool pre = !BIG && c == wv;             // first class of this wave: row prefetched with the conv weights
    float acc = (pre ? wf2a : wr[lane]) * a1s[lane];
    acc = fmaf(pre ? wf2b : wr[lane + 64], a1s[lane + 64], acc);
    acc = dg_wave_sum(acc);
    if (lane == 0) lg[c] = acc + bs[176 + c];
    // the prefetched row also goes to LDS (the conv5 partial-tile area, dead since conv5): the merged training kernels'
    // backward half reads classifier_2's weights from there instead of loading them again
    if (pre && c < 16) { M.cpart[c * DGCNN_HID1 + lane] = wf2a; M.cpart[c * DGCNN_HID1 + lane + 64] = wf2b; }
  }
  dg_lds_barrier();
  // log_softmax over C (C <= 64): wave 0
  if (wv == 0) {
    const float v = lane < C ? lg[lane] : -INFINITY;
    float mx = v;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float e = lane < C ? expf(v - mx) : 0.f;
    e = dg_wave_sum(e);
    if (lane < C) { const float lpv = (v - mx) - logf(e); logp[(size_t)b * C + lane] = lpv; lg[lane] = lpv; }
    // (lg now holds the log-probabilities: the merged training kernel's backward half reads them from LDS)
  }
  RD_MARK(13);
#undef RD_MARK
}
