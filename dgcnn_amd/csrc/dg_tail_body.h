// dg_tail_body.h -- body of the readout backward for one graph (one workgroup of RD_THREADS threads), shared by k_tail_bwd,
// the merged training kernel k_readout_tail (tail.hip) and the graph-chain training kernel (gcn_chain.hip).
#pragma once
#include "dg_common.h"
#include "dg_readout.h"

// ---------------------------------------------------------------------------------------------
// readout backward (data gradients), one workgroup of 1024 threads per graph.  Also scatters the
// SortPooling gradient to dense per-node slabs gp1..gp3 [N,32] and produces
// gas4 = dinv * dL/d(pre-activation of conv4).
// ---------------------------------------------------------------------------------------------
// BIG (many graphs): classifier_1's weights are NOT prefetched into 64 registers at kernel start but read in 4-row
// chunks at their use (they are L2-resident when thousands of workgroups stream the same 180 KB), and the registers
// are capped at 64 so that two workgroups share a CU.  Same arithmetic order: bit-identical results.
// body of the readout backward for graph b (one workgroup of RD_THREADS threads); shared by k_tail_bwd and the merged
// training kernel k_readout_tail (forward readout + this, one launch)
struct TbExt { const float *sp, *W5s, *W6s, *lg, *flat, *a5s, *a1s; const int* sel; int yb;
               const float *wf2s, *x4l, *dvl; int n0, n;
               const float4* wpre; };      // (wpre != null: classifier_1's rows of this thread, already requested by the forward half)     // (LDSOPS: n0 / n = the graph's node range, read once by the caller)     // (optional LDS copies: classifier_2's rows [<= 16][128]; conv4's outputs and dinv by LOCAL node)      // merged kernel: operands the forward left in LDS (+ the label, loaded at kernel start)
// the one-launch training kernel whose GCN backward follows in the same workgroup: the SortPooling gradient STAYS IN LDS in its
// sparse form -- the <= 30 selected nodes' rows gpL [30][96] (columns of x1 | x2 | x3), gas4L [n <= 256] (zero except the
// selected nodes) and slotmap [n] (node -> row of gpL, -1 = not selected) -- instead of the dense slabs gp1..gp3 [N,32] and
// gas4 [N] in global memory (no zero fill of 3 x 128 B per node, no scatter, no round trip before conv4's backward)
struct TbLds { float* gpL; float* gas4L; int* slotmap; };
// HEAD = false (large batches): classifier_2 / classifier_1's backward ran batched over graphs (classifier.hip) and left
// the gradient of conv6's output in gz6g -- steps 1-3 are skipped.
// LDSOPS (the one-launch chain training kernel): ext.wf2s / ext.x4l / ext.dvl are valid -- a COMPILE-TIME promise: selecting
// between an LDS and a global pointer at run time made the loads FLAT instructions (both counters, vector-memory latency).
template <bool BIG, bool MERGED = false, bool HEAD = true, bool LDSOPS = false>
__device__ __forceinline__ void dg_tail_bwd_body(
    int b, int B, int C, const TailW& w, const int* __restrict__ graph_ptr, const int* __restrict__ perm,
    const float* __restrict__ dinv, const float* __restrict__ x4, const float* __restrict__ a5g,
    const float* __restrict__ a6g, const float* __restrict__ a1dg, const float* __restrict__ logp,
    const float* __restrict__ glogp, const int64_t* __restrict__ y, float loss_scale, int training,
    float* __restrict__ dlogit, float* __restrict__ gz1g, float* __restrict__ gz6g,
    float* __restrict__ gz5g, float* __restrict__ gp1, float* __restrict__ gp2, float* __restrict__ gp3,
    float* __restrict__ gas4, float* __restrict__ gb4p, float* __restrict__ lossv,
    float* __restrict__ ptail, const float* __restrict__ pooled, unsigned long long* dbg, TbExt ext = TbExt{},
    int* __restrict__ gpsel = nullptr, float* pacc = nullptr, bool pacc_first = true, int tid_in = -1, TbLds L = TbLds{}) {
#define TB_MARK(k) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[k] = clock64(); } while (0)
  TB_MARK(0);
  // MERGED (k_readout_tail): conv5 / conv6 weights, the pooled rows and the log-probabilities are still in the forward
  // body's LDS -- no reload, and the first barrier no longer waits for a global round trip
  __shared__ float W5s_own[MERGED ? 1 : NW5];
  __shared__ float a1ds[DGCNN_HID1];
  __shared__ float p5s[DGCNN_C5 * DGCNN_T5];
  __shared__ float sps_own[MERGED ? 1 : KCAT];            // this graph's pooled rows (conv5's weight-gradient operand)
  __shared__ float W6s_own[MERGED ? 1 : NW6];
  const float* W5s = MERGED ? ext.W5s : W5s_own;
  const float* W6s = MERGED ? ext.W6s : W6s_own;
  const float* sps = MERGED ? ext.sp : sps_own;
  __shared__ float dl[DGCNN_MAX_C];
  __shared__ float gz1s[DGCNN_HID1];
  __shared__ __attribute__((aligned(16))) float gfh[8][DGCNN_FLAT];
  __shared__ float gz6s[DGCNN_FLAT];
  __shared__ float gp5q[4][DGCNN_C5 * DGCNN_T5];
  __shared__ float gz5s[DGCNN_C5 * DGCNN_K];
  __shared__ float ga4s[DGCNN_K];
  __shared__ int selS[DGCNN_K];
  __shared__ float x4S[DGCNN_K], dvS[DGCNN_K];
  // (tid_in: a walking caller hands in an OPAQUE per-iteration copy of the thread index, so that nothing derived from it is
  //  loop-invariant for the compiler to hoist out of the walk and spill)
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // (the one-launch training kernel hands the range in: re-read behind a barrier it is a scalar-memory round trip in front of
  //  the phase's first address -- the barriers' memory clobber forbids the compiler to keep the first read)
  const int n0 = LDSOPS ? ext.n0 : graph_ptr[b], n = LDSOPS ? ext.n : graph_ptr[b + 1] - n0;
  const int msel = n < DGCNN_K ? n : DGCNN_K;

  // ---- every small global load of steps 0-2 first (VMEM loads complete in order) ----
  DgStage<NW5, RD_THREADS> st5;
  DgStage<NW6, RD_THREADS> st6;
  DgStage<KCAT, RD_THREADS> stp;
  // (a walking caller stages conv5 / conv6's weights with its FIRST graph only: pacc_first; they stay in LDS for the others)
  const bool stage_w = !pacc || pacc_first;
  if (!MERGED) { if (stage_w) { st5.load(w.W5, tid); st6.load(w.W6, tid); } stp.load(pooled + (size_t)b * KCAT, tid); }
  float lp_ = -INFINITY, g_ = 0.f;          // step 1 operands (wave 0)
  int yb_ = 0;
  if (HEAD && wv == 0) {
    if (MERGED) lp_ = lane < C ? ext.lg[lane] : -INFINITY;
    else lp_ = lane < C ? logp[(size_t)b * C + lane] : -INFINITY;
    if (glogp) g_ = lane < C ? glogp[(size_t)b * C + lane] : 0.f;
    else yb_ = MERGED ? ext.yb : (int)y[b];
  }
  // operands of the later steps that live in global memory: loaded NOW (their round trips overlap steps 1-2)
  float a6_ = 0.f, a5a_ = 0.f, a5b_ = 0.f;
  if (tid < DGCNN_FLAT)                                                                // step 3: ReLU mask of conv6
    a6_ = !HEAD ? gz6g[(size_t)b * DGCNN_FLAT + tid] : MERGED ? 0.f : a6g[(size_t)b * DGCNN_FLAT + tid];      // (HEAD = false: the finished gradient)
  // (MERGED: operands the forward half left in LDS are read WHERE THEY ARE USED, not here: held from the prologue they were ~16
  //  registers live across classifier_1's 64-register prefetch, and the kernel sits at its 128-register limit)
  if (tid < DGCNN_C5 * DGCNN_T5) {                                                     // step 5: MaxPool argmax
    const int c = tid / DGCNN_T5, u = tid - c * DGCNN_T5;
    const size_t base = (size_t)b * (DGCNN_C5 * DGCNN_K) + c * DGCNN_K + 2 * u;
    if (!MERGED) { a5a_ = a5g[base]; a5b_ = a5g[base + 1]; }
  }
  int node_ = -1;                                                                      // step 6: scatter targets
  if (tid >= 64 && tid < 64 + DGCNN_K) {
    if (MERGED) { const int ls = ext.sel[tid - 64]; node_ = ls >= 0 ? n0 + ls : -1; }
    else node_ = perm[b * DGCNN_K + (tid - 64)];
  }
  // MERGED with LDS copies of conv4's outputs / dinv (the one-launch training kernel): NO small global load is issued in this
  // body at all -- the only vector-memory loads in flight are classifier_1's rows below.  (A small load issued before them and
  // first used after them cannot be waited for precisely: the big loads sit behind a divergent `if`, the compiler cannot count
  // them and emits vmcnt(0) -- the whole 180 KB awaited a step early; issuing them from all 1024 threads instead made them
  // countable but cost 80 more wave-level load instructions at ~14 cycles of the CU's address path each.)
  constexpr bool lds_ops = MERGED && LDSOPS;
  float x4e_ = 0.f, dve_ = 0.f;
  (void)x4e_; (void)dve_;
  float a1_ = 0.f, wf2_[8];                 // step 2 operands (threads 0..127); classes beyond 8 are read in place
#pragma unroll
  for (int c = 0; c < 8; ++c) wf2_[c] = 0.f;
  if (HEAD && tid < DGCNN_HID1) {
    if (!MERGED) a1_ = a1dg[(size_t)b * DGCNN_HID1 + tid];
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < C && !lds_ops) wf2_[c] = w.Wf2[c * DGCNN_HID1 + tid];
  }
  // ---- then the big one: classifier_1's weights for step 3 (this thread's column m, 64 rows; 180 KB per
  // workgroup, rewritten by the optimizer every step).  Issued LAST and consumed in step 3; in between only
  // LDS-only barriers and no further global load, so they land while steps 1-2 run. ----
  // 704 threads = 8 row groups (16 rows each) x 88 column quads: 16 x 16-byte loads per thread (one quarter of the
  // vector-memory instructions a dword-per-lane mapping needs for the same 180 KB)
  float4 wpre[BIG ? 1 : 16];
  if (HEAD && !BIG && ext.wpre) {
#pragma unroll
    for (int j = 0; j < (BIG ? 1 : 16); ++j) wpre[j] = ext.wpre[j];
  } else if (HEAD && !BIG && tid < 2 * DGCNN_FLAT) {
    const int rg = tid / 88, mq = tid - rg * 88;
    const float* wc = w.Wf1 + (size_t)(rg * 16) * DGCNN_FLAT + 4 * mq;
#pragma unroll
    for (int j = 0; j < (BIG ? 1 : 16); ++j) wpre[j] = *reinterpret_cast<const float4*>(wc + (size_t)j * DGCNN_FLAT);
  }
  if (!MERGED) { if (stage_w) { st5.store(W5s_own, tid); st6.store(W6s_own, tid); } stp.store(sps_own, tid); }     // (waits only for the small loads above)
  // clear this graph's rows of the dense SortPooling-gradient slabs (scatter comes after barriers)
  // gpsel (large batches whose GCN backward is the chain kernels): instead of the zero rows -- 3 x 128 B per node, 57 MB of the
  // launch's 150 MB of HBM writes at 2048 COLLAB graphs -- one flag word per node; the rows of the <= 30 selected nodes are
  // written by the scatter below, the consumers skip the others
  if (L.gpL) {
    if (tid < 256) { L.gas4L[tid] = 0.f; L.slotmap[tid] = -1; }
  } else if (gpsel) {
    for (int t = tid; t < n; t += RD_THREADS) gpsel[n0 + t] = 0;
  } else {
    for (int t = tid; t < n * 32; t += RD_THREADS) {
      gp1[(size_t)n0 * 32 + t] = 0.f; gp2[(size_t)n0 * 32 + t] = 0.f; gp3[(size_t)n0 * 32 + t] = 0.f;
    }
  }
  if (!L.gpL) for (int t = tid; t < n; t += RD_THREADS) gas4[n0 + t] = 0.f;
  if (tid < DGCNN_K) ga4s[tid] = 0.f;

  if (HEAD) {
  // 1. d(loss)/d(logits) from the upstream gradient wrt log-probs (or from labels: NLL mean)
  if (wv == 0) {
    const float lp = lp_;
    float g = g_;
    if (!glogp) {
      const float sc = loss_scale != 0.f ? loss_scale : 1.0f / (float)B;
      // a label outside [0, C) (the reference's NLLLoss raises for it): clamp for memory safety and poison this
      // graph's loss with NaN, which sticks in the metrics accumulator until Trainer.read_metrics raises
      const bool ybad = (unsigned)yb_ >= (unsigned)C;
      const int yb = ybad ? 0 : yb_;
      g = (lane == yb) ? -sc : 0.f;
      // loss and accuracy bookkeeping (train.py:44-45): first max index like torch.argmax
      float mx = lp;
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const unsigned long long ball = __ballot(lane < C && lp == mx);
      const int am = __ffsll((long long)ball) - 1;
      const float lpy = __shfl(lp, yb);
      if (lane == 0) { lossv[2 * b] = ybad ? __builtin_nanf("") : -lpy * sc; lossv[2 * b + 1] = (am == yb) ? 1.f : 0.f; }
    }
    const float sg = dg_wave_sum(g);
    const float d = lane < C ? g - expf(lp) * sg : 0.f;
    if (lane < C) { dl[lane] = d; dlogit[(size_t)b * C + lane] = d; }
  }
  }
  dg_lds_barrier();
  TB_MARK(1);
  float x4n_ = 0.f, dvn_ = 0.f;             // conv4 output / dst scale of the selected nodes (wave 1; used in step 6)
  if (lds_ops) {
    if (tid >= 64 && tid < 64 + DGCNN_K && node_ >= 0) { const int ls = node_ - n0; x4n_ = ext.x4l[ls]; dvn_ = ext.dvl[ls]; }
  }
  else if (node_ >= 0) { x4n_ = x4[node_]; dvn_ = dinv[node_]; }
  // 2. through classifier_2, dropout, ReLU
  if (HEAD && tid < DGCNN_HID1) {
    float ga = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < C) ga = fmaf(dl[c], lds_ops ? ext.wf2s[c * DGCNN_HID1 + tid] : wf2_[c], ga);
    for (int c = 8; c < C; ++c) ga = fmaf(dl[c], w.Wf2[c * DGCNN_HID1 + tid], ga);
    const float a = MERGED ? ext.a1s[tid] : a1_;
    const float gz = (a != 0.f) ? (training ? ga * 2.0f : ga) : 0.f;
    gz1s[tid] = gz;
    a1ds[tid] = a;
    gz1g[(size_t)b * DGCNN_HID1 + tid] = gz;
  }
  if (HEAD) dg_lds_barrier();
  // per-graph partial of classifier_2's weight gradient: dl[c] * a1d[j]  (and bias = dl[c])
  float* pt = ptail + (size_t)b * DG_PTAIL(C);
  // pacc (large batches, a workgroup walks several graphs): the conv5 / conv6 weight-gradient partials are ACCUMULATED in an LDS
  // row of the workgroup instead of stored per graph (every element is owned by the same lane for every graph: no race, and
  // the graphs are added in their order)
  auto put = [&](int idx, float v) {
    if (pacc) pacc[idx] = pacc_first ? v : pacc[idx] + v;
    else pt[idx] = v;
  };
  if (HEAD) {
    for (int t = tid; t < C * DGCNN_HID1; t += RD_THREADS) {
      const int c = t / DGCNN_HID1, j = t - c * DGCNN_HID1;
      pt[DG_PT_WF2 + t] = dl[c] * a1ds[j];
    }
    if (tid < C) pt[DG_PT_WF2 + C * DGCNN_HID1 + tid] = dl[tid];
  }
  TB_MARK(2);
  // 3. through classifier_1: 352 outputs x 128 terms, split in two halves of 64 terms (704 threads); the
  //    weights were prefetched into registers at kernel start
  if (HEAD && tid < 2 * DGCNN_FLAT) {
    const int rg = tid / 88, mq = tid - rg * 88;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BIG) {
      const float* wc = w.Wf1 + (size_t)(rg * 16) * DGCNN_FLAT + 4 * mq;
#pragma unroll 1
      for (int j0 = 0; j0 < 16; j0 += 4) {
        float4 wq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wq[u] = *reinterpret_cast<const float4*>(wc + (size_t)(j0 + u) * DGCNN_FLAT);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float z = gz1s[rg * 16 + j0 + u];
          g.x = fmaf(z, wq[u].x, g.x); g.y = fmaf(z, wq[u].y, g.y);
          g.z = fmaf(z, wq[u].z, g.z); g.w = fmaf(z, wq[u].w, g.w);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < (BIG ? 1 : 16); ++j) {
        const float z = gz1s[rg * 16 + j];
        g.x = fmaf(z, wpre[j].x, g.x); g.y = fmaf(z, wpre[j].y, g.y);
        g.z = fmaf(z, wpre[j].z, g.z); g.w = fmaf(z, wpre[j].w, g.w);
      }
    }
    *reinterpret_cast<float4*>(&gfh[rg][4 * mq]) = g;
  }
  if (HEAD) __syncthreads();
  if (HEAD && tid < DGCNN_FLAT) {     // ... and the ReLU after conv6 ; the 8 row-group partials in a fixed order
    const float gf = ((gfh[0][tid] + gfh[1][tid]) + (gfh[2][tid] + gfh[3][tid])) +
                     ((gfh[4][tid] + gfh[5][tid]) + (gfh[6][tid] + gfh[7][tid]));
    const float g6 = ((MERGED && HEAD) ? ext.flat[tid] : a6_) > 0.f ? gf : 0.f;
    gz6s[tid] = g6;
    gz6g[(size_t)b * DGCNN_FLAT + tid] = g6;
  }
  if (!HEAD && tid < DGCNN_FLAT) gz6s[tid] = a6_;
  __syncthreads();
  TB_MARK(3);
  // 4. conv6 data gradient on the matrix cores: gp5[c][u] = sum_{oc,d} W6[oc][c][d] gz6[oc][u-d]
  //    -> [16 x 16(15)] = [16 x 160] . [160 x 16]; K split over 4 waves (40 each), combined in a fixed order
  if (wv < 4) {
    // (by hand: wave wv takes output channels oc = 8 wv .. 8 wv + 7 of conv6, k-slot of (step u, lane group kq) = (oc = 8 wv + 2 kq
    //  + u / 5, d = u % 5): affine in kq, no division by 5 per operand; two accumulator chains)
    const int mi = lane & 15, kq = lane >> 4;
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    float av[10], bv[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int oc = 8 * wv + 2 * kq + u / DGCNN_KW6, dd = u % DGCNN_KW6;
      av[u] = W6s[(oc * DGCNN_C5 + mi) * DGCNN_KW6 + dd];                  // A[c = mi][k]
      const int tt = mi - dd;
      const bool okb = mi < DGCNN_T5 && tt >= 0 && tt < DGCNN_T6;
      const float g_ = gz6s[oc * DGCNN_T6 + (okb ? tt : 0)];               // B[k][u = mi]
      bv[u] = okb ? g_ : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      if (u & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d1, 0, 0, 0);
      else d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d0, 0, 0, 0);
    }
    if (mi < DGCNN_T5) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gp5q[wv][(4 * kq + r) * DGCNN_T5 + mi] = d0[r] + d1[r];
    }
  }
  __syncthreads();
  TB_MARK(4);      // (the four K-split partial tiles are combined by their only consumer, step 5: one phase and one barrier fewer)
  // 5. MaxPool (first max wins ties, like ATen) + ReLU after conv5 -> [16,30]
  if (tid < DGCNN_C5 * DGCNN_T5) {
    const int c = tid / DGCNN_T5, u = tid - c * DGCNN_T5;
    const size_t base = (size_t)b * (DGCNN_C5 * DGCNN_K) + c * DGCNN_K + 2 * u;
    const float a0 = MERGED ? ext.a5s[c * DGCNN_K + 2 * u] : a5a_, a1 = MERGED ? ext.a5s[c * DGCNN_K + 2 * u + 1] : a5b_;
    p5s[tid] = fmaxf(a0, a1);                      // MaxPool1d output, needed for conv6's weight gradient
    const float gp = (gp5q[0][tid] + gp5q[1][tid]) + (gp5q[2][tid] + gp5q[3][tid]);
    const bool first = !(a1 > a0);
    const float g0 = (first && a0 > 0.f) ? gp : 0.f;
    const float g1 = (!first && a1 > 0.f) ? gp : 0.f;
    gz5s[c * DGCNN_K + 2 * u] = g0;
    gz5s[c * DGCNN_K + 2 * u + 1] = g1;
    gz5g[base] = g0;
    gz5g[base + 1] = g1;
  }
  if (tid >= 64 && tid < 64 + DGCNN_K) { selS[tid - 64] = node_; x4S[tid - 64] = x4n_; dvS[tid - 64] = dvn_; }
  __syncthreads();
  TB_MARK(5);
  // 5b. per-graph partial weight gradients of conv6 and conv5 (everything they need is in LDS / this graph's
  //     pooled rows); k_wgrad then only sums B contiguous partials per element (coalesced)
  // conv6: pW6[oc][(c,d)] = sum_t gz6[oc][t] p5[c][t+d]   -> [32 x 80] = [32 x 12(11)] . [12 x 80]   (10 tiles)
  // conv5: pW5[o][m]      = sum_s gz5[o][s] sp[s][m]       -> [16 x 112(97)] = [16 x 32(30)] . [32 x 112] (7 tiles)
  // (written out per product: the generic tile helper's per-operand bounds tests and index arithmetic were most of these
  //  phases' instructions, and every wave of the workgroup executes them)
  // (5b and 6 form ONE list of 17 + 14 = 31 matrix-core jobs dealt round robin: wave w takes jobs w and w + 16 -- two at most.  As two
  //  loops, wave 0 carried three (jobs 0 and 16 of 5b, job 0 of 6) and was the phase's duration: every other wave waited at the
  //  barrier behind it.)
  for (int job = wv; job < 17; job += RD_THREADS / 64) {
    const int mi = lane & 15, kq = lane >> 4;
    if (job < 10) {
      const int mt = job / 5, nt = job - mt * 5;
      const int n = nt * 16 + mi;                                    // column (c, d) of W6 [32][80]
      const float* ap = gz6s + (mt * 16 + mi) * DGCNN_T6 + kq;       // A[oc][t], t = 4 u + kq
      const float* bp = p5s + (n / DGCNN_KW6) * DGCNN_T5 + (n % DGCNN_KW6) + kq;      // B[t][n] = p5[c][t + d]
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
      float av[3], bv[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const bool tok = u < 2 || kq < DGCNN_T6 - 8;                 // t = 8 + kq < 11
        const float a_ = ap[tok ? 4 * u : 0], b_ = bp[tok ? 4 * u : 0];
        av[u] = tok ? a_ : 0.f; bv[u] = tok ? b_ : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) put(DG_PT_W6 + (mt * 16 + 4 * kq + r) * (DGCNN_C5 * DGCNN_KW6) + n, d[r]);
    } else {
      const int nt = job - 10;
      const int m = nt * 16 + mi;                                    // column of W5 [16][97]
      const bool mok = m < DGCNN_CAT;
      const float* ap = gz5s + mi * DGCNN_K + kq;                    // A[o][sl], sl = 4 u + kq
      const float* bp = sps + kq * DGCNN_CAT + (mok ? m : 0);        // B[sl][m]
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool sok = u < 7 || kq < DGCNN_K - 28;                 // sl = 28 + kq < 30
        const float a_ = ap[sok ? 4 * u : 0], b_ = bp[sok ? 4 * u * DGCNN_CAT : 0];
        av[u] = sok ? a_ : 0.f; bv[u] = (sok && mok) ? b_ : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d1, 0, 0, 0);
        else d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d0, 0, 0, 0);
      }
      if (mok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) put(DG_PT_W5 + (4 * kq + r) * DGCNN_CAT + m, d0[r] + d1[r]);
      }
    }
  }
  if (tid < DGCNN_C6) {
    float acc = 0.f;
#pragma unroll
    for (int tt = 0; tt < DGCNN_T6; ++tt) acc += gz6s[tid * DGCNN_T6 + tt];
    put(DG_PT_B6 + tid, acc);
  }
  if (tid >= 512 && tid < 512 + DGCNN_C5) {
    const int o = tid - 512;
    float acc = 0.f;
    for (int sl = 0; sl < DGCNN_K; ++sl) acc += gz5s[o * DGCNN_K + sl];
    put(DG_PT_B5 + o, acc);
  }
  TB_MARK(6);
  // 6. conv5 data gradient = gradient wrt the pooled rows; scatter to the selected nodes
  // gsp[s][c] = sum_oc gz5[oc][s] W5[oc][c]  -> [32(30) x 112(97)] = [32 x 16] . [16 x 112]  (14 tiles, one wave each)
  for (int job = (wv == 0 ? 16 : wv - 1); job < 14; job += RD_THREADS / 64) {      // (list position 17 + job = wave (job + 1) & 15's second job)
    const int mt = job / 7, nt = job - mt * 7;
    const int mi = lane & 15, kq = lane >> 4;
    const int srow = mt * 16 + mi, ccol = nt * 16 + mi;
    const bool sok = srow < DGCNN_K, cok = ccol < DGCNN_CAT;
    const float* ap = gz5s + kq * DGCNN_K + (sok ? srow : 0);        // A[s][oc] = gz5[oc][s], oc = 4 u + kq
    const float* bp = W5s + kq * DGCNN_CAT + (cok ? ccol : 0);       // B[oc][c] = W5[oc][c]
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a_ = ap[4 * u * DGCNN_K], b_ = bp[4 * u * DGCNN_CAT];
      av[u] = sok ? a_ : 0.f; bv[u] = cok ? b_ : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], d, 0, 0, 0);
    auto st = [&](int sl, int c, float v) {
          if (sl < msel && c < DGCNN_CAT) {
            const int node = selS[sl];
            if (L.gpL) {
              if (c < 96) L.gpL[sl * 96 + c] = v;
              else {
                const float xv = x4S[sl];
                const float ga = v * (1.f - xv * xv);      // tanh'
                L.gas4L[node - n0] = dvS[sl] * ga;
                L.slotmap[node - n0] = sl;
                ga4s[sl] = ga;
              }
              return;
            }
            if (gpsel && c == 0) gpsel[node] = 1;      // (behind several barriers of this workgroup: after the clearing store)
            if (c < 32) gp1[(size_t)node * 32 + c] = v;
            else if (c < 64) gp2[(size_t)node * 32 + c - 32] = v;
            else if (c < 96) gp3[(size_t)node * 32 + c - 64] = v;
            else {
              const float xv = x4S[sl];
              const float ga = v * (1.f - xv * xv);      // tanh'
              gas4[node] = dvS[sl] * ga;
              ga4s[sl] = ga;
            }
          }
        };
#pragma unroll
    for (int r = 0; r < 4; ++r) st(mt * 16 + 4 * kq + r, ccol, d[r]);
  }
  __syncthreads();
  TB_MARK(7);
  if (tid == 0) {     // db4 partial of this graph, fixed order
    float sum = 0.f;
    for (int s = 0; s < DGCNN_K; ++s) sum += ga4s[s];
    gb4p[b] = sum;
  }
}

