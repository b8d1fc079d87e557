// classifier.hip -- classifier_1 / classifier_2 (reference model.py:21-23, 41-45: Linear(352,128) + ReLU + Dropout(0.5),
// Linear(128,C), log_softmax) and their backward for LARGE batches, 16 graphs per workgroup.
//
// Why a second form: the readout kernels (tail.hip) give every graph its own workgroup -- right at the reference's batch
// of 50, where a step is one latency chain per graph.  With thousands of graphs per launch every one of those workgroups
// streams classifier_1's 180 KB through its CU's vector-memory path, forward and again backward: 2048 graphs x 360 KB =
// 0.74 GB of L2 -> CU traffic for 0.37 GFLOP, 5.9 k + 5.6 k of the ~20 k cycles a graph spends in each readout kernel
// (phase clocks, 2048 COLLAB graphs).  Across graphs the two layers are plain GEMMs, [B,352].[352,128] and back
// [B,128].[128,352]: here a workgroup takes 16 graphs, reads the weights ONCE for all of them (2 x 180 KB per 16 graphs)
// and runs both products on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, M = the 16 graphs).  The readout kernels
// then stop at conv6's output (k_readout_fwd<BIG, HEAD = false>) and start at its gradient (k_tail_bwd<BIG, HEAD = false>).
//
// Same arithmetic as the per-graph form up to fp32 summation order (a dot product's terms are added in matrix-core order);
// the dropout mask is the same function of (seed, graph, unit), the loss / accuracy bookkeeping and classifier_2's
// per-graph weight-gradient partials have the same layout -- k_wgrad does not know which form ran.
#include "dg_common.h"
#include "dg_readout.h"

#define CL_GB 16                       // graphs per workgroup (the M of the matrix instruction)
#define CL_THREADS 512                 // 8 waves: one 16-unit column tile of classifier_1 each
#define CL_FS (DGCNN_FLAT + 4)         // row strides of the LDS tiles: 16-byte reads of 16 rows fall on distinct banks
#define CL_HS (DGCNN_HID1 + 4)
#define CL_WS (DGCNN_FLAT / 2 + 4)     // row stride of a staged half of classifier_1's weights

// MODE 0: forward only (inference, and training steps whose backward starts from an upstream gradient);
// MODE 1: forward + backward from labels (NLL mean, train.py:40-45).
template <int MODE>
__global__ void __launch_bounds__(CL_THREADS)
k_classifier(int B, int C, TailW w, const float* __restrict__ a6g, float* __restrict__ a1dg, uint8_t* __restrict__ maskg,
             float* __restrict__ logp, int training, uint64_t seed, const int64_t* __restrict__ y, float loss_scale,
             float* __restrict__ dlogit, float* __restrict__ gz1g, float* __restrict__ gz6g, float* __restrict__ lossv,
             float* __restrict__ ptail, unsigned long long* dbg) {
  // phase stamps: measurement builds only (-DCL_TIMING, tools/phase_classifier.py) -- they use slots 32..42 of the debug
  // buffer, beyond the 16 words dgcnn_debug_phase_clocks promises to touch
#ifdef CL_TIMING
#define CL_MARK(k) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[32 + (k)] = clock64(); } while (0)
#else
#define CL_MARK(k) do { } while (0)
#endif
  CL_MARK(0);
#ifdef CL_TIMING
  if (dbg && threadIdx.x == 0) atomicMin(&dbg[41], wall_clock64());
#endif
  // every kernel argument is brought into scalar registers NOW (an empty use): left to the compiler, the output pointers were
  // loaded where they are first used -- scalar-cache round trips with `s_waitcnt lgkmcnt(0)` between the phases of a workgroup
  // that is one latency chain
#ifndef DG_EMU
  asm volatile("" :: "s"(a1dg), "s"(maskg), "s"(logp), "s"(y), "s"(dlogit), "s"(gz1g), "s"(gz6g), "s"(lossv), "s"(ptail), "s"(seed),
               "s"(training), "s"(loss_scale), "s"(w.Wf2), "s"(w.bf1), "s"(w.bf2));
#endif
  __shared__ __attribute__((aligned(16))) float fl[CL_GB * CL_FS];       // conv6 outputs of the 16 graphs (ReLU mask of the way back)
  __shared__ __attribute__((aligned(16))) float a1s[CL_GB * CL_HS];      // classifier_1 outputs after ReLU / dropout
  __shared__ __attribute__((aligned(16))) float gz1s[CL_GB * CL_HS];     // gradient wrt classifier_1's pre-activation
  __shared__ float lg[CL_GB * DGCNN_MAX_C];                              // logits, then log-probabilities
  __shared__ float dl[CL_GB * DGCNN_MAX_C];                              // gradient wrt the logits
  // every small operand is brought into LDS by the kernel's first loads: a workgroup is ONE latency chain (128 workgroups on
  // 128 CUs at 2048 graphs, nothing else on the CU to hide a round trip), and with a cold global load in front of every
  // phase the first version spent 25 us here -- a dozen dependent round trips -- for 3 us of matrix instructions
  // classifier_1's weights pass through LDS in two halves of 176 columns ([128 units][176 + 4]): straight from global into the
  // B operand, lane (unit nl, kq) reads 16 bytes of ITS unit's row and the 16 lanes of a quad touch 16 different cache lines
  // -- 11.5 k tag lookups for the 180 KB, 7 us of the first version's 23.  Staged, a row's 704 bytes are read by 44
  // consecutive lanes.  classifier_2's weights take the buffer over once the forward product is done.
  __shared__ __attribute__((aligned(16))) float wc[DGCNN_HID1 * CL_WS];
  float* w2s = wc;
  __shared__ float b1s[DGCNN_HID1], b2s[DGCNN_MAX_C];
  __shared__ int ys[CL_GB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nl = lane & 15, kq = lane >> 4;
  const int b0 = blockIdx.x * CL_GB, nb = min(CL_GB, B - b0);
  // (workgroups walk the shared weights from different starting tiles: all of them asking one L2 channel for the same
  //  line at the same moment is the slow way to read 180 KB)
  // (consecutive workgroups land on different XCDs, each with its own L2: the ones that share an L2 are blockIdx.x >> 3 apart)
  const int xr = (int)blockIdx.x >> 3;
  const int ut = (wv + xr) & 7;                                          // this wave's 16-unit tile of classifier_1

  constexpr int NQ = CL_GB * (DGCNN_FLAT / 4), NI = (NQ + CL_THREADS - 1) / CL_THREADS;      // 16-byte pieces of the graphs' rows
  constexpr int HQ = DGCNN_FLAT / 8;                                                           // 16-byte pieces per unit row and half (44)
  constexpr int NH = DGCNN_HID1 * HQ / CL_THREADS;                                             // ... per thread and half (11)
  static_assert(DGCNN_HID1 * HQ % CL_THREADS == 0, "a half of classifier_1 is a whole number of pieces per thread");
  constexpr int NW2 = DGCNN_MAX_C * DGCNN_HID1 / CL_THREADS;
  f32x4 wh[NH], wh1[NH];       // (a native vector type: an array of HIP's float4 structs written in two places stayed in scratch)
  // (macros, not lambdas: captured by reference the register array stayed an alloca -- 192 bytes of scratch per lane)
#define CL_LOAD_HALF(h, wh)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < NH; ++i) {                                                                   \
    const int idx = tid + CL_THREADS * i, u = idx / HQ, q = idx - u * HQ;                                            \
    wh[i] = *reinterpret_cast<const f32x4*>(w.Wf1 + (size_t)u * DGCNN_FLAT + (DGCNN_FLAT / 2) * (h) + 4 * q);      \
  }
#define CL_STORE_HALF(wh)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < NH; ++i) {                                                                   \
    const int idx = tid + CL_THREADS * i, u = idx / HQ, q = idx - u * HQ;                                            \
    *reinterpret_cast<f32x4*>(wc + u * CL_WS + 4 * q) = wh[i];                                                       \
  }
  {   // EVERY load of the set-up is in flight before the first LDS store: a load followed by its own LDS store is a round
      // trip of its own (loads complete in order)
    CL_LOAD_HALF(0, wh)
    CL_LOAD_HALF(1, wh1)                             // (both halves requested at once: one cold round trip, not two)
    float smallv = 0.f;
    int yv = 0;
    if (tid < DGCNN_HID1) smallv = w.bf1[tid];
    else if (tid < DGCNN_HID1 + C) smallv = w.bf2[tid - DGCNN_HID1];
    else if (MODE == 1 && tid >= 256 && tid < 256 + nb) yv = (int)y[b0 + tid - 256];
    float4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {      // the 16 graphs' conv6 outputs
      const int t = tid + CL_THREADS * i, g = t / (DGCNN_FLAT / 4), q = t - g * (DGCNN_FLAT / 4);
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < NQ && g < nb) v[i] = *reinterpret_cast<const float4*>(a6g + (size_t)(b0 + g) * DGCNN_FLAT + 4 * q);
    }
    if (tid < DGCNN_HID1) b1s[tid] = smallv;
    else if (tid < DGCNN_HID1 + C) b2s[tid - DGCNN_HID1] = smallv;
    else if (MODE == 1 && tid >= 256 && tid < 256 + CL_GB) ys[tid - 256] = yv;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int t = tid + CL_THREADS * i, g = t / (DGCNN_FLAT / 4), q = t - g * (DGCNN_FLAT / 4);
      if (t < NQ) *reinterpret_cast<float4*>(fl + g * CL_FS + 4 * q) = v[i];
    }
    CL_STORE_HALF(wh)
  }
  __syncthreads();
  CL_MARK(1);
  // ---- classifier_1: z[g][u] = sum_k flat[g][k] W1[u][k]; ReLU; Dropout(0.5) -------------------------------------------
  // lane (nl, kq): A = flat[graph nl][16j + 4kq + s], B = W1[unit 16 ut + nl][same k] for matrix step 4j + s of a half
  float w2r[NW2];
  {
    // TWO accumulators (the x/z and the y/w k-slots): a chain of 88 dependent matrix instructions ran at half the pipe's rate
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    auto half_product = [&](int h) __attribute__((always_inline)) {
      const float* ar = fl + nl * CL_FS + (DGCNN_FLAT / 2) * h + 4 * kq;
      const float* br = wc + (16 * ut + nl) * CL_WS + 4 * kq;
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(ar + 16 * j);
        const float4 bq = *reinterpret_cast<const float4*>(br + 16 * j);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq.x, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq.y, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq.z, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq.w, acc2, 0, 0, 0);
      }
    };
#pragma unroll
    for (int i = 0; i < NW2; ++i) w2r[i] = (tid + CL_THREADS * i < C * DGCNN_HID1) ? w.Wf2[tid + CL_THREADS * i] : 0.f;
    half_product(0);
    dg_lds_barrier();                                // every wave has read the first half
    CL_STORE_HALF(wh1)
    dg_lds_barrier();
    half_product(1);
    acc += acc2;
    const int u = 16 * ut + nl;                      // this lane: unit u of graphs 4kq .. 4kq + 3
    const float bu = b1s[u];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int g = 4 * kq + r;
      float av = fmaxf(acc[r] + bu, 0.f);
      uint8_t keep = 1;
      if (training) {
        keep = dg_keep(seed, (uint64_t)(b0 + g) * DGCNN_HID1 + u) ? 1 : 0;
        av = keep ? av * 2.0f : 0.f;                 // p = 0.5 -> scale 1/(1-p) = 2
      }
      a1s[g * CL_HS + u] = av;
      if (g < nb) { a1dg[(size_t)(b0 + g) * DGCNN_HID1 + u] = av; maskg[(size_t)(b0 + g) * DGCNN_HID1 + u] = keep; }
    }
  }
  // the way back through classifier_1 reads the same 180 KB by columns: this wave's (up to three) column tiles are requested
  // NOW, into the registers the forward's rows just left, and land while classifier_2 / log_softmax / their backward run.
  // Lane (m = nl, kq) of tile t holds W1[16i + 4kq + s][16t + nl], the k-slot of matrix step 4i + s (64-byte row segments)
  float bw[3][MODE == 1 ? 32 : 1];
  int bt[3];
  if (MODE == 1) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int tt = wv + 8 * q;
      bt[q] = tt < 22 ? (tt + 3 * xr) % 22 : -1;
      const float* wc = w.Wf1 + (size_t)(4 * kq) * DGCNN_FLAT + 16 * max(bt[q], 0) + nl;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) bw[q][4 * i + s] = wc[(size_t)(16 * i + s) * DGCNN_FLAT];
    }
  }
  CL_MARK(2);
  dg_lds_barrier();                                  // every wave has read the second half: classifier_2's weights move in
#pragma unroll
  for (int i = 0; i < NW2; ++i) if (tid + CL_THREADS * i < C * DGCNN_HID1) w2s[tid + CL_THREADS * i] = w2r[i];
  dg_lds_barrier();
  CL_MARK(3);
  // ---- classifier_2: 128 -> C, eight lanes per (graph, class), fixed xor butterfly ---------------------------------------
  {
    const int p8 = tid & 7;
    for (int o = tid >> 3; o < CL_GB * C; o += CL_THREADS / 8) {
      const int g = o / C, c = o - g * C;
      const float* wr = w2s + c * DGCNN_HID1 + p8;
      const float* ar = a1s + g * CL_HS + p8;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 2) { s0 = fmaf(wr[8 * i], ar[8 * i], s0); s1 = fmaf(wr[8 * i + 8], ar[8 * i + 8], s1); }
      float tot = s0 + s1;
      tot += __shfl_xor(tot, 1);
      tot += __shfl_xor(tot, 2);
      tot += __shfl_xor(tot, 4);
      if (p8 == 0) lg[g * DGCNN_MAX_C + c] = tot + b2s[c];
    }
  }
  dg_lds_barrier();
  CL_MARK(4);
  // ---- log_softmax over C, loss / accuracy bookkeeping, d(loss)/d(logits): a graph per half wave (C <= 32: one round for
  //      the 16 graphs) or per wave (two rounds) ------------------------------------------------------------------------
  {
    const bool halves = C <= 32;
    const int hb = halves ? (lane & 32) : 0, cl = lane - hb;            // first lane of this graph's lanes, class of this lane
    for (int g = halves ? 2 * wv + (lane >> 5) : wv; g < CL_GB; g += CL_THREADS / 64) {
      const float v = cl < C ? lg[g * DGCNN_MAX_C + cl] : -INFINITY;
      float mx = v;
      for (int o = halves ? 16 : 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float e = cl < C ? expf(v - mx) : 0.f;
      for (int o = halves ? 16 : 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
      const float lp = cl < C ? (v - mx) - logf(e) : -INFINITY;
      if (g < nb && cl < C) logp[(size_t)(b0 + g) * C + cl] = lp;
      if (MODE == 1) {
        const int b = b0 + g;
        const float sc = loss_scale != 0.f ? loss_scale : 1.0f / (float)B;
        const int yraw = ys[g];
        // a label outside [0, C) (the reference's NLLLoss raises for it): clamped for memory safety, the graph's loss is NaN
        // and sticks in the metrics accumulator until Trainer.read_metrics raises
        const bool ybad = (unsigned)yraw >= (unsigned)C;
        const int yb = ybad ? 0 : yraw;
        float m2 = lp;
        for (int o = halves ? 16 : 32; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o));
        unsigned long long ball = __ballot(cl < C && lp == m2);
        if (halves) ball = (ball >> hb) & 0xffffffffull;
        const int am = __ffsll((long long)ball) - 1;       // first max index, like torch.argmax (train.py:44)
        const float lpy = __shfl(lp, hb + yb);
        if (g < nb && cl == 0) { lossv[2 * b] = ybad ? __builtin_nanf("") : -lpy * sc; lossv[2 * b + 1] = (am == yb) ? 1.f : 0.f; }
        // d(loss)/d(logit c) = g_c - softmax_c * sum(g), the upstream gradient g being -sc at the label
        const float gl = (cl == yb) ? -sc : 0.f;
        const float d = (cl < C && g < nb) ? gl + expf(lp) * sc : 0.f;
        if (cl < C) {
          dl[g * DGCNN_MAX_C + cl] = d;
          if (g < nb) dlogit[(size_t)b * C + cl] = d;
        }
        if (halves) break;
      } else if (halves) break;
    }
  }
  if (MODE == 0) return;
  dg_lds_barrier();
  CL_MARK(5);
  // ---- back through classifier_2, dropout, ReLU: gz1[g][j]; per-graph partials of classifier_2's weight gradient --------
  {
    const int j = tid & (DGCNN_HID1 - 1), g0 = tid >> 7;                 // 4 graphs per pass, 4 passes, all in flight
    float ga[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      const float wc = w2s[c * DGCNN_HID1 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i) ga[i] = fmaf(dl[(g0 + 4 * i) * DGCNN_MAX_C + c], wc, ga[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = g0 + 4 * i;
      const float a = a1s[g * CL_HS + j];
      const float gz = (a != 0.f) ? (training ? ga[i] * 2.0f : ga[i]) : 0.f;
      gz1s[g * CL_HS + j] = gz;
      if (g < nb) gz1g[(size_t)(b0 + g) * DGCNN_HID1 + j] = gz;
    }
  }
  for (int t = tid; t < C * DGCNN_HID1 + C; t += CL_THREADS) {           // element t of every graph's partial: 16 stores in flight
    const bool wpart = t < C * DGCNN_HID1;
    const int c = wpart ? t >> 7 : t - C * DGCNN_HID1, j = t & (DGCNN_HID1 - 1);
    float* pt = ptail + (size_t)b0 * DG_PTAIL(C) + DG_PT_WF2 + t;
#pragma unroll
    for (int g = 0; g < CL_GB; ++g)
      if (g < nb) pt[(size_t)g * DG_PTAIL(C)] = wpart ? dl[g * DGCNN_MAX_C + c] * a1s[g * CL_HS + j] : dl[g * DGCNN_MAX_C + c];
  }
  dg_lds_barrier();
  CL_MARK(6);
  // ---- back through classifier_1: gflat[g][m] = sum_j gz1[g][j] W1[j][m], and the ReLU after conv6 ----------------------
  {
    float4 az[8];
    const float* ar = gz1s + nl * CL_HS + 4 * kq;
#pragma unroll
    for (int i = 0; i < 8; ++i) az[i] = *reinterpret_cast<const float4*>(ar + 16 * i);
    // the wave's (up to) three column tiles advance together: three independent chains of 32 matrix instructions
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(az[i].x, bw[q][4 * i + 0], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(az[i].y, bw[q][4 * i + 1], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(az[i].z, bw[q][4 * i + 2], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(az[i].w, bw[q][4 * i + 3], acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (bt[q] >= 0) {
        const int m = 16 * bt[q] + nl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int g = 4 * kq + r;
          if (g < nb) gz6g[(size_t)(b0 + g) * DGCNN_FLAT + m] = fl[g * CL_FS + m] > 0.f ? acc[q][r] : 0.f;
        }
      }
    }
  }
  CL_MARK(7);
#ifdef CL_TIMING
  if (dbg && threadIdx.x == 0) { atomicMax(&dbg[40], wall_clock64()); if (blockIdx.x == 0) dbg[42] = wall_clock64(); }
#endif
  (void)dbg;
#undef CL_MARK
#undef CL_LOAD_HALF
#undef CL_STORE_HALF
}

// graphs per launch from which the readout pair splits the classifier off (tail.hip's two-workgroups-per-CU forms start at
// the same size)
int dg_launch_classifier(int B, int C, const float* params, const DgParams* pl, const float* a6, float* a1d, uint8_t* drop_mask,
                         float* logp, int training, uint64_t seed, const int64_t* y, float loss_scale, float* dlogit,
                         float* gz1, float* gz6, float* lossv, float* ptail, hipStream_t s) {
  if (B <= 0 || C < 1 || C > DGCNN_MAX_C) return DGCNN_EINVAL;
  const int grid = (B + CL_GB - 1) / CL_GB;
  if (y)
    hipLaunchKernelGGL(k_classifier<1>, dim3(grid), dim3(CL_THREADS), 0, s, B, C, dg_tail_w(params, pl), a6, a1d, drop_mask, logp,
                       training, seed, y, loss_scale, dlogit, gz1, gz6, lossv, ptail, dg_debug_buffer());
  else
    hipLaunchKernelGGL(k_classifier<0>, dim3(grid), dim3(CL_THREADS), 0, s, B, C, dg_tail_w(params, pl), a6, a1d, drop_mask, logp,
                       training, seed, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, dg_debug_buffer());
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}
