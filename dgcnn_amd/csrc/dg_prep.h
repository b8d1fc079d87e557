// dg_prep.h -- fast-path graph preparation bodies (see prep.hip for the algorithm notes), shared by the
// stand-alone kernels k_prep_fast_a / k_prep_fast_b and by the RIDER block ranges of k_readout_fwd /
// k_tail_bwd: those two launches run only B (= 50) workgroups, so ~80% of the chip idles for 10-15 us
// each; the pipelined training step uses that idle capacity to prepare the NEXT batch's graph structure
// (extra workgroups appended to the grid; same stream, so ordering is trivially safe).
#pragma once
#include "dg_common.h"

struct DgPrepRider {
  const int64_t* ei; const int64_t* batch;
  int E, N, B;
  int *rowptr, *colidx, *rowptr_t, *colidx_t, *graph_ptr, *graph_eptr;
  float* dinv;
  const float* x; float* xs; int F;      // aggregate-first conv1: xs[i] = dinv[i]*x[i] ([N,F]); x == nullptr: none
  unsigned int* err;
  unsigned int epoch;
  int nblk;        // rider workgroups appended to the host kernel's grid (0 = none)
};
static inline int dg_prep_fast_work(int E, int N, int B) {
  int work = E > N + 1 ? E : N + 1;
  return B + 1 > work ? B + 1 : work;
}

#ifdef __HIPCC__
// Kernel-A body, thread t of max(E, N+1, B+1): range / self-loop / strict (src,dst) order checks, colidx copies,
// rowptr by ROW-BOUNDARY detection, graph_ptr by binary search on the sorted batch vector.
__device__ __forceinline__ void dg_prep_fast_a_body(int t, const int64_t* __restrict__ ei, int E, int N,
                                                    const int64_t* __restrict__ batch, int B, int* __restrict__ rowptr,
                                                    int* __restrict__ colidx, int* __restrict__ rowptr_t,
                                                    int* __restrict__ colidx_t, int* __restrict__ graph_ptr,
                                                    unsigned int* __restrict__ err, unsigned int epoch) {
  const int64_t* src = ei;
  const int64_t* dst = ei + E;
  if (t < E) {
    const int64_t s = src[t], d = dst[t];
    const bool range = (uint64_t)s >= (uint64_t)N || (uint64_t)d >= (uint64_t)N;
    bool bad = range || s == d;
    int64_t ps = -1;
    if (t > 0) {
      ps = src[t - 1];
      const int64_t pd = dst[t - 1];
      bad = bad || !(ps < s || (ps == s && pd < d));
    }
    if (bad) { err[range ? 0 : 1] = epoch; err[range ? 2 : 3] = ~epoch; }
    // memory safety even when the promise is broken: clamp everything that later indexes memory, so a
    // flagged batch yields garbage numbers but never an out-of-bounds access
    const int dc = (uint64_t)d < (uint64_t)N ? (int)d : 0;
    const int sc = s < 0 ? 0 : (s >= N ? N - 1 : (int)s);
    const int pc = ps < 0 ? -1 : (ps >= N ? N - 1 : (int)ps);
    colidx[t] = dc;
    colidx_t[t] = dc;
    if (pc < sc || t == 0)
      for (int k = pc + 1; k <= sc; ++k) { rowptr[k] = t; rowptr_t[k] = t; }
    if (t == E - 1)
      for (int k = sc + 1; k <= N; ++k) { rowptr[k] = E; rowptr_t[k] = E; }
  }
  if (t <= B) {
    int lo = 0, hi = N;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (batch[mid] < (int64_t)t) lo = mid + 1; else hi = mid;
    }
    graph_ptr[t] = lo;
  }
}
// Kernel-B body: dinv per node, graph_eptr, and (per edge (s,d)) the reverse edge (d,s) must be in row d --
// binary search inside that row only (<= log2(deg) steps).  Needs kernel A's outputs complete.
__device__ __forceinline__ void dg_prep_fast_b_body(int t, const int64_t* __restrict__ ei, int E, int N, int B,
                                                    const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                    const int* __restrict__ graph_ptr, int* __restrict__ graph_eptr,
                                                    float* __restrict__ dinv, unsigned int* __restrict__ err,
                                                    unsigned int epoch, const float* __restrict__ x = nullptr,
                                                    float* __restrict__ xs = nullptr, int F = 0) {
  if (t < N) {
    const float di = 1.0f / sqrtf((float)(rowptr[t + 1] - rowptr[t] + 1));
    dinv[t] = di;
    if (x) {      // pre-scaled raw features for the aggregate-first conv1 gather (one row load per edge, no dinv[j] load)
      for (int f = 0; f < F; ++f) xs[(size_t)t * F + f] = di * x[(size_t)t * F + f];
    }
  }
  if (t <= B) graph_eptr[t] = rowptr[graph_ptr[t]];      // first edge position of each graph's rows
  if (t < E) {
    const int64_t s = ei[t], d = ei[(int64_t)E + t];
    if ((uint64_t)s < (uint64_t)N && (uint64_t)d < (uint64_t)N) {
      const int end = rowptr[d + 1];
      int a = rowptr[d], b = end;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (colidx[mid] < (int)s) a = mid + 1; else b = mid;
      }
      if (!(a < end && colidx[a] == (int)s)) { err[1] = epoch; err[3] = ~epoch; }
    }
  }
}
#endif
