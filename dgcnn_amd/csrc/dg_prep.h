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
  unsigned int* bits; int* dmap;   // dense per-graph block structures (dg_dense.h); bits == nullptr: not built
};
static inline int dg_prep_fast_work(int E, int N, int B) {
  int work = E > N + 1 ? E : N + 1;
  return B + 1 > work ? B + 1 : work;
}

// ---- dense per-graph block structures (consumed by gcn_dense.hip) ----------------------------------------------
// adjacency bitmap: row i of graph g holds bit (j - n0_g) for every neighbour j and for j == i, in ceil(n_g/32) words;
// rows are stored with a power-of-two word stride S = 2^c >= ceil(n_g/32) (five classes, c = 0..4, i.e. graphs of up
// to 32/64/128/256/512 nodes); class c lives at word offset N*(2^c - 1), row i at i*2^c inside it -- every offset is
// a function of (N, i, n_g) only, so building it needs no prefix sum over graphs.  31*N words are reserved.
// work items: one per (graph, group of DGD_ROWS rows); item ids [n0_g/64 + g, n0_{g+1}/64 + g + 1) are graph g's
// (>= ceil(n_g/64) of them; the surplus maps to -1), N/64 + B ids in total -- again no prefix sum.
#define DGD_MAXN 512
#define DGD_ROWS 64
#define DGD_CLASSES 5
static inline int dgd_num_items(int N, int B) { return N / DGD_ROWS + B; }
struct DgDense { const int* graph_ptr; const int* dmap; const unsigned* bits; int N, B, NW; };
#ifdef __HIPCC__
__host__ __device__ __forceinline__ int dgd_class(int ng) {     // smallest c with 32*2^c >= ng  (ng <= 512)
  const int k32 = (ng + 31) >> 5;
  return k32 <= 1 ? 0 : (k32 <= 2 ? 1 : (k32 <= 4 ? 2 : (k32 <= 8 ? 3 : 4)));
}
#endif

#ifdef __HIPCC__
// Kernel-A body, thread t of max(E, N+1, B+1): range / self-loop / strict (src,dst) order checks, colidx copies,
// rowptr by ROW-BOUNDARY detection, graph_ptr by binary search on the sorted batch vector.
__device__ __forceinline__ void dg_prep_fast_a_body(int t, const int64_t* __restrict__ ei, int E, int N,
                                                    const int64_t* __restrict__ batch, int B, int* __restrict__ rowptr,
                                                    int* __restrict__ colidx, int* __restrict__ rowptr_t,
                                                    int* __restrict__ colidx_t, int* __restrict__ graph_ptr,
                                                    unsigned int* __restrict__ err, unsigned int epoch,
                                                    unsigned int* __restrict__ bits = nullptr) {
  if (bits && t < N) {      // adjacency bitmap rows of node t, all five stride classes (phase B ORs the bits in)
#pragma unroll
    for (int c = 0; c < DGD_CLASSES; ++c) {
      unsigned int* row = bits + (size_t)N * ((1 << c) - 1) + (size_t)t * (1 << c);
#pragma unroll
      for (int w = 0; w < (1 << c); ++w) row[w] = 0u;
    }
  }
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
                                                    float* __restrict__ xs = nullptr, int F = 0,
                                                    const int64_t* __restrict__ batch = nullptr,
                                                    unsigned int* __restrict__ bits = nullptr,
                                                    int* __restrict__ dmap = nullptr) {
  if (bits) {
    // dense per-graph block structures (dg_dense.h): bit (j - n0_g) of row i <=> i and j adjacent or i == j; and the
    // work-item -> graph map.  Integer atomics only (order-independent result).
    if (t < N) {                                       // self loop bit
      const int g = (int)batch[t];
      if ((unsigned)g < (unsigned)B) {
        const int n0 = graph_ptr[g], ng = graph_ptr[g + 1] - n0, j = t - n0;
        if (j >= 0 && j < ng && ng <= DGD_MAXN) {
          const int c = dgd_class(ng);
          atomicOr(bits + (size_t)N * ((1 << c) - 1) + (size_t)t * (1 << c) + (j >> 5), 1u << (j & 31));
        }
      }
    }
    if (t < E) {
      const int64_t s = ei[t], d = ei[(int64_t)E + t];
      if ((uint64_t)s < (uint64_t)N && (uint64_t)d < (uint64_t)N) {
        const int g = (int)batch[s];
        if ((unsigned)g < (unsigned)B) {
          const int n0 = graph_ptr[g], ng = graph_ptr[g + 1] - n0, j = (int)d - n0;
          if (j < 0 || j >= ng) { err[1] = epoch; err[3] = ~epoch; }       // the edge leaves its graph
          else if (ng <= DGD_MAXN) {
            const int c = dgd_class(ng);
            atomicOr(bits + (size_t)N * ((1 << c) - 1) + (size_t)s * (1 << c) + (j >> 5), 1u << (j & 31));
          }
        }
      }
    }
    if (t < B) {                                       // work items [slot(t), slot(t+1)) belong to graph t
      const int n0 = graph_ptr[t], n1 = graph_ptr[t + 1], ng = n1 - n0;
      if (ng > DGD_MAXN) { err[1] = epoch; err[3] = ~epoch; }            // max_nodes promise (<= 512) broken
      const int s0 = n0 / DGD_ROWS + t, s1 = n1 / DGD_ROWS + t + 1, used = (ng + DGD_ROWS - 1) / DGD_ROWS;
      for (int w = s0; w < s1; ++w) dmap[w] = (w - s0 < used) ? t : -1;
    }
  }
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
