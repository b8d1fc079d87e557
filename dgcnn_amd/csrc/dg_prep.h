// dg_prep.h -- fast-path graph preparation bodies (see prep.hip for the algorithm notes), shared by the
// stand-alone kernels k_prep_fast_a / k_prep_fast_b and by the RIDER block ranges of k_readout_fwd /
// k_tail_bwd: those two launches run only B (= 50) workgroups, so ~80% of the chip idles for 10-15 us
// each; the pipelined training step uses that idle capacity to prepare the NEXT batch's graph structure
// (extra workgroups appended to the grid; same stream, so ordering is trivially safe).
#pragma once
#include "dg_common.h"

// batch assembly from a prepared dataset (dg_assemble.h: SURVEY N3) -- descriptor of one batch
struct DgAssemble {
  // the prepared dataset (device arrays)
  const int64_t* node_ptr;      // [G+1]
  const int32_t* ds_rowptr;     // [Ntot+1] CSR by target over the whole dataset
  const int32_t* ds_colidx;     // [Etot]   dataset-global node ids
  const float* ds_dinv;         // [Ntot]
  const float* ds_xs;           // [Ntot,F] dinv*x (F <= 32), else nullptr
  const float* ds_x;            // [Ntot,F]
  const uint32_t* ds_bits;      // class-strided by Ntot (31*Ntot words), or nullptr
  const int64_t* ds_y;          // [G]
  int64_t G, Ntot;
  // the batch: graph ids + exclusive prefix sums of their node / edge counts
  const int64_t* ids;           // [B]
  const int32_t* onode;         // [B+1]
  const int32_t* oedge;         // [B+1]
  int N, E, B, F;
  // outputs (batch workspace + the batch's own x / batch / y buffers)
  int32_t* rowptr; int32_t* colidx;        // nullptr: this batch's kernels read no CSR (bitmap forms only)
  float* dinv; float* xs; float* x; int64_t* batch; int64_t* y;
  int32_t* graph_ptr; int32_t* graph_eptr;
  uint32_t* bits;                          // nullptr: no bitmap form for this batch
  unsigned int* err; unsigned int epoch;
};

struct DgPrepRider {
  int mode;        // 0: graph preparation from the batch's int64 edge list (phases A and B below); 1: assembly from a prepared
                   // dataset (`as`; ONE phase, carried where phase A rides; nblk_b = 0)
  DgAssemble as;
  const int64_t* ei; const int64_t* batch;
  int E, N, B;
  int *rowptr, *colidx, *rowptr_t, *colidx_t, *graph_ptr, *graph_eptr;
  float* dinv;
  const float* x; float* xs; int F;      // aggregate-first conv1: xs[i] = dinv[i]*x[i] ([N,F]); x == nullptr: none
  unsigned int* err;
  unsigned int epoch;
  int nblk;        // rider workgroups appended to the host kernel's grid (0 = none): phase A
  int nblk_b;      // ... phase B (fewer when it has no per-edge work: 8 threads per node, not one per edge)
  unsigned int* bits; int* dmap;   // dense per-graph block structures (dg_dense.h); bits == nullptr: not built
  int edge_check;  // 1: the reverse-edge check stays the per-edge binary search of phase B although the bitmap is built
                   // (small batches that take only the chain forward from it: no third launch for the bitmap's symmetry check)
  int max_nodes;   // the host's per-graph node bound for this batch (0 = none given): phase B flags any graph above it, so that a
                   // hint that is too small can never make a size-class kernel skip a graph silently
  // BOTH phases in the launch that carries phase A (the one-launch training kernel, whose graph workgroups leave most CUs idle
  // at the reference's batch of 50): `fused_b` phase-B workgroups follow the `nblk` phase-A ones in the grid and wait on a
  // device counter until every phase-A workgroup has published its stores.  Workgroups are dispatched in index order, so a
  // waiting phase-B workgroup can never keep a phase-A workgroup from starting.  (As rider blocks of k_wgrad phase B was that
  // launch's longest chain: 7.0 us against 5.1 us without it.)
  unsigned int* sync_ctr;      // device counter (monotonic over the pipeline's life); nullptr: not available
  unsigned int* sync_host;     // host copy of the count of phase-A workgroups launched so far
  unsigned int sync_target;    // value the counter reaches when this launch's phase A is complete
  int fused_b;                 // phase-B workgroups in phase A's launch (0: phase B rides on a later launch / runs on its own)
};
static inline int dg_prep_fast_work(int E, int N, int B, bool dense = false) {      // threads of phase A / phase B
  int work = E > N + 1 ? E : N + 1;
  if (dense && 8LL * N < 0x7fffffffLL && 8 * N > work) work = 8 * N;                  // bitmap: 8 lanes per row
  return B + 1 > work ? B + 1 : work;
}

// phase B has per-edge work only when it checks the reverse edges itself (no bitmap, or `edge_check`); otherwise its threads
// beyond 8 per node would start and return -- 8.8 k empty 1024-thread blocks at 2048 COLLAB graphs, a third of the launch
static inline int dg_prep_fast_work_b(int E, int N, int B, bool dense, bool edge_check) {
  int work = N + 1;
  if (!dense || edge_check) work = E > work ? E : work;
  if (dense && 8LL * N < 0x7fffffffLL && 8 * N > work) work = 8 * N;
  return B + 1 > work ? B + 1 : work;
}

// ---- dense per-graph block structures (consumed by gcn_dense.hip) ----------------------------------------------
// adjacency bitmap: row i of graph g holds bit (j - n0_g) for every neighbour j and for j == i, in ceil(n_g/32) words;
// rows are stored with a power-of-two word stride S = 2^c >= ceil(n_g/32) (five classes, c = 0..4, i.e. graphs of up
// to 32/64/128/256/512 nodes); class c lives at word offset N*(2^c - 1), row i at i*2^c inside it -- every offset is
// a function of (N, i, n_g) only, so building it needs no prefix sum over graphs.  31*N words are reserved.
// work items: one per (graph, group of DGD_ROWS = 128 rows: 8 waves x one 16-row tile), packed in graph order: item table `dmap` =
//   [0 .. DGD_SPLITS]            split[k] = first item of share k: the items are cut into DGD_SPLITS contiguous shares of
//                                (nearly) equal COST (cost of an item = 3 * its pipeline stages + 1, i.e. ~ n_g), so that
//                                persistent workgroups that take equal numbers of shares finish together -- static,
//                                hence reproducible, and contiguous, hence L2-friendly
//   [dgd_sched0(N,B) + 2r ..]    graph SCHEDULE of the chain kernels (gcn_chain.hip): entry r = {first node, node count} of the
//                                graph of rank r when the graphs are ordered by 16-row tile count, descending, ties by graph
//                                index (a stable counting sort: deterministic).  Persistent workgroups deal themselves the
//                                entries in snake order (w, 2G-1-w, 2G+w, ...): largest first, equal sums, no atomics
//   [DGD_REC0 + 3w ..]           record of item w: {first node of its graph, node count, first row of the item}
// built by ONE workgroup of graph preparation's second phase with a block-wide prefix sum over the graphs
// (dg_prep_dense_plan); at most N/64 + B items.
#define DGD_MAXN 512
#ifndef DGD_ROWS
#define DGD_ROWS 128
#endif
#ifndef DGD_COST_STAGE
#define DGD_COST_STAGE 3       // cost of an item = DGD_COST_STAGE * stages + 1.  (6 / 12 / 32 measured: same kernel times and the
                               // same 5..10 stages per workgroup -- shares end at item boundaries, the item is the quantum)
#endif
#define DGD_CLASSES 5
#define DGD_SPLITS 3072
#define DGD_NBIG (DGD_SPLITS + 2)     // two words: number of graphs above 128 / above 256 nodes = first schedule entry of the rest
#define DGD_REC0 (DGD_SPLITS + 24)
static inline int dgd_num_items(int N, int B) { return N / DGD_ROWS + B; }       // upper bound
static inline int64_t dgd_sched0(int N, int B) { return (DGD_REC0 + 3 * (int64_t)(dgd_num_items(N, B) + 1) + 1) & ~1LL; }
static inline int64_t dgd_table_ints(int N, int B) { return dgd_sched0(N, B) + 2 * (int64_t)B + 2; }
struct DgDense { const int* graph_ptr; const int* dmap; const unsigned* bits; int N, B, NW; };
#include <type_traits>
#ifdef __HIPCC__
__host__ __device__ __forceinline__ int dgd_class(int ng) {     // smallest c with 32*2^c >= ng  (ng <= 512)
  const int k32 = (ng + 31) >> 5;
  return k32 <= 1 ? 0 : (k32 <= 2 ? 1 : (k32 <= 4 ? 2 : (k32 <= 8 ? 3 : 4)));
}
#endif

#ifdef __HIPCC__
// One workgroup of T threads (tid = its thread index): item records + equal-cost shares, see above.  Needs graph_ptr.
__device__ __forceinline__ void dg_prep_dense_plan(int tid, int T, int B, const int* __restrict__ graph_ptr,
                                                   int* __restrict__ dmap) {
  // (wave-level scans + one combine through LDS: three barriers per chunk of T graphs; the Hillis-Steele form over the
  //  whole workgroup took ~20 barriers per chunk and pass -- a quarter of phase B's time at 2048 graphs)
  __shared__ int wI[16], wC[16];
  __shared__ int carryI, carryC, totalC;
  const int lane = tid & 63, wave = tid >> 6, nw = (T + 63) >> 6;
  auto graph_items = [&](int g, int& n0, int& n, int& items, int& ic) {
    n0 = 0; n = 0;
    if (g < B) { n0 = graph_ptr[g]; n = graph_ptr[g + 1] - n0; if (n > DGD_MAXN) n = DGD_MAXN; if (n < 0) n = 0; }
    items = (n + DGD_ROWS - 1) / DGD_ROWS;
    ic = DGD_COST_STAGE * ((n + 63) / 64) + 1;      // cost of one item (pipeline stages of 64 k-rows + epilogue)
  };
  // ---- schedule of the chain kernels: stable counting sort of the graphs by tile count, descending --------------------
  {
    __shared__ int shist[33], sstart[33], scarry[33], swc[16][33];
    const int N = graph_ptr[B];
    int* sched = dmap + ((DGD_REC0 + 3 * (N / DGD_ROWS + B + 1) + 1) & ~1);
    auto bin_of = [&](int g, int& n0, int& n) {
      n0 = graph_ptr[g]; n = graph_ptr[g + 1] - n0;
      int t = (max(n, 0) + 15) >> 4;
      return 32 - min(t, 32);                 // bin 0: 32 tiles (or more: not admissible, flagged elsewhere) ... bin 32: empty graph
    };
    if (tid < 33) { shist[tid] = 0; scarry[tid] = 0; }
    __syncthreads();
    for (int g = tid; g < B; g += T) { int n0, n; atomicAdd(&shist[bin_of(g, n0, n)], 1); }     // (integer counts: order-free)
    __syncthreads();
    if (tid == 0) {
      int a = 0;
      for (int b = 0; b < 33; ++b) { sstart[b] = a; a += shist[b]; }
      dmap[DGD_NBIG] = sstart[24];            // bins 0..23 = 32..9 tiles = graphs above 128 nodes
      dmap[DGD_NBIG + 1] = sstart[16];        // bins 0..15 = 32..17 tiles = graphs above 256 nodes
    }
    __syncthreads();
    for (int base = 0; base < B; base += T) {
      const int g = base + tid;
      int n0 = 0, n = 0;
      const int bin = g < B ? bin_of(g, n0, n) : 33;
      // rank inside the wave among the lanes of the same bin, and the wave's count per bin: SIX ballots (one per bit of the bin
      // number; the lanes that agree with this lane on every bit are its bin's lanes) instead of one ballot per bin (33)
      unsigned long long same = ~0ull;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const bool bit = (bin >> k) & 1;
        const unsigned long long bk = __builtin_amdgcn_ballot_w64(bit);
        same &= bit ? bk : ~bk;
      }
      const unsigned long long below = same & ((1ull << lane) - 1ull);
      const int myrank = __builtin_popcountll(below);
      if (lane < 33) swc[wave][lane] = 0;                      // (the wave's own row: program order, no barrier)
      DG_LOCKSTEP();
      if (below == 0ull && bin < 33) swc[wave][bin] = __builtin_popcountll(same);
      __syncthreads();
      if (g < B) {
        int off = sstart[bin] + scarry[bin] + myrank;
        for (int w = 0; w < wave; ++w) off += swc[w][bin];
        sched[2 * off] = n0; sched[2 * off + 1] = n;
      }
      __syncthreads();
      if (tid < 33) { int a = 0; for (int w = 0; w < nw; ++w) a += swc[w][tid]; scarry[tid] += a; }
      __syncthreads();
    }
  }
  // pass 0: totals
  {
    int sI = 0, sC = 0;
    for (int g = tid; g < B; g += T) { int n0, n, items, ic; graph_items(g, n0, n, items, ic); sI += items; sC += items * ic; }
    for (int o = 32; o > 0; o >>= 1) { sI += __shfl_xor(sI, o); sC += __shfl_xor(sC, o); }
    if (lane == 0) { wI[wave] = sI; wC[wave] = sC; }
    __syncthreads();
    if (tid == 0) {
      int tI = 0, tC = 0;
      for (int w = 0; w < nw; ++w) { tI += wI[w]; tC += wC[w]; }
      totalC = tC; carryI = 0; carryC = 0;
      dmap[0] = 0; dmap[DGD_SPLITS + 1] = tI;
    }
    __syncthreads();
    if (totalC == 0)                                          // no work at all: every share is empty
      for (int k = tid; k <= DGD_SPLITS; k += T) dmap[k] = 0;
  }
  // pass 1: emit records and share boundaries, chunk by chunk
  const long long tot = totalC > 0 ? totalC : 1;
  for (int base = 0; base < B; base += T) {
    int n0, n, items, ic;
    graph_items(base + tid, n0, n, items, ic);
    int inI = items, inC = items * ic;                        // inclusive scans inside the wave
    for (int o = 1; o < 64; o <<= 1) {
      const int a = __shfl_up(inI, o), c = __shfl_up(inC, o);
      if (lane >= o) { inI += a; inC += c; }
    }
    if (lane == 63) { wI[wave] = inI; wC[wave] = inC; }
    __syncthreads();
    int preI = carryI, preC = carryC;
    for (int w = 0; w < wave; ++w) { preI += wI[w]; preC += wC[w]; }
    const int ioff = preI + inI - items, coff = preC + inC - items * ic;
    for (int r = 0; r < items; ++r) {
      const int w = ioff + r;
      int* rec = dmap + DGD_REC0 + 3 * w;
      rec[0] = n0; rec[1] = n; rec[2] = r * DGD_ROWS;
      const long long c0 = coff + (long long)r * ic, c1 = c0 + ic;
      int klo, khi;
      if (tot < (1ll << 20)) {       // (c * 3072 < 2^32; the usual case: 32-bit divisions -- the 64-bit ones are ~150 instructions each; same values)
        klo = (int)((unsigned)c0 * (unsigned)DGD_SPLITS / (unsigned)tot) + 1; khi = (int)((unsigned)c1 * (unsigned)DGD_SPLITS / (unsigned)tot);
      } else { klo = (int)(c0 * DGD_SPLITS / tot) + 1; khi = (int)(c1 * DGD_SPLITS / tot); }
      for (int k = klo; k <= khi && k <= DGD_SPLITS; ++k) dmap[k] = w + 1;
    }
    __syncthreads();                                          // everybody has read the carries and the wave totals
    if (tid == T - 1) { carryI = preI + inI; carryC = preC + inC; }
    __syncthreads();
  }
}

// Kernel-A body, thread t of max(E, N+1, B+1): range / self-loop / strict (src,dst) order checks, colidx copies,
// rowptr by ROW-BOUNDARY detection, graph_ptr by binary search on the sorted batch vector.
// COH (both phases inside ONE launch, dg_prep.h's fused form): what phase A hands to phase B goes through agent-coherent accesses
// (relaxed atomics: sc1 stores written through to memory, sc1 loads that take no stale line of another XCD's L2) instead of a
// release / acquire fence pair -- a fence writes back / invalidates the XCD's whole L2, and the graph workgroups of the same
// launch live out of that L2 (first version, one acquire per phase-B workgroup: the launch took 80-98 us instead of 36).
template <bool COH> __device__ __forceinline__ int dg_ldc(const int* p) {
  if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool COH> __device__ __forceinline__ void dg_stc(int* p, int v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH = false>
__device__ __forceinline__ void dg_prep_fast_a_body(int t, const int64_t* __restrict__ ei, int E, int N,
                                                    const int64_t* __restrict__ batch, int B, int* __restrict__ rowptr,
                                                    int* __restrict__ colidx, int* __restrict__ rowptr_t,
                                                    int* __restrict__ colidx_t, int* __restrict__ graph_ptr,
                                                    unsigned int* __restrict__ err, unsigned int epoch,
                                                    unsigned int* __restrict__ bits = nullptr) {
  (void)bits;
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
    // (an undirected edge list's CSR by source IS its CSR by target: the model-level callers pass null for the second copy
    //  and their backward reads the first -- 8 of phase A's 24 bytes per edge; dgcnn_graph_prep still fills both)
    dg_stc<COH>(colidx + t, dc);
    if (colidx_t) colidx_t[t] = dc;
    if (pc < sc || t == 0)
      for (int k = pc + 1; k <= sc; ++k) { dg_stc<COH>(rowptr + k, t); if (rowptr_t) rowptr_t[k] = t; }
    if (t == E - 1)
      for (int k = sc + 1; k <= N; ++k) { dg_stc<COH>(rowptr + k, E); if (rowptr_t) rowptr_t[k] = E; }
  }
  if (t <= B) {
    int lo = 0, hi = N;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (batch[mid] < (int64_t)t) lo = mid + 1; else hi = mid;
    }
    dg_stc<COH>(graph_ptr + t, lo);
  }
}
// what a phase-A rider thread does: the first phase of the next batch's preparation, or its whole assembly from a prepared dataset
__device__ __forceinline__ void dg_assemble_body(int t, const DgAssemble& A);
template <bool COH = false>
__device__ __forceinline__ void dg_rider_phase_a(int t, const DgPrepRider& rd) {
  if (rd.mode == 1) { dg_assemble_body(t, rd.as); return; }
  dg_prep_fast_a_body<COH>(t, rd.ei, rd.E, rd.N, rd.batch, rd.B, rd.rowptr, rd.colidx, rd.rowptr_t, rd.colidx_t, rd.graph_ptr, rd.err,
                      rd.epoch, rd.bits);
}
// Kernel-B body: dinv per node, graph_eptr, and (per edge (s,d)) the reverse edge (d,s) must be in row d --
// binary search inside that row only (<= log2(deg) steps).  Needs kernel A's outputs complete.
// THREADS: threads per block of the hosting launch (sizes the LDS row buffers: 8 bytes per thread -- 2 KB in the 256-thread
// launches; as a fixed 8 KB it kept the GCN backward chain kernels' workgroups, which need 2 x 70 KB of a CU's 160 KB, from being
// placed beside a few blocks of the preparation running on the pipeline's side stream)
template <int THREADS = 1024, bool EXTBUF = false, bool COH = false>      // EXTBUF: the 8-byte-per-thread LDS row buffer is handed in (rowbuf_ext)
__device__ __forceinline__ void dg_prep_fast_b_body(int t, const int64_t* __restrict__ ei, int E, int N, int B,
                                                    const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                    const int* __restrict__ graph_ptr, int* __restrict__ graph_eptr,
                                                    float* __restrict__ dinv, unsigned int* __restrict__ err,
                                                    unsigned int epoch, const float* __restrict__ x = nullptr,
                                                    float* __restrict__ xs = nullptr, int F = 0,
                                                    const int64_t* __restrict__ batch = nullptr,
                                                    unsigned int* __restrict__ bits = nullptr,
                                                    int* __restrict__ dmap = nullptr, bool edge_check = false,
                                                    int max_nodes = 0, unsigned int* rowbuf_ext = nullptr) {
  // neighbour id at position i of the CSR.  COH (both phases in one launch): NOT from phase A's int32 copy -- that would be an
  // agent-coherent load, a memory transaction per probe: with ~15 of them per edge the reverse-edge check moved ~150 MB through
  // the fabric beside the graph workgroups -- but from the batch's own sorted edge list, whose targets ARE the CSR's column ids
  // (a stable input: ordinary cached loads), clamped the way phase A clamps its copy.
  auto col_at = [&](int i) -> int {
    if (COH) { const int64_t dd = ei[(int64_t)E + i]; return (uint64_t)dd < (uint64_t)N ? (int)dd : 0; }
    return colidx[i];
  };

  if (bits) {
    // dense per-graph block structures (dg_dense.h): bit (j - n0_g) of row i <=> i and j adjacent or i == j.  EIGHT LANES
    // per row: lane l takes neighbours l, l + 8, ... of the row (int32 colidx copy of phase A; the 8 lanes read 8
    // consecutive ids), ALL of its loads in flight together (one round trip per 64 neighbours), ORs them into its own
    // copy of the row's words, and three xor-shuffles combine the 8 copies; lane k then stores word k -- no atomics, no
    // clearing pass, every word stored once, one load per edge.  (History at 2048 COLLAB-shaped graphs: an atomicOr per
    // edge 365 us; the first edge of every (row, word) group ORs its group 92 us; one thread walking the row 58 us;
    // 8 lanes each scanning the whole row for its own word 34 us.)
    const int row = t >> 3, l8 = t & 7;
    if (8LL * N < 0x7fffffffLL) {
      // (all 8 lanes of a row take the same branches: the shuffles below are executed by whole groups)
      // TWO dependent round trips per row, not four: the row's graph id and its CSR bounds are requested together (on a
      // clamped row, unconditionally), then the graph's bounds and the first 64 neighbours together -- the neighbour loads
      // need only the CSR bounds, their graph-relative ids are formed when both have landed.  (batch -> graph_ptr -> `ok` ->
      // rowptr -> colidx, each gated on the one before, made this a latency chain: 37 us at 2048 COLLAB graphs for 40 MB)
      const bool live = row < N;
      const int rowc = live ? row : 0;
      const int gload = N > 0 ? (int)batch[rowc] : -1;
      const int rs0 = N > 0 ? dg_ldc<COH>(rowptr + (rowc)) : 0, re0 = N > 0 ? dg_ldc<COH>(rowptr + (rowc + 1)) : 0;
      const int g = live ? gload : -1;
      const int gc = (unsigned)g < (unsigned)B ? g : 0;
      int n0 = B > 0 ? dg_ldc<COH>(graph_ptr + (gc)) : 0, ng = B > 0 ? dg_ldc<COH>(graph_ptr + (gc + 1)) - n0 : 0;
      if ((unsigned)g >= (unsigned)B) { n0 = 0; ng = 0; }
      const int sj = row - n0;
      const bool ok = live && (unsigned)g < (unsigned)B && sj >= 0 && sj < ng && ng <= DGD_MAXN;
      const int S = ok ? 1 << dgd_class(ng) : 0;
      const int rs = live ? rs0 : 0, re = live ? re0 : 0;      // (a row whose graph is not `ok` is walked too: it stores nothing, flags nothing)
      bool bad = false;
      auto build = [&](auto tag) {
        constexpr int W = decltype(tag)::value;           // words kept per lane (>= S)
        unsigned int acc[W];
#pragma unroll
        for (int k = 0; k < W; ++k) acc[k] = 0u;
        for (int e = rs + l8; e < re; e += 64) {
          int jj[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) jj[u] = col_at(min(e + 8 * u, re - 1)) - n0;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int j = jj[u];
            if (j < 0 || j >= ng) { bad = true; j = j < 0 ? 0 : ng - 1; }          // the edge leaves its graph
            const unsigned int bit = e + 8 * u < re ? 1u << (j & 31) : 0u;
            const int wi = j >> 5;
#pragma unroll
            for (int k = 0; k < W; ++k) acc[k] |= wi == k ? bit : 0u;
          }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
          acc[k] |= __shfl_xor(acc[k], 1); acc[k] |= __shfl_xor(acc[k], 2); acc[k] |= __shfl_xor(acc[k], 4);
        }
        if (ok) {
          unsigned int* rw = bits + (size_t)N * (S - 1) + (size_t)row * S;
#pragma unroll
          for (int k = 0; k < W; ++k)
            if (k < S && (k & 7) == l8) rw[k] = acc[k] | (k == (sj >> 5) ? 1u << (sj & 31) : 0u);
        }
      };
      // Rows of more than four words (graphs of 129..512 nodes -- where most of a COLLAB batch's edges are): the register
      // form costs 3 VALU operations per (neighbour, word) = 48 per neighbour.  Here the row's 16 words live in LDS (64 bytes
      // per group of 8 lanes); a lane's neighbours ascend (the row is sorted, it takes every 8th), so it collects the bits of
      // its current word in a register and ORs them into the LDS row (ds_or_b32) only when the word changes: ~8 operations
      // per neighbour.  The 8 lanes are in one wave and LDS operations of a wave execute in order: no barrier, only the
      // counter wait.  (2048 COLLAB graphs, phase B riding on k_tail_bwd: that launch 57 -> 51 us.)
      auto build_lds = [&]() {
        __shared__ unsigned int rowbuf[EXTBUF ? 1 : (THREADS / 8) * 16];
        unsigned int* rb = (EXTBUF ? rowbuf_ext : rowbuf) + (threadIdx.x >> 3) * 16;
        rb[l8] = 0u; rb[l8 + 8] = 0u;
        DG_WAIT_LGKM();
        unsigned int cur = 0u;
        int cw = 0;
        for (int e = rs + l8; e < re; e += 64) {
          int jj[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) jj[u] = col_at(min(e + 8 * u, re - 1)) - n0;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (e + 8 * u < re) {
              int j = jj[u];
              if (j < 0 || j >= ng) { bad = true; j = j < 0 ? 0 : ng - 1; }        // the edge leaves its graph
              const int wi = (j >> 5) & 15;
              if (wi != cw) {
                if (cur) __hip_atomic_fetch_or(&rb[cw], cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                cur = 0u; cw = wi;
              }
              cur |= 1u << (j & 31);
            }
          }
        }
        if (cur) __hip_atomic_fetch_or(&rb[cw], cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        DG_WAIT_LGKM();
        if (ok) {
          unsigned int* rw = bits + (size_t)N * (S - 1) + (size_t)row * S;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int k = l8 + 8 * h;
            if (k < S) rw[k] = rb[k] | (k == (sj >> 5) ? 1u << (sj & 31) : 0u);
          }
        }
      };
      // (rows of one wave may belong to graphs of different classes: every group of 8 aligned lanes runs the variant its
      //  own graph needs, and the shuffles stay inside the group)
      if (S <= 4) build(std::integral_constant<int, 4>{}); else build_lds();
      if (bad && ok) { err[1] = epoch; err[3] = ~epoch; }
    }
    // max_nodes promise broken: above the dense structures' bound (512), or above the bound the HOST gave -- the kernels behind
    // this preparation are chosen from that hint (size-class launches skipped, the 256-node chain backward taken), so a graph
    // above it would be left out of them without a trace
    // (max_nodes < 0: dataset-level preparation -- graphs above 512 nodes simply get no bitmap rows, dg_assemble.h)
    if (t < B && max_nodes >= 0 && dg_ldc<COH>(graph_ptr + (t + 1)) - dg_ldc<COH>(graph_ptr + (t)) > ((max_nodes > 0 && max_nodes < DGD_MAXN) ? max_nodes : DGD_MAXN)) { err[1] = epoch; err[3] = ~epoch; }
  }
  if (t < N) {
    const float di = 1.0f / sqrtf((float)(dg_ldc<COH>(rowptr + (t + 1)) - dg_ldc<COH>(rowptr + (t)) + 1));
    dinv[t] = di;
    if (x) {      // pre-scaled raw features for the aggregate-first conv1 gather (one row load per edge, no dinv[j] load)
      for (int f = 0; f < F; ++f) xs[(size_t)t * F + f] = di * x[(size_t)t * F + f];
    }
  }
  if (t <= B) graph_eptr[t] = dg_ldc<COH>(rowptr + (dg_ldc<COH>(graph_ptr + (t))));      // first edge position of each graph's rows
  if (t < E && (!bits || edge_check)) {      // (dense batches verify the reverse edges on the bitmap afterwards: dg_prep_sym_body)
    const int64_t s = ei[t], d = ei[(int64_t)E + t];
    if ((uint64_t)s < (uint64_t)N && (uint64_t)d < (uint64_t)N) {
      const int end = dg_ldc<COH>(rowptr + (d + 1));
      int a = dg_ldc<COH>(rowptr + (d));
      int b = end;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (col_at(mid) < (int)s) a = mid + 1; else b = mid;
      }
      if (!(a < end && col_at(a) == (int)s)) { err[1] = epoch; err[3] = ~epoch; }
    }
  }
}
// Phase C, dense batches only, after phase B: the reverse (d,s) of every edge (s,d) must exist, i.e. every graph's bit
// matrix must be SYMMETRIC.  Checked on the bitmap alone (2.5 MB at 2048 COLLAB-shaped graphs, L2-resident) -- thread
// (node i, word k of its row): for every set bit j, bit i of row j must be set -- instead of phase B's binary search
// through row d's neighbour list per edge (95 of phase B's 150 us), and without streaming the 16 B/edge int64 edge list
// again (a per-edge form of this check read 118 MB per launch: 27.6 us).
__device__ __forceinline__ void dg_prep_sym_body(int t, int N, int B, const int64_t* __restrict__ batch,
                                                 const int* __restrict__ graph_ptr, const unsigned int* __restrict__ bits,
                                                 unsigned int* __restrict__ err, unsigned int epoch) {
  const int i = t >> 2, kq = t & 3;
  if (i >= N) return;
  int g;
  if (batch) g = (int)batch[i];
  else {          // (a PREPARED batch's forward may come without the batch vector: the last graph whose first node is <= i)
    int lo = 0, hi = B;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (graph_ptr[mid] <= i) lo = mid; else hi = mid - 1; }
    g = lo;
  }
  if ((unsigned)g >= (unsigned)B) return;
  const int n0 = graph_ptr[g], ng = graph_ptr[g + 1] - n0, li = i - n0;
  if (ng > DGD_MAXN || li < 0 || li >= ng) return;                               // (flagged by phase B)
  const int S = 1 << dgd_class(ng), K32 = (ng + 31) >> 5;
  const unsigned int* base = bits + (size_t)N * (S - 1);
  const unsigned int* colw = base + (size_t)n0 * S + (li >> 5);                  // word (li >> 5) of row j: colw[j * S]
  const unsigned int ibit = 1u << (li & 31);
  unsigned int ok = ibit;
  for (int k = kq; k < K32; k += 4) {
    unsigned int w = base[(size_t)i * S + k];
    while (w) {                        // four reverse words in flight per round
      int j[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { j[u] = w ? 32 * k + __builtin_ctz(w) : li; w &= w - 1; }      // (exhausted: the diagonal, always set)
      unsigned int r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = colw[(size_t)j[u] * S];
      ok &= r[0] & r[1] & r[2] & r[3];
    }
  }
  if (!(ok & ibit)) { err[1] = epoch; err[3] = ~epoch; }
}
#endif
#include "dg_assemble.h"     // dg_assemble_body (dg_rider_phase_a above calls it)
