// prep.hip -- per-batch graph preparation kernels (gfx950).
//
// Replaces, once per batch, what the reference does inside every forward:
//   remove_self_loops                      /root/reference/model.py:28
//   PyG gcn_norm (degree, deg^-1/2)        inside each GCNConv call, model.py:30-33 (4x per forward)
//   per-graph node ranges                  PyG to_dense_batch inside SortAggregation, model.py:35
// Output: CSR by target (forward gather) + CSR by source (backward gather), neighbour lists
// sorted ascending so every floating-point sum downstream has a fixed order (bit-reproducible),
// dinv[i] = (in-degree(i)+1)^-1/2, graph_ptr[B+1].
//
// All integer work: HBM/latency-bound, int32 atomics only (order-independent results because
// every row is sorted afterwards).
#include "dg_common.h"
#include "dg_prep.h"

// ---- 1. count degrees (skipping self loops) + graph_ptr by binary search on sorted batch ----
__global__ void __launch_bounds__(256)
k_prep_count(const int64_t* __restrict__ ei, int E, int N, const int64_t* __restrict__ batch, int B,
             int* __restrict__ cnt_in, int* __restrict__ cnt_out, int* __restrict__ graph_ptr,
             unsigned int* __restrict__ err, unsigned int epoch) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < E) {
    const int64_t s = ei[t], d = ei[(int64_t)E + t];
    if ((uint64_t)s >= (uint64_t)N || (uint64_t)d >= (uint64_t)N) {
      err[0] = epoch; err[2] = ~epoch;      // epoch-tagged: the words never need clearing
    } else if (s != d) {
      atomicAdd(&cnt_in[(int)d], 1);
      atomicAdd(&cnt_out[(int)s], 1);
    }
  }
  if (t <= B) {  // graph_ptr[t] = first node index with batch >= t
    int lo = 0, hi = N;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (batch[mid] < (int64_t)t) lo = mid + 1; else hi = mid;
    }
    graph_ptr[t] = lo;
  }
}

// ---- 2. exclusive scan of both degree arrays (single workgroup, 1024 threads), dinv ----
// In place: cnt_* become the fill cursors (= rowptr values); rowptr* get the same values.
__global__ void __launch_bounds__(1024)
k_prep_scan(int N, int* __restrict__ cnt_in, int* __restrict__ cnt_out, int* __restrict__ rowptr,
            int* __restrict__ rowptr_t, float* __restrict__ dinv) {
  __shared__ int wsum[2][16];
  __shared__ int carry[2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) { carry[0] = 0; carry[1] = 0; }
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + threadIdx.x;
    const int a = i < N ? cnt_in[i] : 0;
    const int b = i < N ? cnt_out[i] : 0;
    int sa = a, sb = b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int ta = __shfl_up(sa, o), tb = __shfl_up(sb, o);
      if (lane >= o) { sa += ta; sb += tb; }
    }
    if (lane == 63) { wsum[0][w] = sa; wsum[1][w] = sb; }
    __syncthreads();
    if (w == 0) {
      int va = lane < 16 ? wsum[0][lane] : 0, vb = lane < 16 ? wsum[1][lane] : 0;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const int ta = __shfl_up(va, o), tb = __shfl_up(vb, o);
        if (lane >= o) { va += ta; vb += tb; }
      }
      if (lane < 16) { wsum[0][lane] = va; wsum[1][lane] = vb; }
    }
    __syncthreads();
    const int offa = carry[0] + (w ? wsum[0][w - 1] : 0);
    const int offb = carry[1] + (w ? wsum[1][w - 1] : 0);
    if (i < N) {
      const int ea = offa + sa - a, eb = offb + sb - b;
      rowptr[i] = ea; cnt_in[i] = ea;
      rowptr_t[i] = eb; cnt_out[i] = eb;
      dinv[i] = 1.0f / sqrtf((float)(a + 1));   // deg^-1/2 with deg = in-degree + self loop
    }
    __syncthreads();
    if (threadIdx.x == 1023) { carry[0] = offa + sa; carry[1] = offb + sb; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { rowptr[N] = carry[0]; rowptr_t[N] = carry[1]; }
}

// ---- 3. fill both adjacency arrays through atomic cursors (order fixed up by step 4) ----
__global__ void __launch_bounds__(256)
k_prep_fill(const int64_t* __restrict__ ei, int E, int N, int B, int* __restrict__ cur_in,
            int* __restrict__ cur_out, int* __restrict__ colidx, int* __restrict__ colidx_t,
            const int* __restrict__ rowptr, const int* __restrict__ graph_ptr, int* __restrict__ graph_eptr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= B) graph_eptr[t] = rowptr[graph_ptr[t]];      // first edge position of each graph's rows
  if (t >= E) return;
  const int64_t s = ei[t], d = ei[(int64_t)E + t];
  if ((uint64_t)s >= (uint64_t)N || (uint64_t)d >= (uint64_t)N || s == d) return;
  const int p = atomicAdd(&cur_in[(int)d], 1);
  colidx[p] = (int)s;
  const int q = atomicAdd(&cur_out[(int)s], 1);
  colidx_t[q] = (int)d;
}

// ---- 4. sort every row ascending: wave per short row (<= 64), workgroup per long row ----
#define DG_SORT_LDS 8192   // ints of LDS for long rows (32 KiB)

__global__ void __launch_bounds__(256)
k_prep_sort_rows(int N, const int* __restrict__ rowptr, int* __restrict__ colidx,
                 const int* __restrict__ rowptr_t, int* __restrict__ colidx_t) {
  __shared__ int buf[DG_SORT_LDS];
  __shared__ int long_row[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + w;   // rows 0..N-1 -> CSR by target, N..2N-1 -> CSR by source
  long_row[w] = -1;   // each wave owns its slot; published by the barrier below
  DG_LOCKSTEP();
  if (r < 2 * N) {
    const int* rp = r < N ? rowptr : rowptr_t;
    int* col = r < N ? colidx : colidx_t;
    const int i = r < N ? r : r - N;
    const int start = rp[i], d = rp[i + 1] - start;
    if (d > 64) {
      if (lane == 0) long_row[w] = r;
    } else if (d > 1) {
      const int v = lane < d ? col[start + lane] : 0x7fffffff;
      int rank = 0;
      for (int m = 0; m < d; ++m) {
        const int u = __shfl(v, m);
        rank += (u < v || (u == v && m < lane)) ? 1 : 0;
      }
      if (lane < d) col[start + rank] = v;
    }
  }
  __syncthreads();
  for (int q = 0; q < 4; ++q) {
    const int rr = long_row[q];          // workgroup-uniform
    if (rr < 0) continue;
    const int* rp = rr < N ? rowptr : rowptr_t;
    int* col = rr < N ? colidx : colidx_t;
    const int i = rr < N ? rr : rr - N;
    const int start = rp[i], d = rp[i + 1] - start;
    if (d <= DG_SORT_LDS) {
      for (int t = threadIdx.x; t < d; t += blockDim.x) buf[t] = col[start + t];
      __syncthreads();
      dg_block_bitonic<int>(buf, d);
      for (int t = threadIdx.x; t < d; t += blockDim.x) col[start + t] = buf[t];
      __syncthreads();
    } else {
      dg_block_bitonic<int>(col + start, d);   // rare: in place in global memory (same workgroup only)
    }
  }
}

// ---- 0. clear the degree counters (one launch instead of memsets).  Errors are reported through
// EPOCH-TAGGED words err[4]: {range error tag, layout error tag, ~range tag, ~layout tag}; a forward
// call is in error iff err[k] == epoch && err[k+2] == ~epoch, so the words never need clearing. ----
__global__ void __launch_bounds__(256)
k_prep_zero(int N, int* __restrict__ cnt_in, int* __restrict__ cnt_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= N) { cnt_in[t] = 0; cnt_out[t] = 0; }
}

// ---- fast path: the caller promises a COALESCED UNDIRECTED edge list ---------------------------
// (sorted by (src,dst), no duplicates, no self loops, both directions present -- what a TU dataset
// file / PyG `coalesce` + `to_undirected` yields, and what the reference's datasets contain).
// Then CSR-by-source is the edge list itself and, by symmetry, CSR-by-target equals it:
//   rowptr[i] = lower_bound(src, i),  colidx[e] = dst[e],  indeg = outdeg = rowptr[i+1]-rowptr[i].
// No atomics, no sort, ONE launch.  The promise is VERIFIED on the device (sortedness, range, no self
// loop, and the reverse edge found by binary search); a violation writes the epoch tag to err[1]/err[3].
// Kernel A (per edge, O(1) each): range / self-loop / strict (src,dst) order checks, colidx copies,
// and rowptr by ROW-BOUNDARY detection: the thread at the first edge of source s writes rowptr[k] = t
// for every k in (previous source, s] (nodes without edges get empty rows); the last edge's thread
// closes rowptr[(src_last, N]] = E.  graph_ptr by binary search on the sorted batch vector.
__global__ void __launch_bounds__(256)
k_prep_fast_a(const int64_t* __restrict__ ei, int E, int N, const int64_t* __restrict__ batch, int B,
              int* __restrict__ rowptr, int* __restrict__ colidx, int* __restrict__ rowptr_t,
              int* __restrict__ colidx_t, int* __restrict__ graph_ptr, unsigned int* __restrict__ err,
              unsigned int epoch, unsigned int* __restrict__ bits) {
  dg_prep_fast_a_body(blockIdx.x * blockDim.x + threadIdx.x, ei, E, N, batch, B, rowptr, colidx, rowptr_t, colidx_t,
                      graph_ptr, err, epoch, bits);
}

// Kernel B: dinv per node, and (per edge (s,d)) the reverse edge (d,s) must be in row d -- binary
// search inside that row only (<= log2(deg) steps).  Pure verification + dinv; no atomics.  With x given it also
// writes the pre-scaled raw features xs = dinv*x [N,F] for the aggregate-first conv1 (F <= DG_AF_MAX_F).
// 1024 threads per block: block 0 also plans a dense batch (dg_prep_dense_plan, ONE workgroup -- 8 us with 1024 threads at 2048
// graphs, ~30 us with 256, which made this launch 38 us long for 10 us of parallel work)
__global__ void __launch_bounds__(1024)
k_prep_fast_b(const int64_t* __restrict__ ei, int E, int N, int B, const int* __restrict__ rowptr,
              const int* __restrict__ colidx, const int* __restrict__ graph_ptr, int* __restrict__ graph_eptr,
              float* __restrict__ dinv, unsigned int* __restrict__ err, unsigned int epoch, int F,
              const float* __restrict__ x, float* __restrict__ xs, const int64_t* __restrict__ batch,
              unsigned int* __restrict__ bits, int* __restrict__ dmap, int edge_check, int max_nodes) {
  dg_prep_fast_b_body<1024>(blockIdx.x * blockDim.x + threadIdx.x, ei, E, N, B, rowptr, colidx, graph_ptr, graph_eptr, dinv,
                      err, epoch, x, xs, F, batch, bits, dmap, edge_check == 1, max_nodes);
  if (dmap && blockIdx.x == 0) dg_prep_dense_plan(threadIdx.x, 1024, B, graph_ptr, dmap);
}

__global__ void __launch_bounds__(256)
k_prep_sym(int N, int B, const int64_t* __restrict__ batch, const int* __restrict__ graph_ptr,
           const unsigned int* __restrict__ bits, unsigned int* __restrict__ err, unsigned int epoch) {
  dg_prep_sym_body(blockIdx.x * blockDim.x + threadIdx.x, N, B, batch, graph_ptr, bits, err, epoch);
}
int dg_launch_prep_sym(const int64_t* edge_index, int E, int N, int B, const int64_t* batch, const int32_t* graph_ptr,
                       const uint32_t* bits, int32_t* err, uint32_t epoch, hipStream_t s) {
  (void)edge_index;
  if (E <= 0 || !bits) return DGCNN_OK;
  hipLaunchKernelGGL(k_prep_sym, dim3(dg_cdiv(4 * N, 256)), dim3(256), 0, s, N, B, batch, graph_ptr, bits,
                     reinterpret_cast<unsigned int*>(err), epoch);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

// xs[i][f] = dinv[i] * x[i][f]  (general prep path; the fast path does it inside k_prep_fast_b)
__global__ void __launch_bounds__(256)
k_scale_x(int N, int F, const float* __restrict__ x, const float* __restrict__ dinv, float* __restrict__ xs) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < (int64_t)N * F) xs[t] = dinv[t / F] * x[t];
}

// phase B of a rider's preparation as a launch of its own: for hosts whose remaining launches carry no rider range (the
// pipelined EVALUATION step behind a launch that carried phase A only)
int dg_launch_prep_phase_b(const DgPrepRider* rd, hipStream_t s) {
  if (!rd || rd->mode != 0 || rd->E <= 0 || rd->N <= 0 || rd->B <= 0) return DGCNN_EINVAL;
  const int work_b = dg_prep_fast_work_b(rd->E, rd->N, rd->B, rd->bits != nullptr, rd->edge_check == 1);
  hipLaunchKernelGGL(k_prep_fast_b, dim3(dg_cdiv(work_b, 1024)), dim3(1024), 0, s, rd->ei, rd->E, rd->N, rd->B, rd->rowptr, rd->colidx,
                     rd->graph_ptr, rd->graph_eptr, rd->dinv, rd->err, rd->epoch, rd->x ? rd->F : 0, rd->x, rd->xs, rd->batch, rd->bits,
                     rd->bits ? rd->dmap : nullptr, rd->edge_check, rd->max_nodes);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

int dg_launch_prep(const int64_t* edge_index, int E, const int64_t* batch, int N, int B,
                   int32_t* rowptr, int32_t* colidx, int32_t* rowptr_t, int32_t* colidx_t,
                   float* dinv, int32_t* graph_ptr, int32_t* graph_eptr, int32_t* cnt_in, int32_t* cnt_out,
                   int32_t* err, int flags, uint32_t epoch, hipStream_t s, const DgLinFirst* lf, int* lin_done,
                   uint32_t* bits, int32_t* dmap, int edge_check, int max_nodes) {
  if (N <= 0 || E < 0 || B <= 0) return DGCNN_EINVAL;
  if (lin_done) *lin_done = 0;
  if (!bits) dmap = nullptr;                 // (the bitmap alone is a valid request: chain forward of a small batch)
  unsigned int* uerr = reinterpret_cast<unsigned int*>(err);
  if ((flags & DGCNN_FLAG_COALESCED_UNDIRECTED) && E > 0) {
    const int work = dg_prep_fast_work(E, N, B, bits != nullptr);
    hipLaunchKernelGGL(k_prep_fast_a, dim3(dg_cdiv(work, 256)), dim3(256), 0, s, edge_index, E, N, batch, B, rowptr,
                       colidx, rowptr_t, colidx_t, graph_ptr, uerr, epoch, bits);
    DG_CHECK_LAUNCH();
    const bool scale = lf && lf->x && !lf->W && lf->F >= 1 && lf->F <= DGCNN_MAX_F;
    const int work_b = dg_prep_fast_work_b(E, N, B, bits != nullptr, edge_check == 1);
    hipLaunchKernelGGL(k_prep_fast_b, dim3(dg_cdiv(work_b, 1024)), dim3(1024), 0, s, edge_index, E, N, B, rowptr, colidx,
                       graph_ptr, graph_eptr, dinv, uerr, epoch, scale ? lf->F : 0, scale ? lf->x : nullptr,
                       scale ? lf->hs : nullptr, batch, bits, dmap, edge_check, max_nodes);
    if (scale && lin_done) *lin_done = 1;
    DG_CHECK_LAUNCH();
    if (edge_check) return DGCNN_OK;             // (the reverse edges were checked per edge in phase B)
    return dg_launch_prep_sym(edge_index, E, N, B, batch, graph_ptr, bits, err, epoch, s);
  }
  hipLaunchKernelGGL(k_prep_zero, dim3(dg_cdiv(N + 1, 256)), dim3(256), 0, s, N, cnt_in, cnt_out);
  DG_CHECK_LAUNCH();
  const int work = E > B + 1 ? E : B + 1;
  hipLaunchKernelGGL(k_prep_count, dim3(dg_cdiv(work, 256)), dim3(256), 0, s, edge_index, E, N, batch, B,
                     cnt_in, cnt_out, graph_ptr, uerr, epoch);
  DG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_prep_scan, dim3(1), dim3(1024), 0, s, N, cnt_in, cnt_out, rowptr, rowptr_t, dinv);
  DG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_prep_fill, dim3(dg_cdiv(work, 256)), dim3(256), 0, s, edge_index, E, N, B, cnt_in, cnt_out,
                     colidx, colidx_t, rowptr, graph_ptr, graph_eptr);
  DG_CHECK_LAUNCH();
  if (E > 0) {
    hipLaunchKernelGGL(k_prep_sort_rows, dim3(dg_cdiv(2 * N, 4)), dim3(256), 0, s, N, rowptr, colidx, rowptr_t,
                       colidx_t);
    DG_CHECK_LAUNCH();
  }
  if (lf && lf->x && !lf->W && lf->F >= 1 && lf->F <= DGCNN_MAX_F) {     // scale mode on the general path
    const int64_t tot = (int64_t)N * lf->F;
    hipLaunchKernelGGL(k_scale_x, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, N, lf->F, lf->x, dinv, lf->hs);
    DG_CHECK_LAUNCH();
    if (lin_done) *lin_done = 1;
  }
  return DGCNN_OK;
}
