// collate.hip -- mini-batch assembly on the device (gfx950): the disjoint union PyG's DataLoader collate builds on
// the host for every batch (/root/reference/train.py:108-109: x rows concatenated, edge_index shifted by the node
// offset of its graph, the `batch` vector, the labels), here from a dataset that lives in HBM (a whole TU dataset is
// well under 1 GB) -- no per-batch host loop over graphs, no PCIe copy of the batch.
//
// Dataset layout (built once, dgcnn_amd/device_data.py): x_all [Ntot,F] f32, ei_all [2,Etot] i64 with GRAPH-LOCAL node
// ids, node_ptr [G+1] / edge_ptr [G+1] i64, y_all [G] i64.  Per batch the host supplies the graph ids [B] (i64) and the two
// exclusive prefix sums of their node / edge counts (B+1 values each: a few hundred bytes, computed with numpy); ONE
// launch writes x [N,F], edge_index [2,E], batch [N], y [B].  Pure data movement: bit-exact by construction.
#include "dg_common.h"

__device__ __forceinline__ int dg_upper_seg(const int64_t* __restrict__ ptr, int B, int64_t t) {
  // largest g in [0,B) with ptr[g] <= t   (ptr ascending, ptr[0] = 0, t < ptr[B])
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256)
k_collate(int B, int F, int64_t N, int64_t E, int64_t Etot, const int64_t* __restrict__ ids,
          const int64_t* __restrict__ onode, const int64_t* __restrict__ oedge, const float* __restrict__ x_all,
          const int64_t* __restrict__ ei_all, const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr,
          const int64_t* __restrict__ y_all, float* __restrict__ x, int64_t* __restrict__ ei, int64_t* __restrict__ batch,
          int64_t* __restrict__ y, int64_t nblk_nodes) {
  if ((int64_t)blockIdx.x < nblk_nodes) {
    // node range: one thread per (node, feature chunk of 4) keeps wide rows coalesced; F is small (<= a few dozen)
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < N) {
      const int g = dg_upper_seg(onode, B, t);
      const int64_t src = node_ptr[ids[g]] + (t - onode[g]);
      batch[t] = g;
      const float* xs = x_all + src * F;
      float* xd = x + t * F;
      for (int f = 0; f < F; ++f) xd[f] = xs[f];
    }
    if (t < B) y[t] = y_all[ids[t]];
  } else {
    const int64_t t = ((int64_t)blockIdx.x - nblk_nodes) * 256 + threadIdx.x;
    if (t < E) {
      const int g = dg_upper_seg(oedge, B, t);
      const int64_t src = edge_ptr[ids[g]] + (t - oedge[g]);
      const int64_t off = onode[g];
      ei[t] = ei_all[src] + off;
      ei[E + t] = ei_all[Etot + src] + off;
    }
  }
}

// Small batches (B <= DG_COLLATE_SCAN_MAX_B): no per-batch upload at all.  The graph ids of the WHOLE epoch sit in device
// memory (one upload per epoch); every workgroup rebuilds the two prefix sums of this batch's B graphs in LDS from
// node_ptr / edge_ptr (a block-level scan, a few hundred loads) and then does the same copy work as k_collate.
#define DG_COLLATE_SCAN_MAX_B 256
__global__ void __launch_bounds__(256)
k_collate_scan(int B, int F, int64_t N, int64_t E, int64_t Etot, const int64_t* __restrict__ ids,
               const float* __restrict__ x_all, const int64_t* __restrict__ ei_all, const int64_t* __restrict__ node_ptr,
               const int64_t* __restrict__ edge_ptr, const int64_t* __restrict__ y_all, float* __restrict__ x,
               int64_t* __restrict__ ei, int64_t* __restrict__ batch, int64_t* __restrict__ y, int64_t nblk_nodes) {
  __shared__ int64_t onode[DG_COLLATE_SCAN_MAX_B + 1], oedge[DG_COLLATE_SCAN_MAX_B + 1];
  __shared__ int64_t gn0[DG_COLLATE_SCAN_MAX_B], ge0[DG_COLLATE_SCAN_MAX_B];
  const int tid = threadIdx.x;
  int64_t cn = 0, ce = 0;
  if (tid < B) {
    const int64_t g = ids[tid];
    const int64_t a = node_ptr[g], b = edge_ptr[g];
    gn0[tid] = a; ge0[tid] = b;
    cn = node_ptr[g + 1] - a; ce = edge_ptr[g + 1] - b;
  }
  onode[tid + 1 <= DG_COLLATE_SCAN_MAX_B ? tid + 1 : 0] = cn;      // inclusive scan input at [1..B]
  oedge[tid + 1 <= DG_COLLATE_SCAN_MAX_B ? tid + 1 : 0] = ce;
  if (tid == 0) { onode[0] = 0; oedge[0] = 0; }
  __syncthreads();
  for (int off = 1; off < DG_COLLATE_SCAN_MAX_B; off <<= 1) {       // Hillis-Steele over the 256 slots [1..256]
    int64_t an = 0, ae = 0;
    const int k = tid + 1;
    if (k - off >= 1) { an = onode[k - off]; ae = oedge[k - off]; }
    __syncthreads();
    if (k - off >= 1) { onode[k] += an; oedge[k] += ae; }
    __syncthreads();
  }
  if ((int64_t)blockIdx.x < nblk_nodes) {
    const int64_t t = (int64_t)blockIdx.x * 256 + tid;
    if (t < N) {
      const int g = dg_upper_seg(onode, B, t);
      const int64_t src = gn0[g] + (t - onode[g]);
      batch[t] = g;
      const float* xs = x_all + src * F;
      float* xd = x + t * F;
      for (int f = 0; f < F; ++f) xd[f] = xs[f];
    }
    if (t < B) y[t] = y_all[ids[t]];
  } else {
    const int64_t t = ((int64_t)blockIdx.x - nblk_nodes) * 256 + tid;
    if (t < E) {
      const int g = dg_upper_seg(oedge, B, t);
      const int64_t src = ge0[g] + (t - oedge[g]);
      const int64_t off = onode[g];
      ei[t] = ei_all[src] + off;
      ei[E + t] = ei_all[Etot + src] + off;
    }
  }
}

int dg_launch_collate_scan(int B, int F, int64_t N, int64_t E, int64_t Etot, const int64_t* ids_dev, const float* x_all,
                           const int64_t* ei_all, const int64_t* node_ptr, const int64_t* edge_ptr, const int64_t* y_all,
                           float* x, int64_t* ei, int64_t* batch, int64_t* y, hipStream_t s) {
  if (B <= 0 || B > DG_COLLATE_SCAN_MAX_B || F < 1 || N <= 0 || E < 0) return DGCNN_EINVAL;
  const int64_t nb_nodes = (((N > B ? N : (int64_t)B)) + 255) / 256, nb_edges = (E + 255) / 256;
  hipLaunchKernelGGL(k_collate_scan, dim3((unsigned)(nb_nodes + nb_edges)), dim3(256), 0, s, B, F, N, E, Etot, ids_dev, x_all,
                     ei_all, node_ptr, edge_ptr, y_all, x, ei, batch, y, nb_nodes);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

int dg_launch_collate(int B, int F, int64_t N, int64_t E, int64_t Etot, const int64_t* ids, const int64_t* onode,
                      const int64_t* oedge, const float* x_all, const int64_t* ei_all, const int64_t* node_ptr,
                      const int64_t* edge_ptr, const int64_t* y_all, float* x, int64_t* ei, int64_t* batch, int64_t* y,
                      hipStream_t s) {
  if (B <= 0 || F < 1 || N <= 0 || E < 0) return DGCNN_EINVAL;
  const int64_t nb_nodes = (((N > B ? N : (int64_t)B)) + 255) / 256, nb_edges = (E + 255) / 256;
  hipLaunchKernelGGL(k_collate, dim3((unsigned)(nb_nodes + nb_edges)), dim3(256), 0, s, B, F, N, E, Etot, ids, onode, oedge,
                     x_all, ei_all, node_ptr, edge_ptr, y_all, x, ei, batch, y, nb_nodes);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}
