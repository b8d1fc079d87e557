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
