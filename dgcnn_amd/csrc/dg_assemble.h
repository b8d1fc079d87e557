// dg_assemble.h -- batch assembly from a PREPARED dataset (as written: "pre-built per-graph CSR + dinv
// stored once per dataset, batch assembly = offset add + concat kernel").
//
// The reference collates every batch of every epoch on the host (PyG DataLoader, /root/reference/train.py:108-109) and then
// re-derives the normalisation of its block-diagonal adjacency inside each of the four GCNConv calls
// (/root/reference/model.py:27-33).  All of it is a function of the GRAPHS, not of the batch they travel in: a graph's CSR rows,
// its dinv = (indeg+1)^-1/2, its pre-scaled features dinv*x and its bit-packed adjacency rows are the same in every batch.
// dgcnn_dataset_prepare builds them ONCE for the whole dataset (one block-diagonal "batch" of all G graphs through the SAME
// graph-preparation kernels a batch goes through: the numbers are the per-batch ones bit for bit, and the edge lists are
// verified once); a training batch is then this file's copy kernel:
//   per node  : dinv, xs row, x row, bitmap row (re-based between the two class-strided layouts), rowptr (offset add)
//   per edge  : colidx (offset add: dataset-global node id -> batch node id), only where the batch's kernels read a CSR
//   per graph : graph_ptr, graph_eptr, label
// No int64 edge list is assembled or read, nothing is checked again (graph ids are range-checked), 8 B instead of 52 B per edge.
#pragma once
#ifndef DG_ASSEMBLE_H
#define DG_ASSEMBLE_H
#include "dg_common.h"
#include "dg_prep.h"      // dgd_class, DGD_MAXN: the bitmap's class-strided layout

// (struct DgAssemble: dg_prep.h, beside the rider descriptor that embeds it)
// launch of its own (dataset.hip); dmap != null: its first workgroup is the planning workgroup (item table + graph schedule --
// a function of the batch's node prefix sums alone, which the host hands in: it needs nothing the copy produces)
int dg_launch_assemble(const DgAssemble* A, int32_t* dmap, hipStream_t s);
#define DG_ASM_TPG 256                      // virtual threads per graph
static inline int dg_assemble_work(int N, int E, int B, bool csr) { (void)N; (void)E; (void)csr; return DG_ASM_TPG * B; }

#ifdef __HIPCC__
// virtual thread t of dg_assemble_work(...): GRAPH-CENTRIC -- 256 threads per graph read the graph's record once (same-address
// loads) and copy its contiguous runs: dinv / rowptr per node, the feature and pre-scaled feature rows as ONE run of n*F
// floats, the bitmap rows as ONE run of n*S words (a graph's rows are contiguous in both class-strided layouts), the column
// indices as one run of the graph's edges.  (One thread per node / per edge, each finding its graph by a binary search over
// the prefix sums -- 11 dependent loads in front of 4 more -- took 34.6 us for 2048 graphs on the side stream and stretched the
// GCN backward it runs beside from 31 to 51 us.)
__device__ __forceinline__ void dg_assemble_body(int t, const DgAssemble& A) {
  const int g = t / DG_ASM_TPG, l = t - g * DG_ASM_TPG;
  const int N = A.N, E = A.E, B = A.B, F = A.F;
  if (g >= B) return;
  int64_t gid = A.ids[g];
  bool bad = false;
  if ((uint64_t)gid >= (uint64_t)A.G) { bad = true; gid = 0; if (l == 0) { A.err[0] = A.epoch; A.err[2] = ~A.epoch; } }
  const int64_t dn0 = A.node_ptr[gid];
  const int ng_ds = (int)(A.node_ptr[gid + 1] - dn0);
  const int ob = A.onode[g], oe = A.oedge[g];
  int ng = A.onode[g + 1] - ob;
  if (ng != ng_ds && l == 0) { A.err[1] = A.epoch; A.err[3] = ~A.epoch; }      // the prefix sums are not this graph list's
  ng = min(min(ng, ng_ds), N - ob);          // (inconsistent prefix sums: flagged; stay in bounds)
  if (ng < 0) ng = 0;
  if (l == 0) {
    A.graph_ptr[g] = ob; A.graph_eptr[g] = oe;
    if (A.y) A.y[g] = bad ? 0 : A.ds_y[gid];
    if (g == B - 1) { A.graph_ptr[B] = A.onode[B]; A.graph_eptr[B] = A.oedge[B]; if (A.rowptr) A.rowptr[N] = E; }
  }
  const bool want_rows = A.rowptr || A.colidx;
  const int e_lo = want_rows ? A.ds_rowptr[dn0] : 0;
  for (int li = l; li < ng; li += DG_ASM_TPG) {
    A.dinv[ob + li] = A.ds_dinv[dn0 + li];
    if (A.batch) A.batch[ob + li] = g;
    if (A.rowptr) A.rowptr[ob + li] = A.ds_rowptr[dn0 + li] - e_lo + oe;
  }
  const int nf = ng * F;
  if (A.ds_xs) { const float* sp = A.ds_xs + dn0 * F; float* dp = A.xs + (size_t)ob * F; for (int q = l; q < nf; q += DG_ASM_TPG) dp[q] = sp[q]; }
  if (A.x) { const float* sp = A.ds_x + dn0 * F; float* dp = A.x + (size_t)ob * F; for (int q = l; q < nf; q += DG_ASM_TPG) dp[q] = sp[q]; }
  if (A.bits && A.ds_bits && ng_ds <= DGD_MAXN && ng == ng_ds && ng > 0) {
    const int S = 1 << dgd_class(ng);
    const uint32_t* sp = A.ds_bits + (size_t)A.Ntot * (S - 1) + (size_t)dn0 * S;
    uint32_t* dp = A.bits + (size_t)N * (S - 1) + (size_t)ob * S;
    const int nw = ng * S;
    for (int q = l; q < nw; q += DG_ASM_TPG) dp[q] = sp[q];
  }
  if (A.colidx) {
    const int e_hi = A.ds_rowptr[dn0 + ng_ds];
    int ne = min(min(e_hi - e_lo, A.oedge[g + 1] - oe), E - oe);
    const int shift = ob - (int)dn0;         // dataset-global node id -> batch node id   (|dn0| < 2^31: Ntot is an int)
    for (int k = l; k < ne; k += DG_ASM_TPG) {
      const int c = A.ds_colidx[e_lo + k] + shift;
      A.colidx[oe + k] = c < 0 ? 0 : (c >= N ? N - 1 : c);
    }
  }
}
#endif
#endif
