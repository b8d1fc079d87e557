// dataset.hip -- batches from a PREPARED dataset: the copy kernel of dg_assemble.h as a launch of its
// own, and the planning workgroup (item table + graph schedule of the dense / chain kernels) for batches that need one.
// The C entry points (dgcnn_dataset_prepare, dgcnn_assemble) live in api.hip next to the form selection they depend on.
#include "dg_assemble.h"

// workgroup 0 (when a plan is wanted): the planning workgroup, from the host-provided node prefix sums; the others: 4 graphs each
__global__ void __launch_bounds__(1024)
k_assemble(DgAssemble A, int* __restrict__ dmap) {
  int blk = (int)blockIdx.x;
  if (dmap) {
    if (blk == 0) { dg_prep_dense_plan((int)threadIdx.x, 1024, A.B, A.onode, dmap); return; }
    --blk;
  }
  dg_assemble_body(blk * 1024 + (int)threadIdx.x, A);
}

int dg_launch_assemble(const DgAssemble* A, int32_t* dmap, hipStream_t s) {
  if (!A || A->N <= 0 || A->B <= 0 || A->E < 0 || A->F < 1 || !A->ids || !A->onode || !A->oedge || !A->node_ptr || !A->ds_dinv ||
      !A->dinv || !A->graph_ptr || !A->graph_eptr || !A->err)
    return DGCNN_EINVAL;
  if ((A->rowptr || A->colidx) && (!A->ds_rowptr || (A->E > 0 && !A->ds_colidx))) return DGCNN_EINVAL;
  if (A->bits && !A->ds_bits) return DGCNN_EINVAL;
  const int work = dg_assemble_work(A->N, A->E, A->B, A->colidx != nullptr);
  hipLaunchKernelGGL(k_assemble, dim3(dg_cdiv(work, 1024) + (dmap ? 1 : 0)), dim3(1024), 0, s, *A, dmap);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}
