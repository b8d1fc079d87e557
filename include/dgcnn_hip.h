/*
 * dgcnn_hip.h -- C ABI of libdgcnn_hip.so: the MI355X (gfx950) implementation of the
 * DGCNN forward+backward hot path of leftthomas/DGCNN.
 *
 * The reference has NO native interface of its own: its plugin boundary is the Python
 * nn.Module protocol, `Model(num_features, num_classes).forward(data)`
 * (/root/reference/model.py:9-45, used at /root/reference/train.py:13,37,60,97), and all
 * graph arithmetic is delegated to PyTorch Geometric (model.py:5-6).  This header defines
 * what sits UNDER that boundary in this build: plain pointers and
 * sizes, no torch types, every entry point `extern "C"`.  Each function cites the
 * reference line(s) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - returns 0 on success, a negative DGCNN_E* code on a bad argument or a HIP launch error;
 *   - never allocates device memory, never synchronises, never throws; all buffers are
 *     caller-owned DEVICE memory (the caller is PyTorch-ROCm's allocator); work is enqueued on
 *     `stream` (pass torch.cuda.current_stream().cuda_stream) and is complete, in stream order,
 *     when later work on `stream` runs;
 *   - stateless and re-entrant across streams (the only state is the one-shot, thread-local
 *     profiling/debug request of the measurement helpers at the end of this file);
 *   - fp32 arithmetic, int64 graph indices in (as the reference), int32 indices inside;
 *   - results are run-to-run bit-reproducible: no floating-point atomics anywhere.
 */
#ifndef DGCNN_HIP_H
#define DGCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGCNN_ABI_VERSION 20

/* error codes */
#define DGCNN_OK            0
#define DGCNN_EINVAL       -1   /* bad size / null pointer / unsupported shape */
#define DGCNN_ELAUNCH      -2   /* hipGetLastError() != hipSuccess after a launch */
#define DGCNN_EUNSUPPORTED -3   /* shape outside what this build handles */

/* fixed architecture constants of the reference model (/root/reference/model.py:13-23) */
#define DGCNN_HID     32   /* GCN widths 32/32/32/1            model.py:13-16 */
#define DGCNN_CAT     97   /* 32+32+32+1 concatenated channels model.py:34    */
#define DGCNN_K       30   /* SortAggregation(k=30)            model.py:17    */
#define DGCNN_C5      16   /* Conv1d(1,16,97,97)               model.py:18    */
#define DGCNN_C6      32   /* Conv1d(16,32,5,1)                model.py:19    */
#define DGCNN_KW6      5
#define DGCNN_T5      15   /* after MaxPool1d(2,2)             model.py:20    */
#define DGCNN_T6      11   /* 15-5+1                                           */
#define DGCNN_FLAT   352   /* 32*11 = Linear(352,128) in-feats model.py:21    */
#define DGCNN_HID1   128   /* Linear(352,128)                  model.py:21    */
#define DGCNN_MAX_F  512   /* largest num_features this build accepts */
#define DGCNN_MAX_C   64   /* largest num_classes this build accepts  */
#define DGCNN_FUSED_MAX_NODES 576  /* upper bound of the graph-per-workgroup path (LDS capacity) */

typedef void* dgcnn_stream_t;   /* a hipStream_t */

/* flags of dgcnn_graph_prep / dgcnn_model_forward */
#define DGCNN_FLAG_COALESCED_UNDIRECTED 1
#define DGCNN_FLAG_FORCE_FUSED 2    /* use the graph-per-workgroup kernel whenever the hints allow it */
#define DGCNN_FLAG_FORCE_TILED 4    /* never use it */
#define DGCNN_FLAG_PREPARED    8    /* dgcnn_model_forward: the workspace already holds this batch's graph structure
                                       (dgcnn_model_prepare with the SAME sizes, flags and epoch): skip graph prep */
#define DGCNN_FLAG_AGG_SPARSE  16    /* aggregation: always the CSR gather kernels (gcn.hip) */
#define DGCNN_FLAG_AGG_DENSE   32    /* aggregation: dense per-graph block products on the matrix cores (gcn_dense.hip)
                                       whenever the batch admits it (COALESCED_UNDIRECTED promised, max_nodes in 1..512);
                                       neither flag: the library's cost model picks per batch */
#define DGCNN_FLAG_BF16        64    /* bf16 leg (BASELINE config 3): the pre-scaled linear outputs hs are STORED in bf16
                                       (64-B rows) and X.W^T runs on v_mfma_f32_16x16x32_bf16; sums, activations, the
                                       SortPooling key channel and the whole backward stay fp32.  Dense block form only
                                       (DGCNN_EUNSUPPORTED otherwise).  Not the reference's arithmetic: a secondary leg. */
#define DGCNN_FLAG_CHAIN       128   /* graph-chain kernels (gcn_chain.hip): conv1..conv4 of a graph run inside ONE workgroup, the
                                       pre-scaled linear outputs never leave the CU (one launch instead of four; same for the
                                       backward chain).  Needs what the dense block form needs plus num_features <= 32; this
                                       flag asks for it whenever admissible ... */
#define DGCNN_FLAG_NO_CHAIN    256   /* ... this one forbids it; neither: the library's cost model decides per batch */
#define DGCNN_FLAG_EXCLUSIVE_DEVICE 512 /* dgcnn_pipeline_train_step / _eval_step: the caller promises that nothing else runs on this
                                         * device while a step is in flight (no second process, no other stream with spin-waiting
                                         * kernels).  Only then BOTH phases of the next batch's graph preparation join the one-launch
                                         * kernel of a small batch -- its phase-B workgroups wait, on the device, for the phase-A
                                         * workgroups of the same launch, which is deadlock-free only when no OTHER launch's waiting
                                         * workgroups can hold the compute units the producers still need.  Without the promise
                                         * phase B rides on the step's next launch as before round 4 (same results, ~2 us per step at
                                         * the reference's batch of 50).  A wait that does not end is bounded (~4 s) and reported
                                         * through err[4] / err[6], never as a layout error. */
#define DGCNN_FLAG_INFERENCE 1024      /* (ABI v20) the batch is only ever run FORWARD (the body of the reference's test() loop,
                                        * /root/reference/train.py:57-62; inference): a small batch (<= 256 graphs) whose largest
                                        * graph has 257..512 nodes is then prepared for, and run by, the one-launch evaluation kernel's
                                        * two-tiles-per-wave form instead of the launch-per-layer route a training step of such a batch
                                        * takes.  Part of the preparation's form like the other family flags: give it to the
                                        * preparation AND the forward of a batch (the pipeline calls prepare again when they differ).
                                        * Results: same arithmetic as the chain forward + readout launches. */
#define DGCNN_FUSED_MIN_GRAPHS (1 << 30) /* the fused path is never chosen automatically: the tiled kernels measured
                                            faster at every batch size (profiles/r01_sweep.txt); FORCE_FUSED selects it */
/* The caller PROMISES the edge list is coalesced and undirected: sorted by (source,target), no
 * duplicates, no self loops, every edge present in both directions -- what a TU dataset file (and
 * PyG coalesce/to_undirected) holds, i.e. what the reference's loader feeds model.py:27.  Graph
 * preparation then needs no atomics and no sort (one launch).  The promise is verified on the
 * device; a violation is reported through the error words (never silently mis-computed). */

int dgcnn_version(void);

/* ------------------------------------------------------------------------------------
 * Parameter layout.  All learnable parameters live in ONE flat fp32 buffer (also the
 * gradient all-reduce bucket and the Adam state shape).  Segment order, each start
 * rounded up to a multiple of 4 floats:
 *   0 conv1.lin.weight [32,F]   1 conv1.bias [32]     (GCNConv(F,32)   model.py:13)
 *   2 conv2.lin.weight [32,32]  3 conv2.bias [32]     (GCNConv(32,32)  model.py:14)
 *   4 conv3.lin.weight [32,32]  5 conv3.bias [32]     (GCNConv(32,32)  model.py:15)
 *   6 conv4.lin.weight [1,32]   7 conv4.bias [1]      (GCNConv(32,1)   model.py:16)
 *   8 conv5.weight [16,1,97]    9 conv5.bias [16]     (model.py:18)
 *  10 conv6.weight [32,16,5]   11 conv6.bias [32]     (model.py:19)
 *  12 classifier_1.weight [128,352]  13 classifier_1.bias [128]  (model.py:21)
 *  14 classifier_2.weight [C,128]    15 classifier_2.bias [C]    (model.py:23)
 * dgcnn_param_layout fills offsets[16] (in floats) and returns the padded total length,
 * or a negative error code.
 * ---------------------------------------------------------------------------------- */
#define DGCNN_NUM_PARAM_SEGMENTS 16
int64_t dgcnn_param_layout(int F, int C, int64_t offsets[DGCNN_NUM_PARAM_SEGMENTS]);

/* ------------------------------------------------------------------------------------
 * Workspace.  One caller-owned device arena holds the per-batch graph structure, the
 * activations saved for backward and all temporaries.  dgcnn_workspace_bytes gives its
 * size for a batch of N nodes, E directed edges (self loops included in E are fine),
 * B graphs.  dgcnn_workspace_offset returns the byte offset of a named region (for tests
 * and tools; names listed in DESIGN.md), or -1.
 * ---------------------------------------------------------------------------------- */
int64_t dgcnn_workspace_bytes(int N, int E, int B, int F, int C);
int64_t dgcnn_workspace_offset(const char* name, int N, int E, int B, int F, int C);

/* ------------------------------------------------------------------------------------
 * Graph preparation: replaces `remove_self_loops` (model.py:28) and the structural half
 * of PyG `gcn_norm` that the reference re-runs inside each of its four GCNConv calls
 * (model.py:30-33): drop self loops, count in-degree, dinv = (indeg+1)^-1/2, and build
 * CSR by target (for forward) and CSR by source (for backward), neighbours ascending.
 * Also derives graph_ptr[B+1] (node range of each graph) from the sorted `batch` vector
 * (what PyG `to_dense_batch` derives inside SortAggregation, model.py:35).
 *   edge_index [2,E] int64 row-major (row 0 = source, row 1 = target)   model.py:27
 *   batch      [N]   int64, sorted, values in [0,B)                     model.py:27
 * Outputs: rowptr[N+1], colidx[E], rowptr_t[N+1], colidx_t[E] (int32; only the first
 * rowptr[N] entries of colidx are meaningful), dinv[N] f32, graph_ptr[B+1] int32.
 * scratch: 2*N+B+3 int32.  err_flag: 4 int32; afterwards err_flag[0] != 0 if an edge endpoint is
 * out of range, err_flag[1] != 0 if DGCNN_FLAG_COALESCED_UNDIRECTED was promised but does not hold.
 * ---------------------------------------------------------------------------------- */
int dgcnn_graph_prep(const int64_t* edge_index, int E, const int64_t* batch, int N, int B,
                     int32_t* rowptr, int32_t* colidx, int32_t* rowptr_t, int32_t* colidx_t,
                     float* dinv, int32_t* graph_ptr, int32_t* scratch, int32_t* err_flag, int flags,
                     uint32_t* adj_bits, int32_t* item_table, dgcnn_stream_t stream);

/* Dense per-graph block form of the aggregation (gcn_dense.hip).  For batches of SMALL graphs (every graph
 * <= 512 nodes) whose edge list is coalesced + undirected, graph preparation can additionally emit
 *   adj_bits   [dgcnn_dense_bitmap_words(N)] u32 : a bit-packed adjacency row per node (self bit included)
 *   item_table [dgcnn_dense_table_ints(N,B)] i32 : work items (graph, 128-row group) + 1024 equal-cost shares of them
 * (pass both or neither; needs DGCNN_FLAG_COALESCED_UNDIRECTED).  The aggregation kernels then evaluate
 * (A+I)_g . H_g per graph on the matrix cores instead of gathering one 128-B row per edge. */
typedef struct dgcnn_dense_view {
  int32_t B;                      /* graphs in the batch */
  int32_t reserved_;
  const int32_t* graph_ptr;       /* [B+1] */
  const int32_t* item_table;
  const uint32_t* adj_bits;
} dgcnn_dense_view;
int64_t dgcnn_dense_table_ints(int N, int B);
int64_t dgcnn_dense_bitmap_words(int N);

/* ------------------------------------------------------------------------------------
 * One graph-convolution layer, forward: out = tanh( D~^-1/2 (A+I) D~^-1/2 (x W^T) + b ).
 * Replaces `torch.tanh(self.convN(x, edge_index))` (model.py:30-33; PyG GCNConv:
 * linear without bias first, then gather/scale/scatter-add over edges, then + bias).
 *   x [N,Fin] f32, W [Fout,Fin], bias [Fout], Fout in {32, 1}, out [N,Fout] (row stride Fout)
 *   hs_scratch [N,Fout] : the pre-scaled linear output dinv[j] * (x W^T)[j]; f32, or bf16 with DGCNN_FLAG_BF16
 *   flags : 0 | DGCNN_FLAG_AGG_DENSE (needs `dense`) | DGCNN_FLAG_BF16 (dtype flag of the bf16 leg: hs stored in bf16 --
 *           the linear step itself runs in fp32 here and rounds on store; Fout = 32 and `dense` required)
 *   dense : structures from dgcnn_graph_prep, or NULL (CSR gather kernels)
 * ---------------------------------------------------------------------------------- */
int dgcnn_gcn_fwd(int N, const int32_t* rowptr, const int32_t* colidx, const float* dinv,
                  const float* x, int Fin, const float* W, const float* bias, int Fout,
                  float* out, void* hs_scratch, int flags, const dgcnn_dense_view* dense, dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * One link of the graph-convolution BACKWARD chain: the production kernels of dgcnn_model_backward, stand-alone, so
 * that a per-layer gradient can be checked on its own (what `loss.backward()`, /root/reference/train.py:40, runs for
 * one `torch.tanh(self.convN(...))` of model.py:30-33).  Layer l computed x_l = tanh(A_hat (x_{l-1} W_l^T) + b_l).
 * Input  gas [N,Fout] = dinv[i] * dL/d(pre-activation of layer l)[i]   (= dinv * dL/dx_l * (1 - x_l^2))
 *   Fout = 32, first = 0 (conv2 / conv3 form):  x_prev [N,32] = x_{l-1} (a tanh output), gp_prev [N,32] = gradient that
 *       reaches x_{l-1} from elsewhere (SortPooling's; zeros if none):
 *         gW [32,32] = dL/dW_l ;  gb_prev [32] = dL/db_{l-1} ;
 *         gas_prev [N,32] = dinv * (dL/dx_{l-1} + gp_prev) * (1 - x_prev^2)      (the next link's input)
 *       with ax [N,Fa] (Fa <= 32, the saved A_hat X of an aggregate-first conv1) and gW_af: additionally
 *         gW_af [32,Fa] = dL/dW_1, and no gas_prev (may be NULL)
 *   Fout = 32, first = 1 (conv1, linear-first): x_prev [N,Fin] = raw input; only gW [32,Fin]
 *   Fout = 1  (conv4 form): W [1,32]; gW [32], gb_prev [32], gas_prev [N,32] as above
 *   dense : NULL = CSR gather kernels on rowptr_t / colidx_t (CSR by source); else the dense block kernels
 *   scratch : dgcnn_gcn_bwd_scratch_bytes(N, Fin, Fout) bytes (per-workgroup partial rows; reduced in a fixed order)
 * ----------------------------------------------------------------------------------  * colidx_t may be NULL only with `dense` and first == 0 (conv1's own backward, first != 0, always runs the gather kernel on
 * the transposed CSR); DGCNN_EINVAL otherwise. */
int64_t dgcnn_gcn_bwd_scratch_bytes(int N, int Fin, int Fout);
int dgcnn_gcn_bwd(int N, const int32_t* rowptr_t, const int32_t* colidx_t, const float* dinv, const float* gas, int Fout,
                  const float* W, const float* x_prev, int Fin, int first, const float* gp_prev, float* gas_prev,
                  float* gW, float* gb_prev, const float* ax, int Fa, float* gW_af, const dgcnn_dense_view* dense,
                  void* scratch, int64_t scratch_bytes, dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * SortPooling forward: replaces `self.sort_pool(x, batch)` = PyG SortAggregation(k=30)
 * (model.py:17,35).  Per graph: order nodes by the LAST channel descending (ties: lower
 * node index first -- the reference's tie order is undefined), take the first min(n,k)
 * rows, zero-pad to k, flatten node-major.
 *   x1,x2,x3 [N,32], x4 [N]  (the four tanh(GCN) outputs; channel 96 = x4 is the key)
 *   pooled [B, k*97] f32,  perm [B,k] int32 (global node index, -1 = padding)
 * sortpool_bwd scatters a gradient wrt `pooled` back to dense per-node gradients
 * (zero for unselected nodes): g1,g2,g3 [N,32], g4 [N].
 * ---------------------------------------------------------------------------------- */
int dgcnn_sortpool_fwd(int N, int B, const int32_t* graph_ptr,
                       const float* x1, const float* x2, const float* x3, const float* x4,
                       float* pooled, int32_t* perm, dgcnn_stream_t stream);
int dgcnn_sortpool_bwd(int N, int B, const int32_t* graph_ptr, const int32_t* perm,
                       const float* gpooled, float* g1, float* g2, float* g3, float* g4,
                       dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Whole-model forward: replaces `Model.forward(data)` (model.py:26-45).
 *   params : flat parameter buffer (dgcnn_param_layout)
 *   x [N,F] f32, edge_index [2,E] i64, batch [N] i64          (data.x/.edge_index/.batch)
 *   ws     : workspace arena (dgcnn_workspace_bytes); afterwards holds everything
 *            dgcnn_model_backward needs
 *   logp   : [B,C] f32 log-probabilities (F.log_softmax output, model.py:43)
 *   training != 0 applies Dropout(0.5) (model.py:22,42) with a counter-based mask drawn
 *            from `seed` (mask is exported in the workspace region "drop_mask" [B,128] u8)
 *   flags  : 0 or DGCNN_FLAG_COALESCED_UNDIRECTED
 *   max_nodes: host-known upper bound of the node count of any single graph of the batch
 *            (PyG's collate knows it; 0 = unknown).  With max_nodes/max_edges given and
 *            DGCNN_FLAG_FORCE_FUSED, a batch whose largest graph fits the LDS plan runs the
 *            graph-per-workgroup fused kernel (whole forward in one launch, activations and adjacency
 *            in LDS); otherwise the tiled kernels (measured faster at every batch size, hence the default).
 *            Both give bit-identical results.  A hint that is too small is detected on the device and
 *            reported through the error words.
 *   max_edges: host-known upper bound of the directed-edge count of any single graph (0 = unknown);
 *            lets the fused kernel also keep each graph's neighbour ids in LDS when they fit.
 *   epoch  : non-zero tag of this call.  Input errors are reported WITHOUT any host sync or
 *            memset through the workspace region "err" (8 x u32): the call is in error iff
 *            err[k] == epoch && err[k+2] == ~epoch  (k = 0: edge endpoint out of range,
 *            k = 1: COALESCED_UNDIRECTED promised but violated, k = 4: an in-launch wait of the
 *            pipelined preparation timed out -- DGCNN_FLAG_EXCLUSIVE_DEVICE promised on a shared
 *            device; the batch's structures are incomplete).  The caller checks at its next
 *            natural sync point.
 * ---------------------------------------------------------------------------------- */
int dgcnn_model_forward(int N, int E, int B, int F, int C, const float* params,
                        const float* x, const int64_t* edge_index, const int64_t* batch,
                        void* ws, float* logp, int training, uint64_t seed, int flags, int max_nodes,
                        int max_edges, uint32_t epoch, dgcnn_stream_t stream);
/* Which kernel families dgcnn_model_forward / the training step take for a batch of these sizes, flags and max_nodes
 * (a pure function of host-known numbers): DGCNN_FORM_CHAIN = conv1..conv4 as ONE graph-chain launch (gcn_chain.hip),
 * DGCNN_FORM_DENSE = dense per-graph block kernels for the per-layer forward (when not chained) and the backward
 * (gcn_dense.hip); neither = CSR gather kernels (gcn.hip).  Negative: error code. */
#define DGCNN_FORM_DENSE 1
#define DGCNN_FORM_CHAIN 2
#define DGCNN_FORM_CHAIN_TAIL 4   /* a TRAINING step with labels runs chain forward + readout forward + readout backward as one launch */
#define DGCNN_FORM_STEP 8         /* ... and the whole GCN backward of every graph in that same launch (round 4): the step is
                                   * k_chain_readout_tail + k_wgrad.  (DGCNN_STEP_KERNEL=0 in the environment keeps the round-3 form.) */
#define DGCNN_FORM_EVAL 16        /* a forward WITHOUT an in-launch backward -- dgcnn_model_forward, dgcnn_model_eval_step,
                                   * dgcnn_pipeline_eval_step: /root/reference/model.py:26-45 as called from train.py:57-62 -- runs
                                   * chain forward + readout forward (+ the batch's metrics) as ONE launch, k_chain_readout_eval (round 5) */
int dgcnn_forward_form(int N, int E, int B, int F, int flags, int max_nodes);
/* Test / measurement switch of DGCNN_FORM_STEP (process-wide; the environment variable DGCNN_STEP_KERNEL=0 sets the initial
 * value): on = 0 keeps the GCN backward of small training batches in launches of its own (the round-3 form), on = 1 restores the
 * default.  Returns the previous setting.  Replaces nothing of the reference (/root/reference/train.py:40 is one backward). */
int dgcnn_step_kernel_enable(int on);
/* Test / measurement switch of DGCNN_FORM_EVAL (process-wide): on = 0 keeps the chain forward and the readout forward of small
 * batches as two launches (+ k_eval_metrics where labels are given) -- the round-4 form; on = 1 restores the round-5 default; on = 2
 * (ABI v20) additionally takes the one-launch form for chain-form batches with a graph of 257..512 nodes without
 * DGCNN_FLAG_INFERENCE (e.g. under DGCNN_FLAG_CHAIN).  Returns the previous setting.  Replaces nothing of the reference (/root/reference/model.py:26-45 is one forward). */
int dgcnn_eval_kernel_enable(int on);
/* Test / measurement switch of the eight-lanes-per-node ("narrow") gather kernels that the launch-per-layer route of
 * dgcnn_model_forward / dgcnn_model_backward takes for sparse batches of many nodes (more than 4096 nodes, mean in-degree <= 8:
 * DD at the reference's batch of 50, /root/reference/model.py:30-33 + train.py:40): on = 0 keeps the wave-per-node kernels, whose
 * results the narrow forms reproduce bit for bit; on = 1 restores the default; on = 2 (ABI v20, the default since round 6; DGCNN_NARROW_GATHER in the environment sets the initial value) also takes
 * eight-lanes-per-node forms of conv4's two SCALAR gathers (k_gcn_fwd1n / k_gcn_bwd1n: bit-identical for rows of <= 8 neighbours,
 * summation-order rounding beyond).  Returns the previous setting (process-wide).
 * Do not toggle it between dgcnn_model_forward and dgcnn_model_backward of the SAME batch: on this route the switch also decides
 * whether the readout backward leaves the SortPooling-gradient slabs sparse (flag word per node), and the backward's layer
 * kernels must read them the way the forward half of the step wrote them. */
int dgcnn_narrow_gather_enable(int on);

/* Graph preparation of dgcnn_model_forward as a call of its own, writing into the workspace `ws`: everything of the
 * forward that depends on the batch only, not on the parameters -- CSR by target / by source, dinv, graph ranges and
 * (for F <= 32, where conv1 runs aggregate-first) the pre-scaled raw features dinv*x, which is why `x` is an input
 * (may be NULL only when F > 32) -- and, when flags / max_nodes select the dense block form of the aggregation, the
 * bit-packed adjacency.  Follow with dgcnn_model_forward(..., flags | DGCNN_FLAG_PREPARED, same max_nodes, same epoch). */
int dgcnn_model_prepare(int N, int E, int B, int F, int C, const float* x, const int64_t* edge_index,
                        const int64_t* batch, void* ws, int flags, int max_nodes, uint32_t epoch, dgcnn_stream_t stream);
int dgcnn_fused_max_nodes(int F);   /* largest max_nodes the fused path accepts for F input features */
int dgcnn_fused_fits(int max_nodes, int max_edges, int F);   /* 1 if such a batch fits the fused LDS plan */

/* ------------------------------------------------------------------------------------
 * Whole-model backward: what `loss.backward()` (/root/reference/train.py:40) executes for
 * the ops of Model.forward.  Exactly one of (glogp, y) must be non-null:
 *   glogp [B,C] f32 : upstream gradient wrt the log-probabilities (drop-in autograd path)
 *   y     [B]   i64 : labels; the kernel then uses glogp = d/dlogp of nn.NLLLoss() mean
 *                     (train.py:39,98) scaled by `loss_scale` (1/B_global for data parallel;
 *                     pass 0 for 1/B) and writes per-graph loss/correct into ws ("lossv").
 *   training: the same flag the matching dgcnn_model_forward was called with (dropout scale)
 *   grads : flat gradient buffer (same layout as params); every parameter entry is OVERWRITTEN
 *           (not accumulated); the <=3-float alignment gaps between segments are never touched,
 *           so hand in a buffer whose gaps are zero (e.g. allocated zeroed once)
 *   metrics: optional (label mode only) 2-float device accumulator: metrics[0] += sum_b loss_b,
 *           metrics[1] += #correct -- replaces the two `.item()` syncs per batch of train.py:44-45
 *   flags, max_nodes: the values the matching dgcnn_model_forward was given (they select the aggregation form,
 *           whose structures the forward left in the workspace; with DGCNN_FLAG_COALESCED_UNDIRECTED the preparation
 *           keeps ONE CSR -- an undirected edge list's CSR by source is its CSR by target -- and the backward reads it)
 * ---------------------------------------------------------------------------------- */
int dgcnn_model_backward(int N, int E, int B, int F, int C, const float* params,
                         const float* x, void* ws, const float* logp,
                         const float* glogp, const int64_t* y, float loss_scale, int training,
                         float* grads, float* metrics, int flags, int max_nodes, dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Backward + optimizer in one call (single-GPU training step): as dgcnn_model_backward in label
 * mode, and every weight-gradient lane immediately applies the Adam update of its element
 * (`optimizer.step()`, train.py:41, torch.optim.Adam defaults semantics; `step` is 1-based), so no
 * separate optimizer launch exists.  `grads` still receives the gradient (`zero_grad` is moot: every
 * call overwrites it).  Not for data-parallel runs (the all-reduce sits between gradient and update).
 * ---------------------------------------------------------------------------------- */
int dgcnn_model_backward_step(int N, int E, int B, int F, int C, float* params, const float* x, void* ws,
                              const float* logp, const int64_t* y, float loss_scale, int training,
                              float* grads, float* metrics, float* exp_avg, float* exp_avg_sq,
                              int64_t step, float lr, float beta1, float beta2, float eps,
                              int flags, int max_nodes, dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Adam step over the flat buffer: replaces `optimizer.step(); optimizer.zero_grad()`
 * (train.py:41-42) for torch.optim.Adam defaults (train.py:99: lr 1e-3, betas .9/.999,
 * eps 1e-8, no weight decay, no amsgrad).  `step` is the 1-based step count.
 * zero_grads != 0 also clears `grads`.
 * ---------------------------------------------------------------------------------- */
int dgcnn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                    int64_t n, int64_t step, float lr, float beta1, float beta2, float eps,
                    int zero_grads, dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Pipelined training step: the body of the reference's `train()` loop (/root/reference/train.py:36-45)
 * for one batch as ONE call, with the NEXT batch's graph preparation overlapped with it.
 *
 * dgcnn_step_args describes one batch (fields as the same-named arguments of dgcnn_model_forward /
 * dgcnn_model_backward[_step]); host code fills it once per batch object and only touches seed / epoch /
 * step between calls.  exp_avg == NULL selects "forward + backward only" (data parallel: the caller
 * all-reduces `grads` and then calls dgcnn_adam_step); otherwise the Adam update is fused as in
 * dgcnn_model_backward_step.
 *
 * dgcnn_pipeline_train_step(h, cur, next, stream), everything on `stream`:
 *   1. forward + backward (+Adam) of `cur`.  With DGCNN_FLAG_PREPARED in cur->flags (the host states that `cur`
 *      is the batch the previous call was given as `next`: same ws, N, E, B -- anything else is DGCNN_EINVAL)
 *      the step skips its own graph preparation, else it prepares in-stream first.  The preparation's form (bit-packed
 *      adjacency or not, who verifies the reverse edges) follows from next->flags; if cur->flags names ANOTHER kernel family
 *      than the flags it was prepared under (COALESCED_UNDIRECTED, FORCE_*, AGG_*, CHAIN / NO_CHAIN, BF16 differ), the step
 *      prepares again under cur->flags -- no promise goes unverified (an unverified one would be read as its own transpose).
 *   2. if `next` != NULL: its graph structure (what dgcnn_model_prepare builds: CSR by target / by source, dinv,
 *      graph ranges -- a function of the batch only, never of the weights) is built DURING this step, by extra
 *      workgroups appended to two of the step's launches (the SortPooling+tail forward carries phase A; phase B
 *      rides on the tail backward or, when forward and backward of the tail ran as one launch, on the step's last
 *      launch, the weight-gradient kernel): those launches leave most of the 256 CUs idle at the reference's batch
 *      size, the preparation runs on the idle ones.  Same stream, no events.  next->ws must differ from cur->ws.  (General edge lists without
 *      DGCNN_FLAG_COALESCED_UNDIRECTED, or a step that took the fused forward, are prepared in-stream after
 *      the step instead: same results, no overlap.)
 * Every batch's preparation still runs exactly once inside the training loop; only its position changes.
 * No host synchronisation.  Results are bit-identical to the unpipelined calls.
 * ---------------------------------------------------------------------------------- */
struct dgcnn_dataset;
typedef struct dgcnn_step_args {
  int32_t N, E, B, F, C;
  int32_t training;            /* dropout on/off, as model.train() / model.eval() */
  int32_t flags;               /* DGCNN_FLAG_* layout promises / path overrides */
  int32_t max_nodes, max_edges;
  uint32_t epoch;              /* error-word tag of this forward (non-zero, unique per forward on this ws) */
  uint64_t seed;               /* dropout stream of this step */
  int64_t step;                /* Adam step, 1-based (ignored when exp_avg == NULL) */
  float lr, beta1, beta2, eps; /* Adam hyper-parameters */
  float loss_scale;            /* 0 = 1/B, else 1/B_global */
  int32_t reserved_;
  float* params;               /* flat parameters (updated in place when exp_avg != NULL) */
  const float* x;              /* [N,F] */
  const int64_t* edge_index;   /* [2,E] */
  const int64_t* batch;        /* [N] */
  const int64_t* y;            /* [B] labels */
  void* ws;                    /* workspace of dgcnn_workspace_bytes(N,E,B,F,C) bytes */
  float* logp;                 /* [B,C] out */
  float* grads;                /* flat gradient out */
  float* metrics;              /* optional 2-float accumulator */
  float* exp_avg;              /* Adam moments, or NULL */
  float* exp_avg_sq;
  /* batch drawn from a PREPARED dataset (below; all NULL otherwise).  Then edge_index is ignored (may be NULL), and x [N,F],
   * y [B] and batch [N] (optional, may be NULL) are caller-allocated buffers that the ASSEMBLY fills before the forward
   * reads them; N, E, max_nodes, max_edges are the host-known sums / maxima of the chosen graphs' sizes. */
  const struct dgcnn_dataset* ds;
  const int64_t* ds_ids;       /* [B]   graph ids of the batch (device) */
  const int32_t* ds_onode;     /* [B+1] exclusive prefix sums of their node counts (device) */
  const int32_t* ds_oedge;     /* [B+1] ... of their directed-edge counts (device) */
} dgcnn_step_args;

/* Evaluation step, the body of the reference's `test()` loop (/root/reference/train.py:59-64), one call: forward in
 * eval mode (no dropout) into a->logp and, when a->y and a->metrics are given, metrics[0] += NLLLoss-mean of the batch
 * (sum of the per-graph losses times a->loss_scale; 0 = 1/B, data parallel: 1/B_global so that the ranks' sums add up to
 * the global mean), metrics[1] += number of correct argmax predictions.  Uses the same argument block as the training
 * step (optimizer fields ignored). */
int dgcnn_model_eval_step(const dgcnn_step_args* a, dgcnn_stream_t stream);

/* The pipeline object is the ONE place where this library owns device-side resources: from the first training step of 512
 * graphs or more on, a side stream (lowest priority, non-blocking) and two events.  The next batch's graph preparation then
 * runs on that stream, forked from `stream` inside the step and joined to it before the call's last launch completes its
 * dependencies: everything the caller enqueues on `stream` afterwards sees the prepared structure, and nothing enqueued
 * before the call can be overtaken.  Smaller steps carry the preparation as spare workgroups of their own launches.
 * dgcnn_pipeline_destroy synchronises the side stream and releases it. */
int dgcnn_pipeline_create(void** handle);     /* one per training loop: remembers which workspace holds a prepared batch */
int dgcnn_pipeline_destroy(void* handle);
int dgcnn_pipeline_train_step(void* handle, const dgcnn_step_args* cur, const dgcnn_step_args* next,
                              dgcnn_stream_t stream);
/* Pipelined evaluation step: dgcnn_model_eval_step(cur) -- the body of the reference's `test()` loop,
 * /root/reference/train.py:57-64 -- with the NEXT batch's graph preparation (or its assembly from a prepared dataset)
 * overlapped exactly as in dgcnn_pipeline_train_step: rider workgroups of the evaluation launch where the batch takes the
 * one-launch form (DGCNN_FORM_EVAL), launches behind it otherwise.  cur->flags & DGCNN_FLAG_PREPARED says that `cur` is the
 * batch the previous pipelined call (training or evaluation) prepared.  next == NULL: nothing is prepared ahead. */
int dgcnn_pipeline_eval_step(void* handle, const dgcnn_step_args* cur, const dgcnn_step_args* next,
                             dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Mini-batch assembly on the device: what the reference's `DataLoader(data_set[idx], batch_size, shuffle)`
 * (/root/reference/train.py:108-109, PyG collate) produces on the host for every batch -- x rows of the chosen graphs
 * concatenated, their edge lists shifted by each graph's node offset, the `batch` vector, the labels -- from a dataset
 * resident in device memory:
 *   x_all [Ntot,F] f32, ei_all [2,Etot] i64 (GRAPH-LOCAL node ids), node_ptr / edge_ptr [G+1] i64, y_all [G] i64
 * Per batch: graph_ids [B] i64 (device) and the exclusive prefix sums of the chosen graphs' node / edge counts,
 * out_node_ptr / out_edge_ptr [B+1] i64 (device; N = out_node_ptr[B], E = out_edge_ptr[B] are passed by value).
 * Outputs (caller-allocated): x [N,F], edge_index [2,E] i64, batch [N] i64, y [B] i64.  One launch, pure data
 * movement (bit-exact); concatenating coalesced undirected graphs keeps the union coalesced.
 * ---------------------------------------------------------------------------------- */
int dgcnn_collate(int B, int F, int64_t N, int64_t E, int64_t Etot, const int64_t* graph_ids, const int64_t* out_node_ptr,
                  const int64_t* out_edge_ptr, const float* x_all, const int64_t* ei_all, const int64_t* node_ptr,
                  const int64_t* edge_ptr, const int64_t* y_all, float* x, int64_t* edge_index, int64_t* batch,
                  int64_t* y, dgcnn_stream_t stream);

/* dgcnn_collate with the per-batch bookkeeping done here instead of in the host language: from the graph ids of the
 * batch (HOST array) and the dataset's per-graph node / edge counts (HOST arrays) it builds the prefix sums in the
 * caller's pinned staging buffer meta_host [3B+2], uploads them asynchronously to meta_dev [3B+2] and launches the
 * assembly.  out_sizes (host, 4 values) receives N, E, max nodes per graph, max edges per graph of the batch; if N or E
 * exceeds the given buffer capacities the call returns DGCNN_EUNSUPPORTED with out_sizes filled and launches nothing.
 * ev_uploaded (a hipEvent_t from dgcnn_event_create, or NULL) guards the reuse of meta_host across calls.
 * ids_dev (optional): the same B ids in DEVICE memory (e.g. a slice of the epoch's permutation, uploaded once per
 * epoch); with it and B <= 256 nothing is uploaded per batch -- the kernel rebuilds the prefix sums in LDS. */
int dgcnn_collate_ids(int B, int F, const int64_t* ids_host, const int64_t* ids_dev, const int64_t* nodes_per_graph_host,
                      const int64_t* edges_per_graph_host, int64_t num_graphs, int64_t* meta_host, int64_t* meta_dev,
                      void* ev_uploaded, int64_t Etot, const float* x_all, const int64_t* ei_all, const int64_t* node_ptr,
                      const int64_t* edge_ptr, const int64_t* y_all, int64_t cap_nodes, int64_t cap_edges, float* x,
                      int64_t* edge_index, int64_t* batch, int64_t* y, int64_t* out_sizes, dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * PREPARED dataset.  Everything graph preparation derives from a batch -- the CSR by target of the
 * block-diagonal adjacency, dinv = (indeg+1)^-1/2 (the structural half of PyG gcn_norm, which the reference re-runs in each
 * of its four GCNConv calls of EVERY batch of EVERY epoch: /root/reference/model.py:30-33), the pre-scaled features dinv*x, the
 * bit-packed adjacency rows of the dense / chain kernels -- is a function of the graph alone, because a batch is a disjoint
 * union (/root/reference/train.py:108-109).  dgcnn_dataset_prepare builds it ONCE for all G graphs of a dataset resident in
 * device memory, by running the SAME preparation kernels over the dataset as one block-diagonal batch (so a graph's numbers
 * are bit for bit the ones a per-batch preparation produces, and the layout promise is verified once instead of per batch);
 * dgcnn_assemble then makes a batch of B chosen graphs by a copy with offset adds -- no int64 edge list is built or read.
 *   node_ptr [G+1] i64: first dataset node of each graph; y [G] i64 labels; x [Ntot,F] f32 raw features      (inputs)
 *   rowptr [Ntot+1] i32, colidx [Etot] i32 (dataset-global ids, ascending per row), dinv [Ntot] f32,
 *   xs [Ntot,F] f32 (F <= 32, else NULL), adj_bits [dgcnn_dense_bitmap_words(Ntot)] u32 (or NULL: no graph of
 *   the dataset may then take a bitmap form -- pass NULL only for datasets whose batches never do)            (outputs)
 * dgcnn_dataset_prepare inputs: edge_index_global [2,Etot] i64 with DATASET-global node ids (graph-local id + node_ptr of
 * its graph), batch_all [Ntot] i64 (graph of each node); scratch: 2*(G+1) int32; err4: 4 int32 words, non-zero [0]/[1]
 * afterwards = node id out of range / edge list not coalesced-undirected-block-diagonal (read them once, after a sync).
 * Needs DGCNN_FLAG_COALESCED_UNDIRECTED (TU dataset files are; general edge lists stay on the per-batch path).
 * Graphs above 512 nodes get no bitmap rows (their batches take the CSR kernels, exactly as on the per-batch path).
 * ---------------------------------------------------------------------------------- */
typedef struct dgcnn_dataset {
  int64_t G, Ntot, Etot;
  int32_t F, reserved_;
  const int64_t* node_ptr;
  const int64_t* y;
  const float* x;
  int32_t* rowptr;
  int32_t* colidx;
  float* dinv;
  float* xs;
  uint32_t* adj_bits;
} dgcnn_dataset;
int dgcnn_dataset_prepare(const dgcnn_dataset* ds, const int64_t* edge_index_global, const int64_t* batch_all,
                          int32_t* scratch, int32_t* err4, int flags, dgcnn_stream_t stream);
/* Batch assembly from a prepared dataset into the workspace `ws` of dgcnn_workspace_bytes(N,E,B,F,C) bytes -- what
 * dgcnn_model_prepare leaves there for the same graphs in the same order, bit for bit (tests/test_prepared_dataset.py) --
 * plus the batch's x [N,F], y [B] and (optional) batch [N] buffers.  ids / onode / oedge as in dgcnn_step_args.  flags,
 * max_nodes, epoch: the values the following dgcnn_model_forward(..., flags | DGCNN_FLAG_PREPARED, ...) is given; flags must
 * carry DGCNN_FLAG_COALESCED_UNDIRECTED.  A graph id outside [0,G) or prefix sums that do not match the ids are reported
 * through the workspace's error words like any other input error.  One launch (two when the batch needs a graph schedule). */
int dgcnn_assemble(const dgcnn_dataset* ds, int B, int N, int E, int C, const int64_t* ids, const int32_t* onode,
                   const int32_t* oedge, void* ws, float* x, int64_t* batch, int64_t* y, int flags, int max_nodes,
                   uint32_t epoch, dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Data parallel, one-shot exchange (the reference has no multi-GPU code, train.py:75-79): the ranks'
 * flat gradients live in peer-mapped fine-grained device memory and ONE kernel per rank sums them in rank order and
 * applies Adam to its replica -- it replaces `all_reduce(grads)` + dgcnn_adam_step (= `optimizer.step()`, train.py:41).
 *   dgcnn_peer_alloc : allocate `bytes` of exchange memory (zeroed) and return its 64-byte IPC handle; the owner lays out
 *                      [flag u32 x16 | gradient buffer 0 | gradient buffer 1] (buffers alternate by step parity)
 *   dgcnn_peer_open  : map a peer's handle (another process: another GPU of the node, or the same GPU)
 *   dgcnn_allreduce_adam_step: peer_grads[r] / peer_flags[r] (HOST arrays of device pointers, r < world; entry `rank` is
 *                      the caller's own) -- publishes `tag` (non-zero, strictly increasing per step) in the own flag,
 *                      waits (bounded) for every peer's tag, g = sum_r peer_grads[r] in rank order, Adam as
 *                      dgcnn_adam_step.  grad_sum_out (optional) receives g.  err[0] = tag if a peer never arrived.
 * The own gradient must have been written by earlier work on `stream`.
 * ---------------------------------------------------------------------------------- */
int dgcnn_peer_alloc(int64_t bytes, void** dev_ptr, void* ipc_handle64);
/* bound of the exchange kernel's waits (default 20 000 ms).  The timeout verdict of a step is taken once per rank and agreed
 * by all ranks (a step that any rank gave up on is applied by NO rank); err[0] = tag on every rank afterwards. */
int dgcnn_peer_set_timeout_ms(int ms);
/* 1 if the last dgcnn_peer_alloc of this process got fine-grained memory, 0 if it fell back to coarse-grained memory
 * (fine between processes of ONE device; across devices the flag polling is not guaranteed coherent). */
int dgcnn_peer_last_alloc_finegrained(void);
int dgcnn_peer_open(const void* ipc_handle64, void** dev_ptr);
int dgcnn_peer_close(void* dev_ptr);
int dgcnn_peer_free(void* dev_ptr);
int dgcnn_allreduce_adam_step(int world, int rank, const float* const* peer_grads, unsigned int* const* peer_flags,
                              uint32_t tag, float* params, float* exp_avg, float* exp_avg_sq, float* grad_sum_out, int64_t n,
                              int64_t step, float lr, float beta1, float beta2, float eps, uint32_t* err,
                              dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Metrics (stand-alone form of the `metrics` argument above): folds the per-graph loss /
 * correct flags left in the workspace by dgcnn_model_backward (label mode) into
 * metrics[0] += sum_b loss_b (already scaled by loss_scale), metrics[1] += #correct.
 * ---------------------------------------------------------------------------------- */
int dgcnn_accumulate_metrics(int B, const void* ws, int N, int E, int F, int C,
                             float* metrics, dgcnn_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Measurement helpers (used by bench.py for the roofline line; not part of the data path).
 * dgcnn_profile_next_forward arms a ONE-SHOT, thread-local request: the next
 * dgcnn_model_forward issued by this thread records ev_start / ev_stop (hipEvent_t created
 * with dgcnn_event_create) on its stream immediately around the `which`-th 32-wide
 * aggregation launch (0 = conv1, 1 = conv2, 2 = conv3), or around the single fused forward kernel
 * when that path runs.  The other calls are thin wrappers
 * of hipEventCreate / hipEventRecord / hipEventSynchronize+hipEventElapsedTime /
 * hipEventDestroy so the bench does not need a second HIP binding.
 * ---------------------------------------------------------------------------------- */
int dgcnn_profile_next_forward(int which, void* ev_start, void* ev_stop);
int dgcnn_event_create(void** ev);
int dgcnn_event_record(void* ev, dgcnn_stream_t stream);
int dgcnn_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int dgcnn_event_destroy(void* ev);
/* debugging aid: subsequent fused forward launches of this thread store clock64() stamps of workgroup 0's
 * phases into the given device buffer of 16 x u64 (NULL switches it off) */
int dgcnn_debug_phase_clocks(void* dev_u64x16);

#ifdef __cplusplus
}
#endif
#endif /* DGCNN_HIP_H */
