// peer.hip -- one-shot gradient all-reduce + Adam over peer-mapped buffers (data parallel).
//
// The reference is single-device (/root/reference/train.py:75-79).  Data parallel, the step needs ONE exchange: the sum of
// the ranks' flat gradients (208 KB), followed by the identical Adam update on every replica (train.py:41).  Adam must
// precede the next forward, so the exchange cannot hide behind compute; what can go is the collective library's call
// overhead and the separate optimizer launch (+12 us at one rank).  Here every rank's gradient buffer
// lives in fine-grained device memory that all ranks of the node map (hipIpc handles; xGMI peer access between GPUs, the
// same mapping between two processes of one GPU), and ONE kernel per rank
//     publishes "my gradient of step s is complete"      (system-scope release store of the step tag; the gradient was
//                                                          written by the previous kernel of this stream)
//     waits for the same tag of every peer                (bounded poll, acquire; the timeout verdict is agreed by all ranks)
//     sums the R gradients in RANK ORDER                  (every rank forms the identical fp32 sum: replicas stay bit-equal)
//     applies Adam to its replica                         (same arithmetic as k_adam, tail.hip)
// Gradient buffers are double-buffered by step parity: a rank that runs ahead writes step s+1's gradient into the other
// buffer, and cannot reach step s+2 before every peer has published s+1, i.e. finished reading step s.
// No hardware here has more than one GPU: the 2-process-on-one-GPU test (tests/test_dist_gloo.py) checks the kernel
// bit for bit against the all_reduce + dgcnn_adam_step route; no multi-GPU timing exists (DESIGN.md §5 says so).
#include "dg_common.h"

#define DG_PEER_MAX 16
struct DgPeers {
  const float* grad[DG_PEER_MAX];
  unsigned int* flag[DG_PEER_MAX];       // each rank's flag block (64 B): see the word list below
};
// flag block words (fine-grained, mapped by every rank):
//   [0] READY   : tag of the newest gradient this rank has completed
//   [1] DECIDED : tag of the newest step for which this rank has decided go / abort
//   [2] GO      : this rank's own grid-wide verdict for the step, tag | (abort << 31) -- polled by its other workgroups
//   [3] ABORTED : tag of the FIRST step this rank aborted (0: never) -- sticky: written once, never cleared; a peer aborts
//                 every step with tag >= it.  (A word holding the NEWEST aborted tag had a hole: a rank that aborted `tag` and
//                 then `tag + 1` overwrote it, and a very late peer at `tag` saw `tag + 1 != tag` and applied the step.)
#define DG_PW_READY 0
#define DG_PW_DECIDED 1
#define DG_PW_GO 2
#define DG_PW_ABORTED 3

static unsigned int g_peer_spins = 20u * 1000u * 1000u;      // bounded wait, ~1 us per spin: 20 s (dgcnn_peer_set_timeout_ms)
static int g_peer_last_finegrained = 0;

// The wait is bounded (a lost peer must not hang the GPU), and the timeout decision is taken ONCE PER STEP AND AGREED BY ALL
// RANKS: workgroup 0 of every rank waits for the peers' gradients, publishes its verdict, waits for the peers' verdicts,
// and aborts if ANY rank aborted (a late peer that finds our gradient in place still learns that we gave up on the step,
// and gives up too); its other workgroups take the verdict from workgroup 0.  On abort nobody touches parameters or Adam
// state -- replicas stay identical -- and err[0] carries the tag on every rank (Trainer.read_metrics raises everywhere).
__global__ void __launch_bounds__(256)
k_allreduce_adam(DgPeers P, int world, int rank, unsigned int tag, float* __restrict__ params, float* __restrict__ m,
                 float* __restrict__ v, float* __restrict__ gsum_out, int64_t n, float lr, float b1, float b2, float eps,
                 float bc1, float bc2_sqrt, unsigned int* __restrict__ err, unsigned int max_spins) {
  __shared__ int ok;
  // this thread's FOUR elements.  Everything local -- the own gradient (written by the previous kernel of this stream) and the
  // optimizer state -- is requested BEFORE the flag protocol: those loads land while workgroup 0 talks to the peers (two
  // cross-device flag round trips), only the peers' gradients wait for the verdict.  (One element per thread meant 407
  // workgroups polling the GO word of a 104 k-parameter model; four per thread: 102.)
  // (READY goes out FIRST: a release store waits for every earlier load of its wave, the prefetch below must not delay it)
  if (blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(P.flag[rank] + DG_PW_READY, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const bool live = i4 < n;                                        // (n is a multiple of 4: dg_param_layout)
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t ic = live ? i4 : 0;
  const f4 gown = __builtin_nontemporal_load(reinterpret_cast<const f4*>(P.grad[rank] + ic));
  const f4 m0 = *reinterpret_cast<const f4*>(m + ic), v0 = *reinterpret_cast<const f4*>(v + ic),
           p0 = *reinterpret_cast<const f4*>(params + ic);
  if (threadIdx.x == 0) {
    unsigned int* mine = P.flag[rank];
    int good = 1;
    if (blockIdx.x == 0) {
      auto wait_for = [&](int word) {        // every peer's `word` has reached this step's tag (tags only grow)
        for (int r = 0; r < world; ++r) {
          if (r == rank) continue;
          unsigned int spins = 0;
          // (RELAXED polls, one acquire fence behind the last of them: an acquire load invalidates the caches on EVERY poll --
          //  under the other workgroups' operand loads; measured inside the training kernel's launch)
          while ((int)(__hip_atomic_load(P.flag[r] + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - tag) < 0) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > max_spins) return false;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope
        return true;
      };
      auto mark_aborted = [&]() {            // sticky: the first aborted tag stays (only this workgroup ever writes the word)
        if (__hip_atomic_load(mine + DG_PW_ABORTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
          __hip_atomic_store(mine + DG_PW_ABORTED, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      };
      auto aborted_by = [&](int r) {         // rank r gave up on this step or on an earlier one
        const unsigned int a = __hip_atomic_load(P.flag[r] + DG_PW_ABORTED, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        return a != 0u && (int)(a - tag) <= 0;
      };
      if (aborted_by(rank)) good = 0;        // an earlier step of ours was aborted: nothing is applied any more (the host refuses too)
      if (good && !wait_for(DG_PW_READY)) good = 0;
      if (!good) mark_aborted();             // published BEFORE the verdict word: whoever sees DECIDED sees it
      __hip_atomic_store(mine + DG_PW_DECIDED, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (!wait_for(DG_PW_DECIDED)) {
        // a peer's verdict did not arrive: peers that already read ours may apply this step while we do not -- the sticky
        // word makes every rank abort every LATER step, and the host refuses further steps once it has seen err[0]
        good = 0; mark_aborted();
      }
      for (int r = 0; r < world && good; ++r)
        if (r != rank && aborted_by(r)) good = 0;
      if (!good) err[0] = tag;
      // (SYSTEM scope like the peer words: the other workgroups go on to read PEER memory behind this acquire, and whether an
      //  agent-scope acquire orders that for fine-grained memory of another device has never been exercised -- no multi-GPU box)
      __hip_atomic_store(mine + DG_PW_GO, good ? tag : (tag | 0x80000000u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      unsigned int spins = 0, g;
      while (((g = __hip_atomic_load(mine + DG_PW_GO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) & 0x7fffffffu) != tag) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > 4u * max_spins + 1024u) { g = 0x80000000u; err[1] = tag; break; }    // (workgroup 0 always decides first)
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");        // (system scope: the peers' gradients are read behind it)
      good = (g >> 31) ? 0 : 1;
    }
    ok = good;
  }
  __syncthreads();
  if (!ok || !live) return;
  f4 g = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < world; ++r)                              // fixed rank order on every rank
    g += r == rank ? gown : __builtin_nontemporal_load(reinterpret_cast<const f4*>(P.grad[r] + i4));
  if (gsum_out) *reinterpret_cast<f4*>(gsum_out + i4) = g;
  f4 mo, vo, po;
#pragma unroll
  for (int k = 0; k < 4; ++k) {      // (the pinned operation sequence of k_adam: the two routes stay bit-identical)
    float mi, vi, pi;
    dg_adam_elem(g[k], m0[k], v0[k], p0[k], lr / bc1, b1, b2, eps, bc2_sqrt, mi, vi, pi);
    mo[k] = mi; vo[k] = vi; po[k] = pi;
  }
  *reinterpret_cast<f4*>(m + i4) = mo; *reinterpret_cast<f4*>(v + i4) = vo; *reinterpret_cast<f4*>(params + i4) = po;
}

extern "C" {

int dgcnn_peer_set_timeout_ms(int ms) {
  if (ms < 1) return DGCNN_EINVAL;
  g_peer_spins = ms > 2000000 ? 2000000000u : (unsigned int)ms * 1000u;
  return DGCNN_OK;
}
int dgcnn_peer_last_alloc_finegrained(void) { return g_peer_last_finegrained; }

int dgcnn_peer_alloc(int64_t bytes, void** dev_ptr, void* ipc_handle64) {
  if (bytes <= 0 || !dev_ptr || !ipc_handle64) return DGCNN_EINVAL;
  static_assert(sizeof(hipIpcMemHandle_t) <= 64, "handle fits the 64-byte slot");
  void* p = nullptr;
  g_peer_last_finegrained = 1;
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    g_peer_last_finegrained = 0;         // coarse-grained: flag polling across DEVICES is not guaranteed coherent -- the caller decides
    if (hipMalloc(&p, (size_t)bytes) != hipSuccess) return DGCNN_ELAUNCH;
  }
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess) { (void)hipFree(p); return DGCNN_ELAUNCH; }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipFree(p); return DGCNN_ELAUNCH; }
  memset(ipc_handle64, 0, 64);
  memcpy(ipc_handle64, &h, sizeof(h));
  *dev_ptr = p;
  return DGCNN_OK;
}

int dgcnn_peer_open(const void* ipc_handle64, void** dev_ptr) {
  if (!ipc_handle64 || !dev_ptr) return DGCNN_EINVAL;
  hipIpcMemHandle_t h;
  memcpy(&h, ipc_handle64, sizeof(h));
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return DGCNN_ELAUNCH; }
  *dev_ptr = p;
  return DGCNN_OK;
}

int dgcnn_peer_close(void* dev_ptr) {
  if (!dev_ptr) return DGCNN_EINVAL;
  return hipIpcCloseMemHandle(dev_ptr) == hipSuccess ? DGCNN_OK : DGCNN_ELAUNCH;
}

int dgcnn_peer_free(void* dev_ptr) {
  if (!dev_ptr) return DGCNN_EINVAL;
  return hipFree(dev_ptr) == hipSuccess ? DGCNN_OK : DGCNN_ELAUNCH;
}

int dgcnn_allreduce_adam_step(int world, int rank, const float* const* peer_grads, unsigned int* const* peer_flags,
                              uint32_t tag, float* params, float* exp_avg, float* exp_avg_sq, float* grad_sum_out, int64_t n,
                              int64_t step, float lr, float beta1, float beta2, float eps, uint32_t* err,
                              dgcnn_stream_t stream) {
  if (world < 1 || world > DG_PEER_MAX || rank < 0 || rank >= world || !peer_grads || !peer_flags || !params || !exp_avg ||
      !exp_avg_sq || !err || n <= 0 || step < 1 || tag == 0)
    return DGCNN_EINVAL;
  DgPeers P;
  memset(&P, 0, sizeof(P));
  for (int r = 0; r < world; ++r) {
    if (!peer_grads[r] || !peer_flags[r]) return DGCNN_EINVAL;
    P.grad[r] = peer_grads[r]; P.flag[r] = peer_flags[r];
  }
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  if ((n & 3) != 0 || (((uintptr_t)params | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)grad_sum_out) & 15) != 0)
    return DGCNN_EINVAL;                                       // (the flat layout is a multiple of 4 floats, 16-byte aligned)
  for (int r = 0; r < world; ++r) if ((uintptr_t)peer_grads[r] & 15) return DGCNN_EINVAL;
  hipLaunchKernelGGL(k_allreduce_adam, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, world, rank,
                     tag, params, exp_avg, exp_avg_sq, grad_sum_out, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2),
                     err, g_peer_spins);
  DG_CHECK_LAUNCH();
  return DGCNN_OK;
}

}  // extern "C"
