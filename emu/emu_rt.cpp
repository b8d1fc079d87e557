// Scheduler of the CPU SIMT emulation (see emu/include/hip/hip_runtime.h).  TEST INFRASTRUCTURE.
//
// A launch runs its workgroups one after the other (blockIdx.x fastest), a workgroup's lanes as fibres on ONE OS thread:
//   * a lane runs until it finishes, reaches a workgroup barrier, or reaches a wave-level operation (it deposits its operand
//     and yields);
//   * when every live lane of a wave is parked, the wave-level operations are resolved: lanes parked at the same call site form
//     one execution of the instruction with exactly those lanes active; if lanes are parked at different sites (a divergent
//     branch with cross-lane work inside), the site with the lowest code address goes first -- lanes that took the branch catch
//     up with the ones waiting behind it;
//   * when every live lane of the workgroup is parked at the barrier, the barrier opens.
// One lane runs at a time: no data race can show here, no memory-ordering bug, no timing.  What shows is arithmetic and indexing.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <ucontext.h>
#include <sys/mman.h>
#include <stdio.h>
#include <vector>

namespace dg_emu {
Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

static const size_t kStack = 512 * 1024;
static const size_t kSmem = 160 * 1024;      // one CU's LDS: a launch asking for more than the hardware has is refused
static char* g_smem = nullptr;               // the launch's dynamic LDS: exactly the bytes asked for, from the heap -- so that an
char* dyn_smem() { return g_smem; }          // -fsanitize=address build of the emulation (make ... EXTRA=-fsanitize=address) sees overruns

enum State { RUNNABLE, AT_WAVEOP, AT_BARRIER, DONE };
struct Lane {
  ucontext_t ctx;
  char* stack = nullptr;
  State st = DONE;
  const void* in = nullptr;
  Site site{nullptr, 0, 0};
  unsigned nbytes = 0;
  unsigned tid = 0;
};
static std::vector<Lane> g_lanes;                 // grown, never shrunk: stacks are reused across launches
static ucontext_t g_sched;
static Lane* g_cur = nullptr;
static std::function<void()>* g_body = nullptr;
static WaveBuf g_buf;                             // exchange buffer of the operation being executed (lanes consume it one by one)
static uint64_t g_clock = 0;
static bool g_trace = false;

uint64_t clock() { return ++g_clock; }
static int g_in_atomic = 0;
void atomic_begin() { ++g_in_atomic; }
void atomic_end() { --g_in_atomic; }

// ---- LDS race detector (builds with -fsanitize-coverage=trace-pc-guard,trace-loads,trace-stores: emu/Makefile `race`) -----------
// Every plain load / store of the kernel sources calls a hook with its address.  For addresses inside the launch's dynamic LDS or
// the section of the static __shared__ arrays, a shadow word remembers the wave that last wrote it and the waves that read it IN THE
// CURRENT BARRIER EPOCH (the epoch advances whenever the workgroup barrier opens).  Two DIFFERENT waves touching the same word in one
// epoch, at least one of them writing, is a data race on the hardware whatever order this emulation happened to run them in --
// independent of timing, which is exactly what a GPU test cannot promise.  Accesses inside atomic operations are exempt; lanes of one
// wave are not checked against each other (lockstep: DG_LOCKSTEP documents those places).
struct Shadow { uint32_t wepoch, wwave, repoch, rmask; };
static bool g_race = false;
static uint32_t g_epoch = 1;
static size_t g_shmem = 0;
static std::vector<Shadow> g_sh_dyn, g_sh_static;
extern "C" char __start_dg_lds[] __attribute__((weak));
extern "C" char __stop_dg_lds[] __attribute__((weak));
static unsigned g_race_reports = 0;
static const void* g_race_pcs[64];
unsigned long long g_race_total = 0;
static void race_report(const char* kind, const char* where, size_t off, unsigned wave_a, unsigned wave_b, const void* pc) {
  ++g_race_total;
  for (unsigned i = 0; i < g_race_reports; ++i) if (g_race_pcs[i] == pc) return;
  if (g_race_reports >= 64) return;
  g_race_pcs[g_race_reports++] = pc;
  Dl_info di;
  const bool have = dladdr(pc, &di) != 0 && di.dli_fbase;
  FILE* out = stderr;
  if (const char* lp = getenv("DG_EMU_RACE_LOG")) { FILE* f = fopen(lp, "a"); if (f) out = f; }      // (xdist workers: stderr is swallowed)
  fprintf(out, "dg_emu RACE %s: %s LDS +%zu, block %u, waves %u and %u in one barrier epoch, at %s+0x%zx (llvm-symbolizer -e <lib> <offset>)\n",
          kind, where, off, g_blockIdx.x, wave_a, wave_b, have ? di.dli_fname : "?",
          have ? (size_t)((const char*)pc - (const char*)di.dli_fbase) : (size_t)pc);
  if (out != stderr) fclose(out);
}

static bool site_eq(const Site& a, const Site& b) { return a.line == b.line && a.col == b.col && (a.file == b.file || !strcmp(a.file, b.file)); }
static bool site_less(const Site& a, const Site& b) {
  const int c = a.file == b.file ? 0 : strcmp(a.file, b.file);
  if (c) return c < 0;
  return a.line != b.line ? a.line < b.line : a.col < b.col;
}

static inline void lds_access(const void* a, unsigned n, bool wr, const void* pc) {
  if (!g_race || !g_cur || g_in_atomic) return;
  const char* p = static_cast<const char*>(a);
  Shadow* sh; size_t off; const char* where;
  if (g_smem && p >= g_smem && p < g_smem + g_shmem) { off = (size_t)(p - g_smem); sh = g_sh_dyn.data(); where = "dynamic"; }
  else if (__start_dg_lds && p >= __start_dg_lds && p < __stop_dg_lds) { off = (size_t)(p - __start_dg_lds); sh = g_sh_static.data(); where = "static"; }
  else return;
  const uint32_t wave = g_cur->tid >> 6;
  for (size_t w = off >> 2; w <= (off + n - 1) >> 2; ++w) {
    Shadow& s = sh[w];
    if (wr) {
      if (s.wepoch == g_epoch && s.wwave != wave) race_report("write/write", where, 4 * w, s.wwave, wave, pc);
      if (s.repoch == g_epoch && (s.rmask & ~(1u << wave))) race_report("read/write", where, 4 * w, (unsigned)__builtin_ctz(s.rmask & ~(1u << wave)), wave, pc);
      s.wepoch = g_epoch; s.wwave = wave;
    } else {
      if (s.wepoch == g_epoch && s.wwave != wave) race_report("write/read", where, 4 * w, s.wwave, wave, pc);
      if (s.repoch != g_epoch) { s.repoch = g_epoch; s.rmask = 0; }
      s.rmask |= 1u << wave;
    }
  }
}
static void lane_entry() {
  (*g_body)();
  g_cur->st = DONE;
  swapcontext(&g_cur->ctx, &g_sched);
}
static void yield_to_sched() { swapcontext(&g_cur->ctx, &g_sched); }
static void run_lane(Lane* l) {
  g_cur = l;
  g_threadIdx = Idx{l->tid, 0, 0};
  swapcontext(&g_sched, &l->ctx);
}

void barrier() {
  g_cur->st = AT_BARRIER;
  yield_to_sched();
}
const WaveBuf& wave_exchange(const void* in, unsigned nbytes, Site site) {
  if (nbytes > 64) { fprintf(stderr, "dg_emu: operand of %u bytes\n", nbytes); abort(); }
  g_cur->in = in; g_cur->nbytes = nbytes; g_cur->site = site; g_cur->st = AT_WAVEOP;
  yield_to_sched();
  g_buf.lane = (int)(g_cur->tid & 63);
  return g_buf;
}
void wave_release() {}

void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
  static bool init = false;
  if (!init) {
    init = true; g_trace = getenv("DG_EMU_TRACE") != nullptr; g_race = getenv("DG_EMU_RACE") != nullptr;
    if (g_race && __start_dg_lds) g_sh_static.assign((size_t)(__stop_dg_lds - __start_dg_lds) / 4 + 2, Shadow{0, 0, 0, 0});
    if (g_race) g_sh_dyn.assign(kSmem / 4 + 2, Shadow{0, 0, 0, 0});
  }
  const unsigned nthr = block.x * block.y * block.z;
  if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1 || nthr == 0 || nthr > 1024 || (nthr & 63) || shmem > kSmem) {
    fprintf(stderr, "dg_emu: unsupported launch shape grid (%u,%u,%u) block (%u,%u,%u) shmem %zu\n", grid.x, grid.y, grid.z, block.x,
            block.y, block.z, shmem);
    abort();
  }
  if (g_lanes.size() < nthr) g_lanes.resize(nthr);
  for (unsigned t = 0; t < nthr; ++t)
    if (!g_lanes[t].stack) {
      void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (p == MAP_FAILED) { perror("dg_emu: mmap"); abort(); }
      g_lanes[t].stack = static_cast<char*>(p);
    }
  void* sm = nullptr;
  if (posix_memalign(&sm, 256, shmem ? shmem : 16)) { perror("dg_emu: posix_memalign"); abort(); }
  memset(sm, 0xA5, shmem ? shmem : 16);       // (LDS is not zero-initialised on the hardware either)
  g_smem = static_cast<char*>(sm);
  g_shmem = shmem;
  g_body = &body;
  g_blockDim = Idx{block.x, 1, 1};
  g_gridDim = Idx{grid.x, 1, 1};
  const unsigned nw = nthr / 64;
  for (unsigned b = 0; b < grid.x; ++b) {
    g_blockIdx = Idx{b, 0, 0};
    ++g_epoch;                                   // (a new workgroup: nothing of the previous one's accesses counts)
    for (unsigned t = 0; t < nthr; ++t) {
      Lane& l = g_lanes[t];
      l.tid = t; l.st = RUNNABLE; l.in = nullptr; l.site = Site{nullptr, 0, 0};
      getcontext(&l.ctx);
      l.ctx.uc_stack.ss_sp = l.stack;
      l.ctx.uc_stack.ss_size = kStack;
      l.ctx.uc_link = nullptr;
      makecontext(&l.ctx, lane_entry, 0);
    }
    for (;;) {
      bool any_live = false;
      for (unsigned w = 0; w < nw; ++w) {
        Lane* wl = &g_lanes[64 * w];
        for (;;) {
          for (int i = 0; i < 64; ++i)
            if (wl[i].st == RUNNABLE) run_lane(&wl[i]);
          // every lane of the wave is parked or done: resolve one wave-level operation, if any
          const Site* site = nullptr;      // the earliest one in the source: lanes inside a divergent branch catch up with the rest
          for (int i = 0; i < 64; ++i)
            if (wl[i].st == AT_WAVEOP && (!site || site_less(wl[i].site, *site))) site = &wl[i].site;
          if (!site) break;
          const Site cur = *site;
          uint64_t mask = 0;
          for (int i = 0; i < 64; ++i) {
            g_buf.in[i] = nullptr;
            if (wl[i].st == AT_WAVEOP && site_eq(wl[i].site, cur)) { mask |= 1ull << i; g_buf.in[i] = wl[i].in; }
          }
          // inactive lanes read as zeros (64 bytes cover every operand type deposited)
          static const char zeros[64] = {0};
          for (int i = 0; i < 64; ++i) if (!g_buf.in[i]) g_buf.in[i] = zeros;
          g_buf.mask = mask;
          if (g_trace) {
            bool split = false;
            for (int i = 0; i < 64; ++i) if (wl[i].st == AT_WAVEOP && !site_eq(wl[i].site, cur)) split = true;
            if (split) fprintf(stderr, "dg_emu: block %u wave %u: lanes parked at different sites, %s:%d:%d first (mask %016llx)\n", b, w,
                               cur.file, cur.line, cur.col, (unsigned long long)mask);
          }
          // the deposits live on the lanes' stacks: copy them before any participating lane runs on (it may run arbitrarily far and
          // reuse the slot).  Each lane, once resumed, takes its result from the copies before it does anything else; the copies
          // are overwritten only by the next resolution, which cannot start before all lanes of this one are parked again.
          static char copies[64][64];
          for (int i = 0; i < 64; ++i)
            if ((mask >> i) & 1) { memcpy(copies[i], wl[i].in, wl[i].nbytes); g_buf.in[i] = copies[i]; wl[i].st = RUNNABLE; }
        }
        for (int i = 0; i < 64; ++i) if (wl[i].st != DONE) any_live = true;
      }
      if (!any_live) break;
      // every live lane is at the barrier
      for (unsigned t = 0; t < nthr; ++t) {
        if (g_lanes[t].st == AT_WAVEOP || g_lanes[t].st == RUNNABLE) { fprintf(stderr, "dg_emu: scheduler invariant broken\n"); abort(); }
        if (g_lanes[t].st == AT_BARRIER) g_lanes[t].st = RUNNABLE;
      }
      ++g_epoch;                                 // the barrier opens: a new epoch of the race detector
    }
  }
  g_body = nullptr;
  g_cur = nullptr;
  free(g_smem);
  g_smem = nullptr;
}
}  // namespace dg_emu

// instrumentation hooks (clang -fsanitize-coverage=trace-pc-guard,trace-loads,trace-stores); this file itself is built without them
extern "C" {
unsigned long long dg_emu_race_count() { return dg_emu::g_race_total; }
void __sanitizer_cov_trace_pc_guard(uint32_t*) {}
void __sanitizer_cov_trace_pc_guard_init(uint32_t*, uint32_t*) {}
#define DG_HOOK(N)                                                                                                          \
  void __sanitizer_cov_load##N(void* a) { dg_emu::lds_access(a, N, false, __builtin_return_address(0)); }                      \
  void __sanitizer_cov_store##N(void* a) { dg_emu::lds_access(a, N, true, __builtin_return_address(0)); }
DG_HOOK(1) DG_HOOK(2) DG_HOOK(4) DG_HOOK(8) DG_HOOK(16)
}
