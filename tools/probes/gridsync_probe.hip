// Calibration probe (GPU box): cost of an in-kernel grid barrier (cooperative launch, hand-rolled atomic barrier with
// device-scope fences) vs a kernel boundary, for the grid shape of the B=50 aggregation kernels (232 x 1024).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void __launch_bounds__(1024) k_cg(float* buf, int n, int rounds) {
  cg::grid_group grid = cg::this_grid();
  const int t = blockIdx.x * 1024 + threadIdx.x;
  float v = (float)t;
  for (int r = 0; r < rounds; ++r) {
    buf[(r & 1) * n + t] = v;
    grid.sync();
    v = buf[(r & 1) * n + (t + 4099) % n] + 1.f;      // another workgroup's (another XCD's) value
  }
  buf[2 * n + t] = v;
}
// hand-rolled: one atomic arrive per workgroup, spin on a generation counter
__device__ __forceinline__ void grid_barrier(unsigned int* ctr, unsigned int* gen, unsigned int nblk, unsigned int& mygen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned int g = mygen + 1;
    if (atomicAdd(ctr, 1u) == nblk - 1) { atomicExch(ctr, 0u); __threadfence(); atomicExch(gen, g); }
    else { while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != g) __builtin_amdgcn_s_sleep(1); }
    __threadfence();
  }
  mygen += 1;
  __syncthreads();
}
__global__ void __launch_bounds__(1024) k_hand(float* buf, int n, int rounds, unsigned int* ctr, unsigned int* gen, unsigned int gen0) {
  const int t = blockIdx.x * 1024 + threadIdx.x;
  float v = (float)t;
  unsigned int mygen = gen0;
  for (int r = 0; r < rounds; ++r) {
    __builtin_nontemporal_store(v, &buf[(r & 1) * n + t]);
    grid_barrier(ctr, gen, gridDim.x, mygen);
    v = __builtin_nontemporal_load(&buf[(r & 1) * n + (t + 4099) % n]) + 1.f;
  }
  buf[2 * n + t] = v;
}
int main() {
  const int grid = 232, n = grid * 1024;
  float* buf; unsigned int* sync;
  hipMalloc(&buf, 3 * n * 4); hipMalloc(&sync, 8); hipMemset(sync, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rounds : {1, 9, 33}) {
    int nn = n, rr = rounds;
    void* args[] = {&buf, &nn, &rr};
    for (int i = 0; i < 3; ++i) hipLaunchCooperativeKernel((void*)k_cg, dim3(grid), dim3(1024), args, 0, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchCooperativeKernel((void*)k_cg, dim3(grid), dim3(1024), args, 0, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("cooperative grid.sync : rounds %2d -> %.2f us/kernel\n", rounds, ms * 1e3f / 200);
  }
  unsigned int gen0 = 0;
  for (int rounds : {1, 9, 33}) {
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(k_hand, dim3(grid), dim3(1024), 0, 0, buf, n, rounds, sync, sync + 1, gen0); gen0 += rounds; }
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(k_hand, dim3(grid), dim3(1024), 0, 0, buf, n, rounds, sync, sync + 1, gen0); gen0 += rounds; }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hand-rolled barrier   : rounds %2d -> %.2f us/kernel\n", rounds, ms * 1e3f / 200);
  }
  hipError_t err = hipGetLastError();
  printf("last error: %s\n", hipGetErrorString(err));
  return 0;
}
