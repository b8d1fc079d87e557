// Calibration probe (GPU box): steady-state cost of back-to-back dependent kernels in one stream, as a function of
// the dependent-load chain inside each workgroup.  hipcc --offload-arch=gfx950 floor_probe.hip -o /tmp/floor && /tmp/floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) k_empty(const int* a, const int* b, float* out, int n) {}
__global__ void __launch_bounds__(1024) k_hop1(const int* a, const int* b, float* out, int n) {
  const int t = blockIdx.x * 1024 + threadIdx.x;
  out[t] = (float)a[t];
}
__global__ void __launch_bounds__(1024) k_hop2(const int* a, const int* b, float* out, int n) {
  const int t = blockIdx.x * 1024 + threadIdx.x;
  out[t] = (float)b[a[t]];
}
__global__ void __launch_bounds__(1024) k_hop3(const int* a, const int* b, float* out, int n) {
  const int t = blockIdx.x * 1024 + threadIdx.x;
  out[t] = (float)b[b[a[t]]];
}
__global__ void __launch_bounds__(1024) k_hop3b(const int* a, const int* b, float* out, int n) {   // + 2 barriers
  __shared__ float s[1024];
  const int t = blockIdx.x * 1024 + threadIdx.x;
  s[threadIdx.x] = (float)b[b[a[t]]];
  __syncthreads();
  float v = s[(threadIdx.x + 64) & 1023];
  __syncthreads();
  out[t] = v;
}
template <typename K> static float run(K k, int grid, const int* a, const int* b, float* out, int n, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, 0, a, b, out, n);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, 0, a, b, out, n);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}
int main() {
  for (int grid : {50, 232, 2048}) {
    const int n = grid * 1024;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919 + 13) % n);
    int *a, *b; float* out;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&out, n * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice);
    printf("grid %4d x 1024: empty %.2f  1 hop %.2f  2 hops %.2f  3 hops %.2f  3 hops + 2 barriers %.2f  us/kernel\n", grid,
           run(k_empty, grid, a, b, out, n, 2000), run(k_hop1, grid, a, b, out, n, 2000), run(k_hop2, grid, a, b, out, n, 2000),
           run(k_hop3, grid, a, b, out, n, 2000), run(k_hop3b, grid, a, b, out, n, 2000));
    hipFree(a); hipFree(b); hipFree(out);
  }
  return 0;
}
