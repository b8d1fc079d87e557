// tr16_probe.hip -- hardware facts the chain kernels (gcn_chain.hip) rely on, checked on a real MI355X:
//   (1) ds_read_b64_tr_b16: lane i of a 16-lane group supplies the address of 4 contiguous b16; lane c of the group
//       receives element (c & 3) of the pieces addressed by lanes 4j + (c >> 2), j = 0..3
//   (2) v_mfma_f32_16x16x32_bf16 operand / result layout with the roles used by the transposed block product
//   (3) v_mfma_f32_16x16x4_f32 layout with swapped operands (W as A, x as B)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_tr(const unsigned short* in, const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int t = threadIdx.x; t < 4096; t += 64) lds[t] = in[t];
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

__global__ void k_mfma_bf16(const float* A, const float* B, float* D) {      // D[16x16] = A[16x32] . B[32x16]
  const int l = threadIdx.x, i = l & 15, kg = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[i * 32 + 8 * kg + j]; b[j] = (__bf16)B[(8 * kg + j) * 16 + i]; }
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * kg + r) * 16 + i] = d[r];
}

__global__ void k_mfma_f32(const float* A, const float* B, float* D) {       // D[16x16] = A[16x4] . B[4x16]
  const int l = threadIdx.x, i = l & 15, kq = l >> 4;
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + kq], B[kq * 16 + i], d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + i] = d[r];
}

int main() {
  int bad = 0;
  {   // (1)
    unsigned short h[4096]; int ad[64]; unsigned short o[256];
    for (int t = 0; t < 4096; ++t) h[t] = (unsigned short)t;
    srand(1);
    for (int l = 0; l < 64; ++l) ad[l] = 4 * (rand() % 1000);           // arbitrary 8-B aligned pieces
    unsigned short *din, *dout; int* dad;
    hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o); hipMalloc(&dad, sizeof ad);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice); hipMemcpy(dad, ad, sizeof ad, hipMemcpyHostToDevice);
    k_tr<<<1, 64>>>(din, dad, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int g = l & ~15, c = l & 15;
        const int want = ad[g + 4 * j + (c >> 2)] + (c & 3);
        if (o[l * 4 + j] != want) { if (bad < 8) printf("tr: lane %d elem %d got %d want %d\n", l, j, o[l * 4 + j], want); ++bad; }
      }
    printf("tr16 semantics: %s\n", bad ? "MISMATCH" : "ok");
  }
  {   // (2)
    float A[512], B[512], D[256], *dA, *dB, *dD; int b2 = 0;
    for (int t = 0; t < 512; ++t) { A[t] = (float)((t * 7) % 5 - 2); B[t] = (float)((t * 3) % 7 - 3); }
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dD, sizeof D);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    k_mfma_bf16<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j];
      if (s != D[i * 16 + j]) { if (b2 < 4) printf("mfma bf16: D[%d][%d] %g want %g\n", i, j, D[i * 16 + j], s); ++b2; }
    }
    printf("mfma 16x16x32 bf16 layout: %s\n", b2 ? "MISMATCH" : "ok"); bad += b2;
  }
  {   // (3)
    float A[64], B[64], D[256], *dA, *dB, *dD; int b3 = 0;
    for (int t = 0; t < 64; ++t) { A[t] = (float)((t * 5) % 9 - 4); B[t] = (float)((t * 11) % 7 - 3); }
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dD, sizeof D);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    k_mfma_f32<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      float s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j];
      if (s != D[i * 16 + j]) { if (b3 < 4) printf("mfma f32: D[%d][%d] %g want %g\n", i, j, D[i * 16 + j], s); ++b3; }
    }
    printf("mfma 16x16x4 f32 layout: %s\n", b3 ? "MISMATCH" : "ok"); bad += b3;
  }
  return bad ? 1 : 0;
}
