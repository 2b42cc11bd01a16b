// Ground-truth probe (GPU box) for the fp8 path: v_mfma_f32_16x16x128_f8f6f4 (e4m3 x e4m3).
//   1. semantics: operand slots, C/D layout and what the scale arguments mean, checked with exact integer data
//      against a host dot product - with the k-slot assignment the GEMM uses (lane group g holds bytes
//      [32g, 32g+32) of the 128-byte K step for BOTH operands; any consistent assignment gives the same sum);
//   2. rate: MFMA-only loop, fp8 16x16x128 against bf16 16x16x32, every CU busy.
// Build: hipcc --offload-arch=gfx950 -O3 tools/f8_mfma_probe.hip -o gpurun_out/f8_probe ; run it on the box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// decode OCP e4m3fn
static float e4m3(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) return NAN;
  if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}

template <int SCALE_MODE>
__global__ void sem_kernel(const unsigned char* A, const unsigned char* B, float* D) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  v8i a, b;
  const int* ap = reinterpret_cast<const int*>(A + r * 128 + g * 32);
  const int* bp = reinterpret_cast<const int*>(B + r * 128 + g * 32);
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (SCALE_MODE == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  else if (SCALE_MODE == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x80808080, 0, 0x7F7F7F7F);   // A x 2
  // C/D: col = lane & 15 (B's row index), row = 4 * (lane >> 4) + reg (A's row index)
#pragma unroll
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}

template <int F8>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  v8i a8, b8;
  bf16x8 a16, b16;
  for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 + threadIdx.x; b8[i] = 0x3C383430 + i; a16[i] = (__bf16)(0.5f + i); b16[i] = (__bf16)(1.0f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (F8) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16, b16, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  std::vector<unsigned char> A(16 * 128), B(16 * 128);
  srand(1);
  // small exactly representable values: codes for 0, +-0.5, +-1, +-1.5, +-2, +-3
  const unsigned char codes[] = {0x00, 0x30, 0xB0, 0x38, 0xB8, 0x3C, 0xBC, 0x40, 0xC0, 0x44, 0xC4};
  for (auto& v : A) v = codes[rand() % 11];
  for (auto& v : B) v = codes[rand() % 11];
  unsigned char *dA, *dB;
  float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 0) hipLaunchKernelGGL(sem_kernel<0>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    else if (mode == 1) hipLaunchKernelGGL(sem_kernel<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    else hipLaunchKernelGGL(sem_kernel<2>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double maxerr = 0, ratio = 0;
    int nz = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += (double)e4m3(A[i * 128 + k]) * e4m3(B[j * 128 + k]);
        maxerr = fmax(maxerr, fabs(D[i * 16 + j] - ref));
        if (ref != 0) { ratio += D[i * 16 + j] / ref; ++nz; }
      }
    printf("scale mode %d (%s): max |D - ref| = %g, mean D/ref = %g\n", mode,
           mode == 0 ? "scale args 0,0" : mode == 1 ? "E8M0 127,127" : "E8M0 128 (A), 127 (B)", maxerr, ratio / nz);
  }
  float* dout;
  hipMalloc(&dout, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int f8 = 0; f8 < 2; ++f8) {
    const int iters = 20000, blocks = 256 * 4;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (f8) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, dout, iters);
      else hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, dout, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * (f8 ? 128 : 32);
    printf("%s MFMA-only: %.1f TFLOP/s\n", f8 ? "fp8 16x16x128" : "bf16 16x16x32", flop / (ms * 1e-3) / 1e12);
  }
  return 0;
}
