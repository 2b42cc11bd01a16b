// Ground-truth probe (GPU box): what does each ingredient of the GEMM main loop cost on this chip?
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe ; run: /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int MODE>
__global__ void __launch_bounds__(256, 2) probe(const unsigned short* g, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // fill LDS
  for (int i = threadIdx.x; i < 72 * 1024 / 16; i += 256) ((uint4*)smem)[i] = ((const uint4*)g)[i];
  __syncthreads();
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 xf[8], wf[4];
  const char* base = smem + (lane & 15) * 64 + ((lane >> 4) << 4) + wave * 8192;
  for (int i = 0; i < 8; ++i) xf[i] = *(const bf16x8*)(base + i * 1024);
  for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(base + 16384 + i * 256);
  const unsigned short* gp = g + (size_t)blockIdx.x * 4096 + wave * 1024 + lane * 8;
  for (int it = 0; it < iters; ++it) {
    const char* sb = base + (it % 3) * 24576;
    if (MODE >= 3) {
      if (MODE >= 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (MODE >= 4) {
      char* dst = smem + ((it + 2) % 3) * 24576 + wave * 6 * 1024;
      for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR(gp + ((it * 6 + i) & 63) * 64), LDS_PTR(dst + i * 1024), 16, 0, 0);
    }
    if (MODE >= 2) {
      for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(sb + 16384 + i * 256);
      for (int i = 0; i < 8; ++i) xf[i] = *(const bf16x8*)(sb + i * 1024);
    }
    for (int mi = 0; mi < 8; ++mi)
      for (int ni = 0; ni < 4; ++ni)
        acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[mi * 4 + ni], 0, 0, 0);
    if (MODE == 1) { for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(xf[i])); }
  }
  if (MODE >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const unsigned short* g, float* out, int blocks, const char* name) {
  const int iters = 2000, lds = 72 * 1024;
  hipFuncSetAttribute((const void*)&probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), lds, 0, g, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double fl = (double)blocks * 4 * iters * 32 * 16384.0;
  printf("%-44s blocks=%d  %.3f ms  %.0f TFLOP/s\n", name, blocks, best, fl / (best * 1e-3) / 1e12);
}

int main() {
  unsigned short* g; float* out;
  hipMalloc(&g, 64 << 20); hipMalloc(&out, 4 << 20);
  std::vector<unsigned short> h((64 << 20) / 2);
  unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = 0x3c00 + ((x >> 16) & 0x3ff) + ((x >> 31) << 15); }
  hipMemcpy(g, h.data(), 64 << 20, hipMemcpyHostToDevice);
  for (int blocks : {512, 256}) {
    run<1>(g, out, blocks, "1 mfma only");
    run<2>(g, out, blocks, "2 + 12 ds_read_b128 per 32 mfma");
    run<3>(g, out, blocks, "3 + s_barrier per step");
    run<4>(g, out, blocks, "4 + 6 LDS-DMA pieces per step (L2 resident)");
  }
  return 0;
}
