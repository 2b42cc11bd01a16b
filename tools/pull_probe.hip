// Ground-truth probe (GPU box): how fast does ONE CU pull HBM-miss data, by access shape and by how many CUs pull at once?
// The fp32-stream residual epilogue reads a 256-row x 128-column fp32 tile of a [M, 768] tensor per workgroup (rows 3072 B apart):
//   shape A  the epilogue's: a wave instruction = 16 rows x 64 B (4 lanes x 16 B per row), 4 instructions per 16-row group
//   shape B  full 128-byte lines: a wave instruction = 8 rows x 128 B
//   shape C  a wave instruction = 4 rows x 256 B (the wave's whole 64-column span of a row)
//   shape D  1 KiB contiguous per wave instruction (a flat buffer: the best case)
// Build: hipcc --offload-arch=gfx950 -O3 tools/pull_probe.hip -o /tmp/pull_probe ; run: /tmp/pull_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

// MODE 0: loads only; 1: stores only (to y); 2: the epilogue's sequence - 16 loads of x, wait, 16 stores to y - per tile
template <int SHAPE, int MODE>
__global__ void __launch_bounds__(512) pull(const float* __restrict__ x, float* __restrict__ y, float* out, unsigned long long* cyc, int ld, int tiles_per_wg, int tile_stride_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;          // 4 x 2 waves: 64 rows x 64 columns each (the 256 x 128 tiling)
  float4 acc = make_float4(0, 0, 0, 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles_per_wg; ++t) {
    const long row0 = ((long)blockIdx.x * tiles_per_wg + t) * tile_stride_rows + wm * 64;
    float4 v[16];
    long off[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      long r, c;
      if (SHAPE == 0) { r = row0 + (i >> 2) * 16 + (lane & 15); c = wn * 64 + (i & 3) * 16 + (lane >> 4) * 4; }
      else if (SHAPE == 1) { r = row0 + (i >> 1) * 8 + (lane >> 3); c = wn * 64 + (i & 1) * 32 + (lane & 7) * 4; }
      else if (SHAPE == 2) { r = row0 + i * 4 + (lane >> 4); c = wn * 64 + (lane & 15) * 4; }
      else { r = 0; c = (((long)blockIdx.x * tiles_per_wg + t) * 8 + wave) * 4096 + i * 256 + lane * 4; }   // flat
      off[i] = r * ld + c;
      if (MODE != 1) v[i] = *reinterpret_cast<const float4*>(x + off[i]);
      else v[i] = make_float4((float)i, 1.f, 2.f, (float)lane);
    }
    if (MODE != 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { v[i].x += 1.f; *reinterpret_cast<float4*>(y + off[i]) = v[i]; }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
  }
  if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the CU
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int ld = 768, rows = 1 << 19;                       // 1.5 GiB fp32: far beyond the 256 MiB Infinity Cache
  float* x; float* y; float* out; unsigned long long* cyc;
  hipMalloc(&x, (size_t)rows * ld * 4); hipMalloc(&y, (size_t)rows * ld * 4); hipMalloc(&out, 256 * 512 * 4 * 4); hipMalloc(&cyc, 1024 * 8);
  hipMemset(x, 0, (size_t)rows * ld * 4); hipMemset(y, 0, (size_t)rows * ld * 4);
  const char* names[4] = {"A 16 rows x 64 B ", "B  8 rows x 128 B", "C  4 rows x 256 B", "D 1 KiB flat     "};
  const char* modes[3] = {"loads only", "stores only", "16 loads, wait, 16 stores"};
  for (int mode = 0; mode < 3; ++mode)
  for (int nwg : {16, 256}) {
    for (int shape = 0; shape < 4; ++shape) {
      const int tiles = 4;                                  // 4 tiles of 128 KiB per workgroup, one after the other
      std::vector<unsigned long long> h(nwg);
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {
        const int stride = 256 + 7 * rep;                   // different rows every repetition: never cache resident
        const size_t o = (size_t)rep * 97 * ld * 1024 % ((size_t)rows * ld / 2);
#define L(S, M) if (shape == S && mode == M) hipLaunchKernelGGL((pull<S, M>), dim3(nwg), dim3(512), 0, 0, x + o, y + o, out, cyc, ld, tiles, stride);
        L(0, 0) L(1, 0) L(2, 0) L(3, 0) L(0, 1) L(1, 1) L(2, 1) L(3, 1) L(0, 2) L(1, 2) L(2, 2) L(3, 2)
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        best = std::min(best, (double)h[nwg / 2]);
      }
      const double bytes = tiles * 131072.0 * (mode == 2 ? 2 : 1);
      printf("%-26s %3d workgroups  shape %s  median %7.0f cycles per workgroup = %5.1f B/clk/CU\n", modes[mode], nwg, names[shape], best, bytes / best);
    }
  }
  return 0;
}
