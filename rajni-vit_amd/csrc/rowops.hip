// HBM-bound row kernels: LayerNorm (SURVEY k1,k17,k20) and the keep_idx row gather (k10,k15).
// One wave per row, 16-byte accesses, everything else in registers.
#include "common.h"

namespace {

constexpr int LN_MAX_CHUNKS = 4;  // 16-byte chunks per lane: C <= 64*8*4 = 2048

template <bool XF32>
__global__ void __launch_bounds__(256) layernorm_bf16(const void* __restrict__ xv, long xs,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ b,
                                                      bf16_t* __restrict__ y, int rows, int C,
                                                      float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int nchunk = C >> 3;
  float v[LN_MAX_CHUNKS][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      if (XF32) {
        const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xv) + (long)row * xs + c * 8);
        const float4 a = q[0], d = q[1];
        v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w;
        v[i][4] = d.x; v[i][5] = d.y; v[i][6] = d.z; v[i][7] = d.w;
      } else {
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(xv) + (long)row * xs + c * 8), v[i]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[i][j] - mean;
        ss += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
  bf16_t* yr = y + (long)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      const float4 w0 = *reinterpret_cast<const float4*>(w + c * 8);
      const float4 w1 = *reinterpret_cast<const float4*>(w + c * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(b + c * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(b + c * 8 + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf((v[i][j] - mean) * rstd, wv[j], bv[j]);
      *reinterpret_cast<uint4*>(yr + c * 8) = pack8(o);
    }
  }
}

// dst[b, j, :] = src[b, idx[b, j], :]   rows of `row_chunks` 16-byte chunks; one wave per row
__global__ void __launch_bounds__(256) gather_rows_kernel(const uint4* __restrict__ src,
                                                         const int* __restrict__ idx,
                                                         uint4* __restrict__ dst, int B, int n_src,
                                                         int n_dst, int row_chunks) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long total = (long)B * n_dst;
  for (long r = (long)blockIdx.x * 4 + wave; r < total; r += (long)gridDim.x * 4) {
    const int b = (int)(r / n_dst);
    const int s = idx[r];
    const uint4* sp = src + ((long)b * n_src + s) * row_chunks;
    uint4* dp = dst + r * row_chunks;
    for (int c = lane; c < row_chunks; c += 64) dp[c] = sp[c];
  }
}

}  // namespace

int launch_layernorm(const void* x, long xs, const float* w, const float* b, void* y, int rows,
                     int C, float eps, int x_f32, hipStream_t s) {
  RAJNI_REQUIRE(x && w && b && y, RAJNI_ERR_INVALID, "rajni_layernorm: null pointer");
  RAJNI_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAX_CHUNKS && xs % 8 == 0,
                RAJNI_ERR_UNSUPPORTED, "rajni_layernorm: need C %% 8 == 0, C <= 2048, stride %% 8 == 0 (C=%d)", C);
  ProfScope prof(KC_LAYERNORM, s, 8.0 * rows * C, (x_f32 ? 6.0 : 4.0) * rows * C);
  if (x_f32)
    hipLaunchKernelGGL(layernorm_bf16<true>, dim3((rows + 3) / 4), dim3(256), 0, s, x, xs, w, b,
                       (bf16_t*)y, rows, C, eps);
  else
    hipLaunchKernelGGL(layernorm_bf16<false>, dim3((rows + 3) / 4), dim3(256), 0, s, x, xs, w, b,
                       (bf16_t*)y, rows, C, eps);
  RAJNI_CHECK_LAUNCH("layernorm_bf16");
  return RAJNI_OK;
}

int launch_gather_rows(const void* src, const int32_t* idx, void* dst, int B, int n_src, int n_dst,
                       int row_bytes, hipStream_t s) {
  RAJNI_REQUIRE(src && idx && dst, RAJNI_ERR_INVALID, "rajni_gather_rows: null pointer");
  RAJNI_REQUIRE(B > 0 && n_src > 0 && n_dst > 0 && row_bytes > 0 && row_bytes % 16 == 0,
                RAJNI_ERR_INVALID, "rajni_gather_rows: row bytes must be a multiple of 16 (%d)", row_bytes);
  const long rows = (long)B * n_dst;
  long blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  ProfScope prof(KC_GATHER, s, 0.0, 2.0 * rows * row_bytes);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((int)blocks), dim3(256), 0, s, (const uint4*)src, idx,
                     (uint4*)dst, B, n_src, n_dst, row_bytes / 16);
  RAJNI_CHECK_LAUNCH("gather_rows_kernel");
  return RAJNI_OK;
}
