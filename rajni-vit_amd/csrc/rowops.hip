// HBM-bound row kernels: LayerNorm (SURVEY k1,k17,k20) and the keep_idx row gather (k10,k15).
// One wave per row, 16-byte accesses, everything else in registers.
#include "common.h"

namespace {

#ifndef RAJNI_LN8_ROWS
#define RAJNI_LN8_ROWS 2  // ... of the e4m3-output LayerNorm
#endif
#ifndef RAJNI_LN_ROWS
#define RAJNI_LN_ROWS 2   // rows per wave of the benchmark path's LayerNorm (1 = the one-row kernel; 3: 728, 4: 750 us per forward against 715)
#endif
constexpr int LN_MAX_CHUNKS = 4;  // 16-byte chunks per lane: C <= 64*8*4 = 2048

template <typename TX, typename TY>
__global__ void __launch_bounds__(256) layernorm_kernel(const TX* __restrict__ x, long xs,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        TY* __restrict__ y, int rows, int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int nchunk = C >> 3;
  const TX* xr = x + (long)row * xs;
  float v[LN_MAX_CHUNKS][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      load8<TX>(xr + c * 8, v[i]);
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
  TY* yr = y + (long)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      float wv[8], bv[8], o[8];
      load8<float>(w + c * 8, wv);
      load8<float>(b + c * 8, bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf((v[i][j] - mean) * rstd, wv[j], bv[j]);
      store8<TY>(yr + c * 8, o);
    }
  }
}

// R rows per wave (fp32 rows in, bf16 out, C <= 1024): R times the loads in flight per wave.  The one-row kernel keeps 3 KB per wave in
// flight at 8 waves per SIMD - Little's law put that short of the HBM rate: LayerNorm was a latency-bound kernel, not a bandwidth-bound one
// (two rows, still 64 VGPRs = 8 waves: 772 -> 715 us per forward; the e4m3-output kernel, with four wave reductions per row, loses with two).
template <int NC, int R>
__global__ void __launch_bounds__(256) layernorm_rows_kernel(const float* __restrict__ x, long xs, const float* __restrict__ w,
                                                             const float* __restrict__ b, bf16_t* __restrict__ y, int rows, int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + wave) * R;
  if (r0 >= rows) return;
  const int nchunk = C >> 3;
  float v[R][NC][8];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float* xr = x + (long)(r0 + r < rows ? r0 + r : rows - 1) * xs;      // clamped: a short last group re-reads its last row
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) load8<float>(xr + c * 8, v[r][i]);
    }
  }
  float mean[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[r][i][j];
      }
    }
    mean[r] = wave_sum(sum) / (float)C;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = v[r][i][j] - mean[r];
          ss += d * d;
        }
      }
    }
    rstd[r] = rsqrtf(wave_sum(ss) / (float)C + eps);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      float wv[8], bv[8];
      load8<float>(w + c * 8, wv);
      load8<float>(b + c * 8, bv);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf((v[r][i][j] - mean[r]) * rstd[r], wv[j], bv[j]);
        if (r0 + r < rows) store8<bf16_t>(y + (long)(r0 + r) * C + c * 8, o);
      }
    }
  }
}

// LayerNorm whose output row is quantised for the fp8 matrix pipe (see rajni_layernorm_fp8 in the header): the
// wave holds the whole normalised row in registers, so the row maximum is one more wave reduction.
//   yscale[r] = max |o| / 448 (1 if 0);  yq = e4m3_rne_sat(o * (1 / yscale[r]))   - the reciprocal is a correctly
//   rounded fp32 division and the product one fp32 multiply, so a host restatement reproduces the bytes;
//   hscale[r] (optional) = (1.0625 * ||o||_2 * wnorm + bmax) / 448: per-row scale of the MLP hidden activations
//   from the Cauchy-Schwarz bound (the margin covers the <= 2^-4 relative change of ||o|| under quantisation).
// NC: 16-byte chunks per lane.  2 serves C <= 1024 - every ViT-B / ViT-L row - in 50 VGPRs instead of 70: 8 waves per SIMD instead of
// 7, and no dead third / fourth chunk iterations: 31 -> 28 us per launch on ViT-B at batch 256 (the bf16-output kernel gains nothing
// from the same change: it sits on the HBM rate; this one was short of waves to cover its four wave reductions per row)
template <typename TX, int NC>
__global__ void __launch_bounds__(256) layernorm_fp8_kernel(const TX* __restrict__ x, long xs,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            unsigned char* __restrict__ yq, float* __restrict__ yscale,
                                                            float* __restrict__ hscale, float wnorm, float bmax,
                                                            int rows, int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int nchunk = C >> 3;
  const TX* xr = x + (long)row * xs;
  float v[NC][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      load8<TX>(xr + c * 8, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
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
  float amax = 0.f, osq = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      float wv[8], bv[8];
      load8<float>(w + c * 8, wv);
      load8<float>(b + c * 8, bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float o = fmaf((v[i][j] - mean) * rstd, wv[j], bv[j]);
        v[i][j] = o;
        amax = fmaxf(amax, fabsf(o));
        osq = fmaf(o, o, osq);
      }
    }
  }
  amax = wave_max(amax);
  osq = wave_sum(osq);
  const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
  const float inv = 1.0f / scale;
  unsigned char* yr = yq + (long)row * C;
  int plo[NC], phi[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    plo[i] = phi[i] = 0;
    const int c = lane + i * 64;
    if (c < nchunk) {
      float q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = __builtin_amdgcn_fmed3f(v[i][j] * inv, -448.f, 448.f);
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], lo, true);
      int hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], 0, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], hi, true);
      plo[i] = lo; phi[i] = hi;
    }
  }
  // 16-byte stores: an even lane takes its odd neighbour's 8 bytes (adjacent chunks) - with 8-byte stores a wave
  // instruction wrote half sectors
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + i * 64;
    const int nlo = __shfl_down(plo[i], 1, 64), nhi = __shfl_down(phi[i], 1, 64);
    if (c < nchunk && (lane & 1) == 0) {
      if (c + 1 < nchunk) *reinterpret_cast<uint4*>(yr + c * 8) = make_uint4((unsigned)plo[i], (unsigned)phi[i], (unsigned)nlo, (unsigned)nhi);
      else *reinterpret_cast<uint2*>(yr + c * 8) = make_uint2((unsigned)plo[i], (unsigned)phi[i]);
    }
  }
  if (lane == 0) {
    yscale[row] = scale;
    if (hscale != nullptr) {
      const float bound = fmaf(1.0625f * sqrtf(osq), wnorm, bmax);
      hscale[row] = bound > 0.f ? bound / 448.0f : 1.0f;
    }
  }
}

// The e4m3-output LayerNorm with R rows per wave, stage by stage ACROSS the rows (their four wave reductions each overlap instead of
// queueing up): fp32 rows, C <= 1024.  Same arithmetic per row as layernorm_fp8_kernel: the same bytes.
template <int NC, int R>
__global__ void __launch_bounds__(256) layernorm_fp8_rows_kernel(const float* __restrict__ x, long xs, const float* __restrict__ w,
                                                                 const float* __restrict__ b, unsigned char* __restrict__ yq,
                                                                 float* __restrict__ yscale, float* __restrict__ hscale, float wnorm,
                                                                 float bmax, int rows, int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + wave) * R;
  if (r0 >= rows) return;
  const int nchunk = C >> 3;
  float v[R][NC][8];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float* xr = x + (long)(r0 + r < rows ? r0 + r : rows - 1) * xs;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) load8<float>(xr + c * 8, v[r][i]);
    }
  }
  float mean[R], rstd[R], amax[R], osq[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[r][i][j];
      }
    }
    mean[r] = sum;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) mean[r] = wave_sum(mean[r]) / (float)C;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = v[r][i][j] - mean[r];
          ss += d * d;
        }
      }
    }
    rstd[r] = ss;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(wave_sum(rstd[r]) / (float)C + eps);
#pragma unroll
  for (int r = 0; r < R; ++r) { amax[r] = 0.f; osq[r] = 0.f; }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      float wv[8], bv[8];
      load8<float>(w + c * 8, wv);
      load8<float>(b + c * 8, bv);
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float o = fmaf((v[r][i][j] - mean[r]) * rstd[r], wv[j], bv[j]);
          v[r][i][j] = o;
          amax[r] = fmaxf(amax[r], fabsf(o));
          osq[r] = fmaf(o, o, osq[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) { amax[r] = wave_max(amax[r]); osq[r] = wave_sum(osq[r]); }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = r0 + r;
    const bool live = row < rows;
    const float scale = amax[r] > 0.f ? amax[r] / 448.0f : 1.0f;
    const float inv = 1.0f / scale;
    unsigned char* yr = yq + (long)row * C;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      int lo = 0, hi = 0;
      const int c = lane + i * 64;
      if (c < nchunk) {
        float q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = __builtin_amdgcn_fmed3f(v[r][i][j] * inv, -448.f, 448.f);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], hi, true);
      }
      const int nlo = __shfl_down(lo, 1, 64), nhi = __shfl_down(hi, 1, 64);
      if (live && c < nchunk && (lane & 1) == 0) {
        if (c + 1 < nchunk) *reinterpret_cast<uint4*>(yr + c * 8) = make_uint4((unsigned)lo, (unsigned)hi, (unsigned)nlo, (unsigned)nhi);
        else *reinterpret_cast<uint2*>(yr + c * 8) = make_uint2((unsigned)lo, (unsigned)hi);
      }
    }
    if (live && lane == 0) {
      yscale[row] = scale;
      if (hscale != nullptr) {
        const float bound = fmaf(1.0625f * sqrtf(osq[r]), wnorm, bmax);
        hscale[row] = bound > 0.f ? bound / 448.0f : 1.0f;
      }
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
                     int C, float eps, int x_f32, int dtype, hipStream_t s) {
  RAJNI_REQUIRE(x && w && b && y, RAJNI_ERR_INVALID, "rajni_layernorm: null pointer");
  RAJNI_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAX_CHUNKS && xs % 8 == 0,
                RAJNI_ERR_UNSUPPORTED, "rajni_layernorm: need C %% 8 == 0, C <= 2048, stride %% 8 == 0 (C=%d)", C);
  const bool f32io = dtype == RAJNI_F32;
  ProfScope prof(KC_LAYERNORM, s, 8.0 * rows * C, (f32io ? 8.0 : (x_f32 ? 6.0 : 4.0)) * rows * C);
  const dim3 grid((rows + 3) / 4), block(256);
  if (f32io)
    hipLaunchKernelGGL((layernorm_kernel<float, float>), grid, block, 0, s, (const float*)x, xs, w, b, (float*)y, rows, C, eps);
  else if (x_f32 && RAJNI_LN_ROWS > 1 && C <= 1024 && rows >= 4096)
    hipLaunchKernelGGL((layernorm_rows_kernel<2, RAJNI_LN_ROWS>), dim3((rows + 4 * RAJNI_LN_ROWS - 1) / (4 * RAJNI_LN_ROWS)), block, 0, s,
                       (const float*)x, xs, w, b, (bf16_t*)y, rows, C, eps);
  else if (x_f32)
    hipLaunchKernelGGL((layernorm_kernel<float, bf16_t>), grid, block, 0, s, (const float*)x, xs, w, b, (bf16_t*)y, rows, C, eps);
  else
    hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, xs, w, b, (bf16_t*)y, rows, C, eps);
  RAJNI_CHECK_LAUNCH("layernorm_kernel");
  return RAJNI_OK;
}

int launch_layernorm_fp8(const void* x, long xs, const float* w, const float* b, void* yq, float* yscale,
                         float* hscale, float wnorm, float bmax, int rows, int C, float eps, int x_f32, hipStream_t s) {
  RAJNI_REQUIRE(x && w && b && yq && yscale, RAJNI_ERR_INVALID, "rajni_layernorm_fp8: null pointer");
  RAJNI_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAX_CHUNKS && xs % 8 == 0,
                RAJNI_ERR_UNSUPPORTED, "rajni_layernorm_fp8: need C %% 8 == 0, C <= 2048, stride %% 8 == 0 (C=%d)", C);
  RAJNI_REQUIRE(wnorm >= 0.f && bmax >= 0.f, RAJNI_ERR_INVALID, "rajni_layernorm_fp8: the hidden bound's constants must be >= 0");
  ProfScope prof(KC_LAYERNORM, s, 10.0 * rows * C, (x_f32 ? 5.0 : 3.0) * rows * C);
  const dim3 grid((rows + 3) / 4), block(256);
  const bool small = C <= 64 * 8 * 2;
  if (x_f32 && small && RAJNI_LN8_ROWS > 1 && rows >= 4096)
    hipLaunchKernelGGL((layernorm_fp8_rows_kernel<2, RAJNI_LN8_ROWS>), dim3((rows + 4 * RAJNI_LN8_ROWS - 1) / (4 * RAJNI_LN8_ROWS)), block, 0, s,
                       (const float*)x, xs, w, b, (unsigned char*)yq, yscale, hscale, wnorm, bmax, rows, C, eps);
  else if (x_f32 && small)
    hipLaunchKernelGGL((layernorm_fp8_kernel<float, 2>), grid, block, 0, s, (const float*)x, xs, w, b, (unsigned char*)yq,
                       yscale, hscale, wnorm, bmax, rows, C, eps);
  else if (x_f32)
    hipLaunchKernelGGL((layernorm_fp8_kernel<float, LN_MAX_CHUNKS>), grid, block, 0, s, (const float*)x, xs, w, b, (unsigned char*)yq,
                       yscale, hscale, wnorm, bmax, rows, C, eps);
  else if (small)
    hipLaunchKernelGGL((layernorm_fp8_kernel<bf16_t, 2>), grid, block, 0, s, (const bf16_t*)x, xs, w, b, (unsigned char*)yq,
                       yscale, hscale, wnorm, bmax, rows, C, eps);
  else
    hipLaunchKernelGGL((layernorm_fp8_kernel<bf16_t, LN_MAX_CHUNKS>), grid, block, 0, s, (const bf16_t*)x, xs, w, b, (unsigned char*)yq,
                       yscale, hscale, wnorm, bmax, rows, C, eps);
  RAJNI_CHECK_LAUNCH("layernorm_fp8_kernel");
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
