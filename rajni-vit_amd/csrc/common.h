// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave64, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/rajni_hip.h"
#include "../../include/rajni_hip_debug.h"

typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// 8 bf16 (one 16-byte chunk) -> 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// 8 consecutive elements of the activation dtype (bf16_t or float) <-> 8 floats
template <typename T> __device__ __forceinline__ void load8(const T* p, float* f);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* f) {
  unpack8(*reinterpret_cast<const uint4*>(p), f);
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* f);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* f) {
  *reinterpret_cast<uint4*>(p) = pack8(f);
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float* f) {
  reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}
// value as the activation dtype would hold it, and scalar load/store
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host side -----------------------------------------------------------------------------
void rajni_set_error(const char* fmt, ...);
// the device the calling thread's launches go to, and its CU count (hipDeviceProp_t::multiProcessorCount,
// read once per device; 256 on MI355X).  Per-device caches are sized RAJNI_MAX_DEVICES.
#define RAJNI_MAX_DEVICES 32
int rajni_current_device();
int rajni_num_cus();

enum KClass {
  KC_GEMM_BIAS = 0, KC_GEMM_GELU = 1, KC_GEMM_RESID = 2, KC_GEMM_PATCH = 3, KC_ATTENTION = 4,
  KC_LAYERNORM = 5, KC_SCORE_SELECT = 6, KC_IMPORTANCE = 7, KC_SELECT = 8, KC_GATHER = 9,
  KC_CLS_POS = 10, KC_OTHER = 11,
  KC_GEMM_RESID_SQ = 12,  // residual GEMM with K <= N (the attention projection): bound by its fp32-stream epilogue
  KC_GEMM8_BIAS = 13, KC_GEMM8_GELU = 14, KC_GEMM8_RESID = 15,  // fp8 x fp8 GEMMs (act_fp8 plans)
  KC_GEMM8_RESID_SQ = 16   // ... with K <= N (proj on e4m3 attention output)
};
// event bracket around one launch when the class is enabled
struct ProfScope {
  int kc; hipStream_t s; void* rec;
  ProfScope(int kclass, hipStream_t stream, double flops, double bytes);
  ~ProfScope();
};

#define RAJNI_CHECK_LAUNCH(name)                                              \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      rajni_set_error("%s launch failed: %s", name, hipGetErrorString(e__));  \
      return RAJNI_ERR_LAUNCH;                                                \
    }                                                                         \
  } while (0)

#define RAJNI_REQUIRE(cond, code, ...)       \
  do {                                       \
    if (!(cond)) {                           \
      rajni_set_error(__VA_ARGS__);          \
      return code;                           \
    }                                        \
  } while (0)

// diagnostic builds only (-DRAJNI_GEMM_STAMPS / -DRAJNI_ATTN_STAMPS): device buffer for s_memtime stamps
extern unsigned long long* rajni_g_stamps;

// internal launchers shared between the per-op ABI and the whole-forward plan
int launch_linear(const rajni_linear_args& a, hipStream_t s);
int launch_patch_embed(const void* images, const void* w, const float* bias, const void* cls,
                       const void* pos, int pos_has_cls, void* x, int out_f32, int B, int Cin, int S,
                       int P, int C, int dtype, void* ws, size_t ws_bytes, hipStream_t s);
size_t patch_embed_workspace_bytes(int B, int Cin, int S, int P, int dtype);   // 0: im2col fused into the GEMM loads
int launch_layernorm(const void* x, long xs, const float* w, const float* b, void* y, int rows,
                     int C, float eps, int x_f32, int dtype, hipStream_t s);
int launch_layernorm_fp8(const void* x, long xs, const float* w, const float* b, void* yq, float* yscale,
                         float* hscale, float wnorm, float bmax, int rows, int C, float eps, int x_f32, hipStream_t s);
int launch_attention(const void* qkv, const int32_t* keep_idx, void* out, int B, int n_src, int np,
                     int H, int D, float scale, int dtype, hipStream_t s);
int launch_attention_fp8(const void* qkv, const int32_t* keep_idx, void* out_q, float out_scale, float* row_scale,
                         int B, int n_src, int np, int H, int D, float scale, hipStream_t s);
int launch_attention_cls(const void* qkv, void* out, int B, int N, int H, int D, float scale, int dtype,
                         hipStream_t s, float q_scale = 0.f);   // q_scale > 0: the row passes through rajni_attention_fp8's e4m3 rounding
int launch_score_select(const void* qkv, const void* scores_in, int B, int N, int H, int D,
                        float eps, int keep, void* scores_out, int32_t* keep_idx,
                        void* next_scores, int dtype, hipStream_t s);
int launch_gather_rows(const void* src, const int32_t* idx, void* dst, int B, int n_src, int n_dst,
                       int row_bytes, hipStream_t s);
