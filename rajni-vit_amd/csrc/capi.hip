// extern "C" surface of librajni_hip.so (include/rajni_hip.h): argument checks, error strings,
// the HIP-event measurement hooks, and thin wrappers over the launchers.
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "common.h"

namespace {
thread_local char g_err[512] = "";

struct ProfRec { int kc; hipEvent_t e0, e1; double flops, bytes; };
unsigned g_prof_mask = 0;
std::vector<ProfRec> g_pending;
std::vector<hipEvent_t> g_pool;
long long g_launches[RAJNI_NUM_KCLASS];
double g_ms[RAJNI_NUM_KCLASS], g_flops[RAJNI_NUM_KCLASS], g_bytes[RAJNI_NUM_KCLASS];

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
const char* const kNames[RAJNI_NUM_KCLASS] = {
    "gemm_bf16_tn<bias>", "gemm_bf16_tn<bias,gelu>", "gemm_bf16_tn<bias,ls,resid>",
    "gemm_bf16_tn<patch>", "attn_bf16_d64", "layernorm_kernel", "score_select_kernel<fused>",
    "score_select_kernel<scores>", "score_select_kernel<select>", "gather_rows_kernel",
    "cls_pos_kernel", "other", "gemm_bf16_tn<bias,ls,resid> K<=N",
    "gemm_f8_tn<bias>", "gemm_f8_tn<bias,gelu,requant>", "gemm_f8_tn<bias,ls,resid>",
    "gemm_f8_tn<bias,ls,resid> K<=N"};
}  // namespace

unsigned long long* rajni_g_stamps = nullptr;

int rajni_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RAJNI_MAX_DEVICES) return 0;
  return dev;
}
int rajni_num_cus() {
  static int cus[RAJNI_MAX_DEVICES] = {};     // 0 = not read yet (benign race: every writer stores the same value)
  const int dev = rajni_current_device();
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

void rajni_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

ProfScope::ProfScope(int kclass, hipStream_t stream, double flops, double bytes)
    : kc(kclass), s(stream), rec(nullptr) {
  if (!(g_prof_mask & (1u << kclass))) return;
  ProfRec r{kclass, get_event(), get_event(), flops, bytes};
  if (!r.e0 || !r.e1) return;
  (void)hipEventRecord(r.e0, s);
  g_pending.push_back(r);
  rec = reinterpret_cast<void*>(g_pending.size());  // index + 1
}
ProfScope::~ProfScope() {
  if (!rec) return;
  const size_t i = reinterpret_cast<size_t>(rec) - 1;
  (void)hipEventRecord(g_pending[i].e1, s);
}

extern "C" {

int rajni_abi_version(void) { return RAJNI_ABI_VERSION; }
const char* rajni_last_error(void) { return g_err; }

int rajni_device_check(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    rajni_set_error("no HIP device visible: %s", hipGetErrorString(e));
    return RAJNI_ERR_LAUNCH;
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    rajni_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
    return RAJNI_ERR_LAUNCH;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    rajni_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
    return RAJNI_ERR_UNSUPPORTED;
  }
  return RAJNI_OK;
}

#define NEED_DTYPE(name)                                                                     \
  RAJNI_REQUIRE(dtype == RAJNI_BF16 || dtype == RAJNI_F32, RAJNI_ERR_INVALID, name ": bad dtype %d", dtype)

int rajni_importance(const void* qkv, void* scores_out, int B, int N, int H, int D, float eps,
                     int dtype, rajni_stream_t stream) {
  NEED_DTYPE("rajni_importance");
  RAJNI_REQUIRE(qkv && scores_out, RAJNI_ERR_INVALID, "rajni_importance: null pointer");
  return launch_score_select(qkv, nullptr, B, N, H, D, eps, 0, scores_out, nullptr, nullptr, dtype,
                             (hipStream_t)stream);
}

int rajni_select_topk(const void* scores, int B, int N, int keep, int32_t* keep_idx,
                      void* next_scores, int dtype, rajni_stream_t stream) {
  NEED_DTYPE("rajni_select_topk");
  RAJNI_REQUIRE(scores && keep_idx, RAJNI_ERR_INVALID, "rajni_select_topk: null pointer");
  RAJNI_REQUIRE(keep >= 1, RAJNI_ERR_INVALID, "rajni_select_topk: keep must be >= 1");
  return launch_score_select(nullptr, scores, B, N, 0, 0, 0.f, keep, nullptr, keep_idx, next_scores, dtype,
                             (hipStream_t)stream);
}

int rajni_score_select(const void* qkv, int B, int N, int H, int D, float eps, int keep,
                       void* scores_out, int32_t* keep_idx, void* next_scores, int dtype,
                       rajni_stream_t stream) {
  NEED_DTYPE("rajni_score_select");
  RAJNI_REQUIRE(qkv && keep_idx, RAJNI_ERR_INVALID, "rajni_score_select: null pointer");
  RAJNI_REQUIRE(keep >= 1, RAJNI_ERR_INVALID, "rajni_score_select: keep must be >= 1");
  return launch_score_select(qkv, nullptr, B, N, H, D, eps, keep, scores_out, keep_idx, next_scores, dtype,
                             (hipStream_t)stream);
}

int rajni_gather_rows(const void* src, const int32_t* idx, void* dst, int B, int n_src, int n_dst,
                      int row_elems, int dtype, rajni_stream_t stream) {
  RAJNI_REQUIRE(dtype == RAJNI_BF16 || dtype == RAJNI_F32, RAJNI_ERR_INVALID, "rajni_gather_rows: bad dtype");
  const int es = dtype == RAJNI_BF16 ? 2 : 4;
  return launch_gather_rows(src, idx, dst, B, n_src, n_dst, row_elems * es, (hipStream_t)stream);
}

int rajni_attention(const void* qkv, const int32_t* keep_idx, void* out, int B, int n_src, int np,
                    int H, int D, float scale, int dtype, rajni_stream_t stream) {
  NEED_DTYPE("rajni_attention");
  return launch_attention(qkv, keep_idx, out, B, n_src, np, H, D, scale, dtype, (hipStream_t)stream);
}

extern "C" int rajni_attention_fp8(const void* qkv, const int32_t* keep_idx, void* out_q, float out_scale, float* row_scale,
                                   int B, int n_src, int np, int H, int D, float scale, rajni_stream_t stream) {
  return launch_attention_fp8(qkv, keep_idx, out_q, out_scale, row_scale, B, n_src, np, H, D, scale, (hipStream_t)stream);
}

int rajni_layernorm(const void* x, long x_row_stride, const float* w, const float* b, void* y,
                    int rows, int C, float eps, int dtype, int x_f32, rajni_stream_t stream) {
  NEED_DTYPE("rajni_layernorm");
  return launch_layernorm(x, x_row_stride, w, b, y, rows, C, eps, x_f32, dtype, (hipStream_t)stream);
}

int rajni_layernorm_fp8(const void* x, long x_row_stride, const float* w, const float* b, void* y_q,
                        float* y_scale, float* hid_scale, float w1_rownorm_max, float b1_absmax,
                        int rows, int C, float eps, int x_f32, rajni_stream_t stream) {
  return launch_layernorm_fp8(x, x_row_stride, w, b, y_q, y_scale, hid_scale, w1_rownorm_max, b1_absmax, rows, C, eps,
                              x_f32, (hipStream_t)stream);
}

int rajni_linear(const rajni_linear_args* args, rajni_stream_t stream) {
  RAJNI_REQUIRE(args != nullptr, RAJNI_ERR_INVALID, "rajni_linear: null args");
  return launch_linear(*args, (hipStream_t)stream);
}

int rajni_patch_embed(const void* images, const void* w, const float* bias, const void* cls,
                      const void* pos, int pos_has_cls, void* x, int x_f32, int B, int Cin, int S,
                      int P, int C, int dtype, void* workspace, size_t workspace_bytes, rajni_stream_t stream) {
  NEED_DTYPE("rajni_patch_embed");
  return launch_patch_embed(images, w, bias, cls, pos, pos_has_cls, x, x_f32, B, Cin, S, P, C, dtype,
                            workspace, workspace_bytes, (hipStream_t)stream);
}
size_t rajni_patch_embed_workspace_bytes(int B, int Cin, int S, int P, int dtype) {
  return patch_embed_workspace_bytes(B, Cin, S, P, dtype);
}

void rajni_profile_enable(unsigned mask) { g_prof_mask = mask; }
const char* rajni_profile_class_name(int k) {
  return (k >= 0 && k < RAJNI_NUM_KCLASS) ? kNames[k] : "";
}
void rajni_profile_reset(void) {
  for (auto& r : g_pending) { (void)hipEventSynchronize(r.e1); g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
  g_pending.clear();
  for (int i = 0; i < RAJNI_NUM_KCLASS; ++i) { g_launches[i] = 0; g_ms[i] = g_flops[i] = g_bytes[i] = 0.0; }
}
int rajni_profile_collect(long long* launches, double* ms, double* flops, double* bytes) {
  for (auto& r : g_pending) {
    hipError_t e = hipEventSynchronize(r.e1);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.e0, r.e1);
    if (e != hipSuccess) {
      rajni_set_error("profile: %s", hipGetErrorString(e));
      return RAJNI_ERR_LAUNCH;
    }
    g_launches[r.kc] += 1; g_ms[r.kc] += t; g_flops[r.kc] += r.flops; g_bytes[r.kc] += r.bytes;
    g_pool.push_back(r.e0); g_pool.push_back(r.e1);
  }
  g_pending.clear();
  for (int i = 0; i < RAJNI_NUM_KCLASS; ++i) {
    if (launches) launches[i] = g_launches[i];
    if (ms) ms[i] = g_ms[i];
    if (flops) flops[i] = g_flops[i];
    if (bytes) bytes[i] = g_bytes[i];
  }
  return RAJNI_OK;
}

}  // extern "C"
