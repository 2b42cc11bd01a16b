// Whole-forward orchestration: RAJNIViTWrapper.forward (reference model.py:30-69) as one host call
// that enqueues every kernel on the caller's stream.  Token counts are data independent,
// so all shapes are known up front: no allocation, no host sync, no device->host traffic inside.
//
// With plan.act_fp8 (opt-in): LN1 / LN2 emit per-row-scaled e4m3 rows, QKV / FC1 / FC2 run on the fp8 matrix pipe
// (gemm_f8.h) and FC1's GELU epilogue re-quantises the hidden activations; attention, proj, the residual stream,
// patch embed and head are unchanged.
// Per block:  LN1 -> QKV GEMM (all N tokens) -> [score+select] -> attention on kept tokens (gather
// fused into its loads) -> proj GEMM whose epilogue gathers the residual row, applies LayerScale and
// adds -> LN2 -> FC1 GEMM + GELU -> FC2 GEMM + LayerScale + residual (in place).
// The reference's three gathers (qkv, scores, x) never exist as kernels here.
#include "common.h"

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Workspace {
  char *xa, *xb, *xn, *qkv, *att, *hid, *clsn, *scf, *cols;
  float *xs, *hs;          // act_fp8 plans: per-row scales of the e4m3 LayerNorm output / MLP hidden activations
  size_t total, cols_bytes;
};

Workspace carve(const rajni_vit_plan& p) {
  const size_t gw = p.img_size / p.patch_size, n0 = gw * gw + 1;
  const size_t rows = (size_t)p.B * n0, es = p.dtype == RAJNI_F32 ? 4 : 2, xs = (p.dtype == RAJNI_F32 || !p.resid_bf16) ? 4 : 2;
  Workspace w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
  const size_t oxa = take(rows * p.C * xs), oxb = take(rows * p.C * xs), oxn = take(rows * p.C * es);
  const size_t oqkv = take(rows * 3 * p.C * es), oatt = take(rows * p.C * es);
  const size_t ohid = take(rows * p.hidden * es), ocls = take((size_t)p.B * p.C * es);
  const size_t oscf = take(rows * es);
  w.cols_bytes = patch_embed_workspace_bytes(p.B, p.in_chans, p.img_size, p.patch_size, p.dtype);   // 0 when fused
  const size_t ocols = take(w.cols_bytes);
  const size_t oxs = take(p.act_fp8 ? rows * sizeof(float) : 0), ohs = take(p.act_fp8 ? rows * sizeof(float) : 0);
  char* base = (char*)p.workspace;
  w.xa = base + oxa; w.xb = base + oxb; w.xn = base + oxn; w.qkv = base + oqkv; w.att = base + oatt;
  w.hid = base + ohid; w.clsn = base + ocls; w.scf = base + oscf; w.cols = base + ocols;
  w.xs = reinterpret_cast<float*>(base + oxs); w.hs = reinterpret_cast<float*>(base + ohs);
  w.total = off;
  return w;
}

// next_scores[b, j] = scores[b, idx[b, j]]   (attention.py:58) - used with a forced selection
template <typename T>
__global__ void carry_scores_kernel(const T* scores, const int* idx, T* out, int B, int N, int np) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * np) return;
  const int b = i / np;
  out[i] = scores[(long)b * N + idx[i]];
}

int check_plan(const rajni_vit_plan& p) {
  RAJNI_REQUIRE(p.dtype == RAJNI_BF16 || p.dtype == RAJNI_F32, RAJNI_ERR_INVALID, "rajni_vit_forward: bad dtype %d", p.dtype);
  RAJNI_REQUIRE(p.B > 0 && p.depth > 0 && p.blocks != nullptr, RAJNI_ERR_INVALID, "rajni_vit_forward: bad plan");
  RAJNI_REQUIRE(p.C == p.H * p.D && p.D >= 8 && p.D <= 128 && p.D % 8 == 0, RAJNI_ERR_UNSUPPORTED,
                "rajni_vit_forward: need C == H*D and a head dim that is a multiple of 8 up to 128 (C=%d H=%d D=%d)", p.C, p.H, p.D);
  RAJNI_REQUIRE(p.C % 64 == 0 && p.hidden % 64 == 0, RAJNI_ERR_UNSUPPORTED,
                "rajni_vit_forward: C and hidden must be multiples of 64");
  RAJNI_REQUIRE(p.patch_w && p.cls_token && p.pos_embed && p.norm_w && p.norm_b && p.head_w,
                RAJNI_ERR_INVALID, "rajni_vit_forward: null weight pointer");
  if (p.act_fp8) {
    RAJNI_REQUIRE(p.dtype == RAJNI_BF16, RAJNI_ERR_UNSUPPORTED, "rajni_vit_forward: act_fp8 needs a bf16 model");
    RAJNI_REQUIRE(p.C % 256 == 0 && p.hidden % 256 == 0 && p.C >= 512, RAJNI_ERR_UNSUPPORTED,
                  "rajni_vit_forward: act_fp8 needs C %% 256 == 0, C >= 512 and hidden %% 256 == 0 (C=%d hidden=%d)", p.C, p.hidden);
    for (int i = 0; i < p.depth; ++i)
      RAJNI_REQUIRE(p.blocks[i].qkv_s && p.blocks[i].fc1_s && p.blocks[i].fc2_s, RAJNI_ERR_INVALID,
                    "rajni_vit_forward: act_fp8 needs e4m3 qkv / fc1 / fc2 weights with scales (block %d)", i);
  }
  return RAJNI_OK;
}

}  // namespace

extern "C" size_t rajni_vit_workspace_bytes(const rajni_vit_plan* plan) {
  if (!plan || plan->patch_size <= 0) return 0;
  rajni_vit_plan tmp = *plan;
  tmp.workspace = nullptr;
  return carve(tmp).total;
}

extern "C" int rajni_vit_forward(const rajni_vit_plan* plan, const void* images, void* logits,
                                 rajni_stream_t stream) {
  RAJNI_REQUIRE(plan && images && logits, RAJNI_ERR_INVALID, "rajni_vit_forward: null pointer");
  const rajni_vit_plan& p = *plan;
  int rc = check_plan(p);
  if (rc != RAJNI_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const Workspace w = carve(p);
  RAJNI_REQUIRE(p.workspace != nullptr && p.workspace_bytes >= w.total, RAJNI_ERR_INVALID,
                "rajni_vit_forward: workspace too small (%zu < %zu)", p.workspace_bytes, w.total);
  const int B = p.B, C = p.C;
  const int gw = p.img_size / p.patch_size;
  int N = gw * gw + 1;

  const int dt = p.dtype;
  const int sf32 = (dt == RAJNI_BF16 && !p.resid_bf16) ? 1 : 0;  // bf16 model with an fp32 residual stream
  rc = launch_patch_embed(images, p.patch_w, p.patch_b, p.cls_token, p.pos_embed, p.pos_has_cls,
                          w.xa, sf32, B, p.in_chans, p.img_size, p.patch_size, C, dt, w.cols, w.cols_bytes, s);
  if (rc != RAJNI_OK) return rc;

  char* cur = w.xa;
  char* oth = w.xb;
  const void* carried = nullptr;  // scores of the tokens currently in `cur` (model.py:39,53,63)

  for (int i = 0; i < p.depth; ++i) {
    const rajni_block& blk = p.blocks[i];
    if (p.token_counts) p.token_counts[i] = N;  // model.py:43
    const int M = B * N;
    // ---- norm1 + qkv on ALL N tokens (model.py:51, attention.py:21-22)
    if (p.act_fp8) rc = launch_layernorm_fp8(cur, C, blk.norm1_w, blk.norm1_b, w.xn, w.xs, nullptr, 0.f, 0.f, M, C, p.ln_eps, sf32, s);
    else rc = launch_layernorm(cur, C, blk.norm1_w, blk.norm1_b, w.xn, M, C, p.ln_eps, sf32, dt, s);
    if (rc != RAJNI_OK) return rc;
    rajni_linear_args g{};
    g.dtype = dt;
    g.x = w.xn; g.lda = C; g.w = blk.qkv_w; g.ldw = C; g.bias = blk.qkv_b; g.w_scale = blk.qkv_s;
    if (p.act_fp8) g.x_scale = w.xs;
    g.y = w.qkv; g.ldc = 3 * C; g.M = M; g.N = 3 * C; g.K = C; g.epilogue = RAJNI_EPI_BIAS;
    rc = launch_linear(g, s);
    if (rc != RAJNI_OK) return rc;

    if (p.cls_only_last_block && i == p.depth - 1 && blk.keep == 0 && N > 1) {
      // ---- last block, not a pruning stage, caller opted in: only x[:, 0] reaches the head (model.py:65-66),
      //      so attention runs for the CLS query alone (over all N keys) and proj / MLP on the B CLS rows
      // (an act_fp8 block whose all-rows form would emit e4m3 attention rows rounds the CLS row the same way)
      const bool att8c = p.act_fp8 && blk.attn_out_scale > 0.f && blk.proj_s != nullptr && p.D == 64 && N <= 224;
      rc = launch_attention_cls(w.qkv, w.att, B, N, p.H, p.D, p.attn_scale, dt, s, att8c ? blk.attn_out_scale : 0.f);
      if (rc != RAJNI_OK) return rc;
      g = rajni_linear_args{};
      g.dtype = dt;
      g.x = w.att; g.lda = C; g.w = blk.proj_w; g.ldw = C; g.bias = blk.proj_b; g.gamma = blk.ls1; g.w_scale = blk.proj_s;
      g.resid = cur; g.ldr = (long)N * C;            // residual row of image b = its CLS row
      g.y = oth; g.ldc = C; g.M = B; g.N = C; g.K = C; g.epilogue = RAJNI_EPI_BIAS_RESID; g.stream_f32 = sf32;
      rc = launch_linear(g, s);
      if (rc != RAJNI_OK) return rc;
      { char* t = cur; cur = oth; oth = t; }
      N = 1;                                          // the stream is now [B, 1, C]
      if (p.act_fp8) rc = launch_layernorm_fp8(cur, C, blk.norm2_w, blk.norm2_b, w.xn, w.xs, w.hs, blk.fc1_rownorm_max,
                                               blk.fc1_bias_absmax, B, C, p.ln_eps, sf32, s);
      else rc = launch_layernorm(cur, C, blk.norm2_w, blk.norm2_b, w.xn, B, C, p.ln_eps, sf32, dt, s);
      if (rc != RAJNI_OK) return rc;
      g = rajni_linear_args{};
      g.dtype = dt;
      g.x = w.xn; g.lda = C; g.w = blk.fc1_w; g.ldw = C; g.bias = blk.fc1_b; g.w_scale = blk.fc1_s;
      g.y = w.hid; g.ldc = p.hidden; g.M = B; g.N = p.hidden; g.K = C; g.epilogue = RAJNI_EPI_BIAS_GELU;
      if (p.act_fp8) { g.x_scale = w.xs; g.y_scale = w.hs; }
      rc = launch_linear(g, s);
      if (rc != RAJNI_OK) return rc;
      g = rajni_linear_args{};
      g.dtype = dt;
      g.x = w.hid; g.lda = p.hidden; g.w = blk.fc2_w; g.ldw = p.hidden; g.bias = blk.fc2_b; g.gamma = blk.ls2; g.w_scale = blk.fc2_s;
      if (p.act_fp8) g.x_scale = w.hs;
      g.resid = cur; g.ldr = C; g.y = cur; g.ldc = C; g.M = B; g.N = C; g.K = p.hidden;
      g.epilogue = RAJNI_EPI_BIAS_RESID; g.stream_f32 = sf32;
      rc = launch_linear(g, s);
      if (rc != RAJNI_OK) return rc;
      break;
    }

    int Np = N;
    const int32_t* idx = nullptr;
    if (blk.keep > 0) {  // scheduled block (model.py:50)
      RAJNI_REQUIRE(blk.keep <= N - 1, RAJNI_ERR_INVALID, "block %d: keep=%d but only %d patch tokens", i, blk.keep, N - 1);
      RAJNI_REQUIRE(blk.keep_idx && blk.next_scores, RAJNI_ERR_INVALID, "block %d: keep_idx/next_scores buffers missing", i);
      Np = blk.keep + 1;
      const bool recompute = blk.update || carried == nullptr;  // attention.py:25
      if (blk.forced_keep_idx) {
        const void* full = carried;
        if (recompute) {
          void* dst = blk.scores ? blk.scores : (void*)w.scf;
          rc = launch_score_select(w.qkv, nullptr, B, N, p.H, p.D, 1e-6f, 0, dst, nullptr, nullptr, dt, s);
          if (rc != RAJNI_OK) return rc;
          full = dst;
        }
        const int n = B * Np;
        if (dt == RAJNI_F32)
          hipLaunchKernelGGL(carry_scores_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s,
                             (const float*)full, blk.forced_keep_idx, (float*)blk.next_scores, B, N, Np);
        else
          hipLaunchKernelGGL(carry_scores_kernel<bf16_t>, dim3((n + 255) / 256), dim3(256), 0, s,
                             (const bf16_t*)full, blk.forced_keep_idx, (bf16_t*)blk.next_scores, B, N, Np);
        RAJNI_CHECK_LAUNCH("carry_scores_kernel");
        idx = blk.forced_keep_idx;
      } else {
        if (recompute)
          rc = launch_score_select(w.qkv, nullptr, B, N, p.H, p.D, 1e-6f, blk.keep, blk.scores,
                                   blk.keep_idx, blk.next_scores, dt, s);
        else
          rc = launch_score_select(nullptr, carried, B, N, 0, 0, 0.f, blk.keep, nullptr,
                                   blk.keep_idx, blk.next_scores, dt, s);
        if (rc != RAJNI_OK) return rc;
        idx = blk.keep_idx;
      }
      carried = blk.next_scores;  // attention.py:58,60
    } else {
      carried = nullptr;          // model.py:63
    }

    // ---- attention on the kept tokens, gather fused (attention.py:42-54)
    // act_fp8 plans: e4m3 output rows + their (one) scale into w.xs - norm1's row scales there were consumed by QKV -
    // wherever the persistent head-dim-64 kernel serves the launch (rajni_attention_fp8); proj then runs fp8 x fp8
    const bool att8 = p.act_fp8 && blk.attn_out_scale > 0.f && blk.proj_s != nullptr && p.D == 64 && Np <= 224;
    if (att8) rc = launch_attention_fp8(w.qkv, idx, w.att, blk.attn_out_scale, w.xs, B, N, Np, p.H, p.D, p.attn_scale, s);
    else rc = launch_attention(w.qkv, idx, w.att, B, N, Np, p.H, p.D, p.attn_scale, dt, s);
    if (rc != RAJNI_OK) return rc;

    // ---- proj + (gathered) residual + LayerScale (attention.py:55-56, model.py:55-58)
    const int Mp = B * Np;
    g = rajni_linear_args{};
    g.dtype = dt;
    g.x = w.att; g.lda = C; g.w = blk.proj_w; g.ldw = C; g.bias = blk.proj_b; g.gamma = blk.ls1; g.w_scale = blk.proj_s;
    g.resid = cur; g.ldr = C; g.M = Mp; g.N = C; g.K = C; g.epilogue = RAJNI_EPI_BIAS_RESID; g.stream_f32 = sf32;
    if (att8) g.x_scale = w.xs;
    if (idx) {
      g.r_idx = idx; g.r_np = Np; g.r_nsrc = N;
      g.y = oth; g.ldc = C;
      rc = launch_linear(g, s);
      char* t = cur; cur = oth; oth = t;
    } else {
      g.y = cur; g.ldc = C;  // each element is read and written by the same lane: in place is safe
      rc = launch_linear(g, s);
    }
    if (rc != RAJNI_OK) return rc;
    N = Np;

    // ---- MLP (model.py:59): norm2 -> fc1 + GELU -> fc2 + LayerScale + residual (in place)
    if (p.act_fp8) rc = launch_layernorm_fp8(cur, C, blk.norm2_w, blk.norm2_b, w.xn, w.xs, w.hs, blk.fc1_rownorm_max,
                                                  blk.fc1_bias_absmax, Mp, C, p.ln_eps, sf32, s);
    else rc = launch_layernorm(cur, C, blk.norm2_w, blk.norm2_b, w.xn, Mp, C, p.ln_eps, sf32, dt, s);
    if (rc != RAJNI_OK) return rc;
    g = rajni_linear_args{};
    g.dtype = dt;
    g.x = w.xn; g.lda = C; g.w = blk.fc1_w; g.ldw = C; g.bias = blk.fc1_b; g.w_scale = blk.fc1_s;
    g.y = w.hid; g.ldc = p.hidden; g.M = Mp; g.N = p.hidden; g.K = C; g.epilogue = RAJNI_EPI_BIAS_GELU;
    if (p.act_fp8) { g.x_scale = w.xs; g.y_scale = w.hs; }   // e4m3 in, e4m3 out (per-row scales)
    rc = launch_linear(g, s);
    if (rc != RAJNI_OK) return rc;
    g = rajni_linear_args{};
    g.dtype = dt;
    g.x = w.hid; g.lda = p.hidden; g.w = blk.fc2_w; g.ldw = p.hidden; g.bias = blk.fc2_b; g.gamma = blk.ls2; g.w_scale = blk.fc2_s;
    if (p.act_fp8) g.x_scale = w.hs;
    g.resid = cur; g.ldr = C; g.y = cur; g.ldc = C; g.M = Mp; g.N = C; g.K = p.hidden;
    g.epilogue = RAJNI_EPI_BIAS_RESID; g.stream_f32 = sf32;
    rc = launch_linear(g, s);
    if (rc != RAJNI_OK) return rc;
  }

  // ---- final norm on the CLS rows only (LN is per token; model.py:65-66) + head
  rc = launch_layernorm(cur, (long)N * C, p.norm_w, p.norm_b, w.clsn, B, C, p.ln_eps, sf32, dt, s);
  if (rc != RAJNI_OK) return rc;
  rajni_linear_args g{};
  g.dtype = dt;
  g.x = w.clsn; g.lda = C; g.w = p.head_w; g.ldw = C; g.bias = p.head_b;
  const int ld = p.logits_ld > 0 ? p.logits_ld : p.num_classes;
  RAJNI_REQUIRE(ld % 8 == 0 && ld >= p.num_classes, RAJNI_ERR_INVALID,
                "rajni_vit_forward: logits row stride must be a multiple of 8 and >= num_classes (%d)", ld);
  g.y = logits; g.ldc = ld; g.M = B; g.N = p.num_classes; g.K = C; g.epilogue = RAJNI_EPI_BIAS;
  return launch_linear(g, s);
}
