/*
 * rajni_hip.h - C ABI of librajni_hip.so: the MI355X (gfx950) token-pruning forward path.
 *
 * The reference (dRaniwal/RAJNI-ViT) has no FFI of its own: its hot path is Python calling ATen
 * (SURVEY.md section 8b).  Each entry point below therefore replaces a *Python* call site of the reference,
 * cited as file:line relative to /root/reference/rajni/.  Conventions:
 *   - plain C, no torch types; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller owns all buffers, nothing is allocated or freed inside, no host sync inside;
 *   - every launch goes to the hipStream_t given (pass torch's current stream);
 *   - returns RAJNI_OK (0) or an error code; rajni_last_error() gives a host string (thread local);
 *   - `dtype` is the activation/weight element type of the model: RAJNI_BF16 (the fast path: bf16
 *     MFMA, fp32 accumulation) or RAJNI_F32 (accuracy path: every tensor fp32, v_mfma_f32_16x16x4_f32
 *     GEMMs, VALU attention; ~1/16 of the bf16 MFMA rate).
 *   - activations are row-major [B, N, C]; qkv is [B, N, 3*C] with the last axis laid out
 *     [3][H][D] (timm convention; importance.py:14, attention.py:46-47);
 *   - keep_idx is int32 on the device ([B, keep+1], slot 0 = CLS = 0, rest ascending); the Python
 *     surface widens to int64 to match attention.py:38.
 */
#ifndef RAJNI_HIP_H
#define RAJNI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rajni_stream_t; /* hipStream_t */

enum { RAJNI_F32 = 0, RAJNI_BF16 = 1 };

enum {
  RAJNI_OK = 0,
  RAJNI_ERR_INVALID = 1,     /* bad argument (null pointer, shape, alignment) */
  RAJNI_ERR_UNSUPPORTED = 2, /* valid request the build does not implement (dtype, head dim ...) */
  RAJNI_ERR_LAUNCH = 3       /* HIP runtime error at launch */
};

/* epilogues of rajni_linear */
enum {
  RAJNI_EPI_BIAS = 0,      /* y = x W^T + b                                  attention.py:22 (qkv) */
  RAJNI_EPI_BIAS_GELU = 1, /* y = gelu_erf(x W^T + b)                        model.py:59 (mlp.fc1+act) */
  RAJNI_EPI_BIAS_RESID = 2 /* y = resid[row or gathered row] + gamma*(x W^T + b)
                              attention.py:55 + model.py:55-58 (proj, gather x, ls1, add);
                              model.py:59 (mlp.fc2, ls2, add)                                       */
};

#define RAJNI_ABI_VERSION 8 /* bumped whenever a struct or an entry point changes; checked by the ctypes binding */
int rajni_abi_version(void); /* == RAJNI_ABI_VERSION of the header the library was built from */
const char* rajni_last_error(void);
/* 0 when a gfx950 device is usable by this process, else an error code (message in last_error) */
int rajni_device_check(void);

/* ---- a1: compute_importance(qkv, num_heads, eps)                         importance.py:4-34 ----
 * scores_out [B,N] in `dtype` (the reference returns qkv's dtype, importance.py:34). */
int rajni_importance(const void* qkv, void* scores_out, int B, int N, int H, int D, float eps,
                     int dtype, rajni_stream_t stream);

/* ---- a5,a6,a10: top-k + sort + CLS prepend + score carry                 attention.py:31-39,58 ----
 * scores [B,N] in `dtype`; keep = max(1, int(keep_ratio*(N-1))) is computed by the caller
 * (attention.py:31-32, Python-double semantics).  Tie rule (the reference leaves it unspecified):
 * larger score first, then lower index; NaN ranks as +inf.
 * keep_idx [B,keep+1] int32; next_scores [B,keep+1] in `dtype` (may be NULL). */
int rajni_select_topk(const void* scores, int B, int N, int keep, int32_t* keep_idx,
                      void* next_scores, int dtype, rajni_stream_t stream);

/* ---- a1+a6+a10 fused (one launch per pruning stage): scores never leave the chip between the
 * two steps.  scores_out may be NULL. */
int rajni_score_select(const void* qkv, int B, int N, int H, int D, float eps, int keep,
                       void* scores_out, int32_t* keep_idx, void* next_scores, int dtype,
                       rajni_stream_t stream);

/* ---- a7,a13: torch.gather(t, 1, keep_idx[..., None].expand(...))   attention.py:42-43, model.py:55-56
 * src [B,n_src,row_elems] -> dst [B,n_dst,row_elems]; row_elems*elem_size must be a multiple of 16. */
int rajni_gather_rows(const void* src, const int32_t* idx, void* dst, int B, int n_src, int n_dst,
                      int row_elems, int dtype, rajni_stream_t stream);

/* ---- a8: softmax(q k^T * scale) v on the kept tokens                      attention.py:46-54 ----
 * qkv [B,n_src,3*H*D]; keep_idx [B,np] int32 or NULL (NULL: identity, np == n_src - the unpruned
 * block of model.py:62).  The row gather of attention.py:42-43 is fused into the tile loads.
 * out [B,np,H*D].  D % 8 == 0, 8 <= D <= 128 (D = 64 has the tuned
 * kernels; other head dims take a general MFMA kernel). */
int rajni_attention(const void* qkv, const int32_t* keep_idx, void* out, int B, int n_src, int np,
                    int H, int D, float scale, int dtype, rajni_stream_t stream);
/* The same attention (bf16 qkv) with its output rows quantised for an fp8 x fp8 proj (opt-in "fp8_mfma" format; the
 * reference has no fp8 semantics - this is the build's rule):
 *   out_q[b,q,c] = e4m3_rne_sat(attn[b,q,c] * (1 / out_scale))  as bytes [B,np,H*D];  row_scale[b*np + q] = out_scale
 * (the per-row dequantisation scales the proj launch takes as rajni_linear_args.x_scale).  ONE scale per launch: an
 * attention row spans H (image, head) work items, so no item can know the row's maximum; the caller passes a bound -
 * attention output is a convex combination of V rows, |V[j,c]| <= ||ln1(x)[j]||_2 ||Wv[c]||_2 + |bv[c]| and
 * ||ln1(x)[j]||_2 <= sqrt(C) max|gamma1| + ||beta1||_2 (a normalised row has norm <= sqrt(C)), so
 *   out_scale = (1.0625 * (sqrt(C) * max|gamma1| + ||beta1||_2) * max_c ||Wv[c]||_2 + max|bv|) / 448
 * (the 1.0625 covers the e4m3 rounding of the LayerNorm rows; the conversion saturates).  e4m3 is a floating-point
 * format: a bound a few binades above the true maximum costs range at the bottom, not precision.
 * Head dim 64 and np <= 224 only (RAJNI_ERR_UNSUPPORTED otherwise). */
int rajni_attention_fp8(const void* qkv, const int32_t* keep_idx, void* out_q, float out_scale, float* row_scale,
                        int B, int n_src, int np, int H, int D, float scale, rajni_stream_t stream);

/* ---- LayerNorm over the last axis (blk.norm1 / norm2 / m.norm)            model.py:51,59,65 ----
 * x rows are `x_row_stride` elements apart (lets the final norm read CLS rows only), y is dense
 * [rows, C] in `dtype`.  x is `dtype`, or fp32 when x_f32 != 0 (the fp32 residual stream).
 * w,b are fp32 [C]. C % 8 == 0. */
int rajni_layernorm(const void* x, long x_row_stride, const float* w, const float* b, void* y,
                    int rows, int C, float eps, int dtype, int x_f32, rajni_stream_t stream);
/* The same LayerNorm with its output quantised per row for the fp8 matrix pipe (opt-in "fp8_mfma" format, BASELINE
 * configs[4]; the reference has no fp8 semantics - these are the build's, SURVEY 7 "hard parts"):
 *   y_scale[r] = max_c |ln(x)[r,c]| / 448   (1 for an all-zero row),   y_q[r,c] = e4m3_rne_sat(ln(x)[r,c] * (1 / y_scale[r]))
 * with ln(x) evaluated in fp32; y_q is [rows, C] bytes (OCP e4m3 "fn").  When hid_scale != NULL it also receives a
 * per-row scale for the block's MLP hidden activations, from a bound rather than their maximum (which no GEMM
 * epilogue can know): |gelu(ln(x) W1^T + b1)| <= ||ln(x)[r]||_2 * max_n ||W1[n]||_2 + max |b1|, so
 *   hid_scale[r] = (1.0625 * ||ln(x)[r]||_2 * w1_rownorm_max + b1_absmax) / 448   (1 when that is 0);
 * e4m3 is a floating-point format, so a bound a few binades above the true maximum costs no precision.
 * x is fp32 when x_f32 != 0, else bf16.  C % 8 == 0, C <= 2048. */
int rajni_layernorm_fp8(const void* x, long x_row_stride, const float* w, const float* b, void* y_q,
                        float* y_scale, float* hid_scale, float w1_rownorm_max, float b1_absmax,
                        int rows, int C, float eps, int x_f32, rajni_stream_t stream);

/* ---- linear layers with fused epilogues (a3, a9, a13, a15) ----
 * y[M,N] = epi(x[M,K] W[N,K]^T).  W must be allocated with its row count padded up to a multiple
 * of 256 (rows >= N are never read into results but must be readable); K % 64 == 0; lda/ldw/ldc/ldr
 * in elements, multiples of 8.  bias/gamma are fp32 [N] (NULL = 0 / 1).
 * RESID: resid row for output row m is  (m / r_np) * r_nsrc + r_idx[m]  when r_idx != NULL
 * (r_idx = keep_idx flattened [B*r_np]), else m. */
typedef struct {
  const void* x; long lda;
  const void* w; long ldw;
  const float* bias;
  const float* gamma;
  const void* resid; long ldr;
  const int32_t* r_idx; int r_np; int r_nsrc;
  void* y; long ldc;
  int M, N, K;
  int epilogue;
  int dtype;
  int stream_f32;  /* RESID only: resid and y are the fp32 residual stream (1) instead of `dtype` (0) */
  /* fp8 weights (BASELINE config 5): when non-NULL, `w` holds fp8 e4m3 (OCP "fn": no inf, max 448) bytes
   * [N(pad256),K], ldw in bytes and a multiple of 16, and w_scale[n] (fp32 [N]) is the dequantisation
   * scale of row n: y = epi(x (q*s)^T) with bf16 x and fp32 accumulation.  dtype must be RAJNI_BF16. */
  const float* w_scale;
  /* fp8 activations on the fp8 matrix pipe (v_mfma_f32_16x16x128_f8f6f4; requires w_scale): when x_scale != NULL,
   * `x` holds e4m3 bytes [M,K] (lda in bytes, % 16 == 0) and x_scale[m] (fp32 [M]) is the dequantisation scale of
   * row m - what rajni_layernorm_fp8 writes: y = epi((xq*xs) (wq*ws)^T), fp32 accumulation.  K % 256 == 0, K >= 512.
   * With epilogue RAJNI_EPI_BIAS_GELU, y_scale (fp32 [M], required) selects an e4m3 OUTPUT: y is [M,N] bytes
   * (ldc in bytes, % 16 == 0) holding e4m3_rne_sat(gelu(.) * (1 / y_scale[m])) - the next linear's x / x_scale. */
  const float* x_scale;
  const float* y_scale;
} rajni_linear_args;
int rajni_linear(const rajni_linear_args* args, rajni_stream_t stream);

/* ---- a12: patch-embed + CLS + pos-embed                                    model.py:34-37 ----
 * images [B,Cin,S,S] -> x [B, 1+(S/P)^2, C].  conv weight w [C(pad256), ceil64(Cin*P*P)] (k order c,ky,kx,
 * zero-padded columns), bias fp32 [C]; cls [C]; pos [(1 or 0)+(S/P)^2, C] (`pos_has_cls`=0 is timm
 * no_embed_class: SURVEY B3); x is written as fp32 when x_f32 != 0.  S % P == 0.  For a power-of-two P >= 8
 * with S % 8 == 0 and Cin*P*P % 64 == 0 the im2col is fused into the GEMM's tile loads and no workspace is
 * needed; any other patch size (14: ViT-L/14, ViT-H/14, DINOv2) materialises the zero-padded column matrix in
 * `workspace` (16-byte aligned, >= rajni_patch_embed_workspace_bytes(...), which is 0 for the fused case). */
int rajni_patch_embed(const void* images, const void* w, const float* bias, const void* cls,
                      const void* pos, int pos_has_cls, void* x, int x_f32, int B, int Cin, int S,
                      int P, int C, int dtype, void* workspace, size_t workspace_bytes, rajni_stream_t stream);
size_t rajni_patch_embed_workspace_bytes(int B, int Cin, int S, int P, int dtype);

/* ---- a11-a16: the whole RAJNIViTWrapper.forward                            model.py:30-69 ---- */
typedef struct {
  const float* norm1_w; const float* norm1_b;
  const void* qkv_w; const float* qkv_b;      /* [3C(pad),C], [3C] */
  const void* proj_w; const float* proj_b;    /* [C(pad),C], [C]  */
  const float* ls1;                           /* [C] or NULL      */
  const float* norm2_w; const float* norm2_b;
  const void* fc1_w; const float* fc1_b;      /* [Hd(pad),C], [Hd] */
  const void* fc2_w; const float* fc2_b;      /* [C(pad),Hd], [C]  */
  const float* ls2;
  /* schedule entry for this block (model.py:13-20): keep = 0 -> not scheduled (model.py:62) */
  int keep;                /* kept PATCH tokens (attention.py:31-32), output has keep+1 tokens */
  int update;              /* attention.py:25: recompute scores iff update or no carried scores */
  int32_t* keep_idx;       /* [B,keep+1] out (required when keep>0) */
  void* scores;            /* [B,N] out, dtype (optional) */
  void* next_scores;       /* [B,keep+1] out, dtype (required when keep>0: carried to the next block) */
  const int32_t* forced_keep_idx; /* test hook: use this selection instead (selection-conditional parity) */
  /* per-row scales of fp8 e4m3 weights (see rajni_linear_args.w_scale); NULL = that weight is `dtype` */
  const float* qkv_s; const float* proj_s; const float* fc1_s; const float* fc2_s;
  /* act_fp8 plans only: max_n ||W1deq[n,:]||_2 and max_n |b1[n]| of this block's fc1 (the hidden-activation
   * bound of rajni_layernorm_fp8) */
  float fc1_rownorm_max, fc1_bias_absmax;
  /* act_fp8 plans only: out_scale of rajni_attention_fp8 for this block (> 0: where the block's attention launch has
   * head dim 64 and at most 224 tokens, the attention output is emitted as e4m3 rows and proj runs on the fp8 matrix
   * pipe; 0: proj keeps bf16 activations x e4m3 weights) */
  float attn_out_scale;
} rajni_block;

typedef struct {
  int dtype;
  int B, in_chans, img_size, patch_size;
  int C, H, D, depth, hidden, num_classes;
  float ln_eps, attn_scale;
  int pos_has_cls;
  const void* patch_w; const float* patch_b; const void* cls_token; const void* pos_embed;
  const rajni_block* blocks;                   /* host array [depth] */
  const float* norm_w; const float* norm_b;
  const void* head_w; const float* head_b;     /* [classes(pad),C], [classes] */
  void* workspace; size_t workspace_bytes;     /* >= rajni_vit_workspace_bytes() */
  int32_t* token_counts;                       /* HOST int32[depth] out: tokens at block entry (model.py:43) */
  int logits_ld;                               /* row stride of `logits` in elements (0 = num_classes); % 8 == 0 */
  int cls_only_last_block;                     /* 1: when the last block is not a pruning stage, compute it for the
                                                  CLS row only - attention with the CLS query over all tokens,
                                                  proj / MLP on B rows.  model.py:65-66 feeds only x[:, 0] to the
                                                  head, so the logits are the same function; the other rows of the
                                                  last block are never formed.  0 (default): every row, like the
                                                  reference's op graph */
  int resid_bf16;                              /* 0 (default): the residual stream x is kept in fp32 between
                                                  blocks (2x closer to the fp32 reference than a bf16 stream, see
                                                  DESIGN.md); 1: keep it in bf16 like the reference's bf16 model */
  int act_fp8;                                 /* 1 (opt-in, needs e4m3 block weights): norm1 / norm2 emit per-row
                                                  scaled e4m3 activations, QKV / FC1 / FC2 run on the fp8 matrix
                                                  pipe, FC1's GELU epilogue re-quantises the hidden activations
                                                  (rajni_layernorm_fp8, rajni_linear_args.x_scale); blocks with attn_out_scale > 0
                                                  also emit the attention output as e4m3 rows and run proj on that pipe
                                                  (rajni_attention_fp8).  Attention's products, patch embed, head and the
                                                  residual stream are unchanged.
                                                  C % 256 == 0 and hidden % 256 == 0.  0 (default): bf16 activations */
} rajni_vit_plan;

size_t rajni_vit_workspace_bytes(const rajni_vit_plan* plan);
/* images [B,Cin,S,S] dtype; logits [B,num_classes] dtype */
int rajni_vit_forward(const rajni_vit_plan* plan, const void* images, void* logits,
                      rajni_stream_t stream);

/* ---- measurement hooks (bench.py roofline): HIP-event timing per kernel class on the launch
 * stream.  mask bit i enables class i; classes listed by rajni_profile_class_name(). ---- */
enum { RAJNI_NUM_KCLASS = 17 };
void rajni_profile_enable(unsigned mask);
const char* rajni_profile_class_name(int kclass);
/* synchronises the recorded events, ADDS them into the accumulators, returns them: per class the
 * number of launches, total milliseconds, algorithmic flops and algorithmic bytes. */
int rajni_profile_collect(long long* launches, double* ms, double* flops, double* bytes);
void rajni_profile_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* RAJNI_HIP_H */
