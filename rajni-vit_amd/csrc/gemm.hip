// bf16 "TN" GEMM with fused epilogues for the packed-token linears (SURVEY k2,k13,k17,k18,k20).
//
//   Y[m][n] = epi( sum_k X[m][k] * W[n][k] )        X = activations [M,K], W = torch Linear weight [N,K]
//
// gfx950 design (v1: 128x128x64 tile, 4 waves, 2 workgroups per CU):
//   * both operands are K-contiguous, so every MFMA fragment is one 16-byte LDS read;
//   * tiles are staged HBM->LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction),
//     double buffered, one barrier per 64-deep K step;
//   * the LDS image is XOR-swizzled on the SOURCE address (LDS-DMA writes lane-linear), the same
//     involution is applied on the fragment reads -> conflict-free ds_read_b128;
//   * v_mfma_f32_16x16x32_bf16 with W as the A operand and X as the B operand, so that a lane owns
//     ONE output row m and - through a permutation of which W row feeds which fragment row -
//     16 CONSECUTIVE output columns: epilogue loads/stores are 16-byte and row-contiguous, and
//     per-row epilogue data (gathered residual row, patch->token row remap) is per-lane constant;
//   * workgroup -> tile mapping walks N fastest inside an XCD-contiguous chunk, so the X panel of a
//     tile row is fetched from HBM once per XCD and W stays L2-resident.
#include "common.h"

namespace {

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3 };
enum { ALOAD_PLAIN = 0, ALOAD_PATCH = 1 };

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // X + W
constexpr int GEMM_LDS = 2 * STAGE_BYTES;      // double buffered: 64 KiB

struct GemmParams {
  const bf16_t* X; long lda;
  const bf16_t* W; long ldw;
  const float* bias;
  const float* gamma;
  const void* R; long ldr;      // residual stream rows (bf16 or fp32, see SF32)
  const int* ridx; int r_np, r_nsrc;
  void* Y; long ldc;            // output (fp32 when SF32: RESID / PATCH write the residual stream)
  int M, N, K;
  int tiles_n, total_tiles;
  // patch-embed A loader / epilogue
  int cin, S, log2ps, gw, npatch;
  const bf16_t* pos; int pos_off;
};

// swizzle keys (3 bits) of a tile row; must give 16 distinct LDS slots to the 16 rows one
// ds_read_b128 lane group touches (row stride 128 B, bank row 256 B -> slot = (row&1)*8 + chunk^key)
__device__ __forceinline__ int key_x(int row) { return (row >> 1) & 7; }
// W fragment rows are read permuted: {16a + 4*ni + b : a,b in 0..3}
__device__ __forceinline__ int key_w(int row) { return ((row >> 4) & 3) * 2 + ((row >> 1) & 1); }

__device__ __forceinline__ float gelu_erf(float x) {
  // exact-erf GELU (timm nn.GELU()); erf by Abramowitz-Stegun 7.1.26, |err| < 1.5e-7
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = 1.0f - poly * __expf(-z * z);
  const float erfv = copysignf(e, x);
  return 0.5f * x * (1.0f + erfv);
}

// load / store 16 consecutive stream elements (bf16 or fp32) as floats
template <bool F32>
__device__ __forceinline__ void load16(const void* base, long off, float* f) {
  if (F32) {
    const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 t = q[i]; f[4*i] = t.x; f[4*i+1] = t.y; f[4*i+2] = t.z; f[4*i+3] = t.w; }
  } else {
    const bf16_t* q = reinterpret_cast<const bf16_t*>(base) + off;
    unpack8(*reinterpret_cast<const uint4*>(q), f);
    unpack8(*reinterpret_cast<const uint4*>(q + 8), f + 8);
  }
}
template <bool F32>
__device__ __forceinline__ float load1(const void* base, long off) {
  return F32 ? reinterpret_cast<const float*>(base)[off] : bf2f(reinterpret_cast<const bf16_t*>(base)[off]);
}
template <bool F32>
__device__ __forceinline__ void store16(void* base, long off, const float* v) {
  if (F32) {
    float4* q = reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = make_float4(v[4*i], v[4*i+1], v[4*i+2], v[4*i+3]);
  } else {
    bf16_t* q = reinterpret_cast<bf16_t*>(base) + off;
    *reinterpret_cast<uint4*>(q) = pack8(v);
    *reinterpret_cast<uint4*>(q + 8) = pack8(v + 8);
  }
}
template <bool F32>
__device__ __forceinline__ void store1(void* base, long off, float v) {
  if (F32) reinterpret_cast<float*>(base)[off] = v;
  else reinterpret_cast<bf16_t*>(base)[off] = f2bf(v);
}

// SF32: the residual-stream tensors this launch touches (R and Y of RESID, Y of PATCH) are fp32
template <int EPI, int ALOAD, bool SF32>
__global__ void __launch_bounds__(256, 2) gemm_bf16_tn(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

  // ---- XCD-aware tile id: blocks b and b+8 share an XCD; give each XCD a contiguous tile range
  const int total = p.total_tiles;
  const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging addresses: wave w stages pieces 4w..4w+3 of each operand; a piece = 8 rows x 128 B
  const int r_in = lane >> 3, pch = lane & 7;
  const bf16_t* xsrc[4];
  const bf16_t* wsrc[4];
  int xk[4];  // patch loader: this lane's k offset inside a K step (elements)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + r_in;
    const int cx = pch ^ key_x(row);
    const int cw = pch ^ key_w(row);
    int m = m0 + row;
    if (m > p.M - 1) m = p.M - 1;  // clamp: duplicates are computed but never stored
    if (ALOAD == ALOAD_PLAIN) {
      xsrc[i] = p.X + (long)m * p.lda + cx * 8;
      xk[i] = 0;
    } else {
      const int b = m / p.npatch, pp = m - b * p.npatch;
      const int py = pp / p.gw, px = pp - py * p.gw;
      xsrc[i] = p.X + ((long)b * p.cin * p.S + (py << p.log2ps)) * p.S + (px << p.log2ps);
      xk[i] = cx * 8;
    }
    wsrc[i] = p.W + (long)(n0 + row) * p.ldw + cw * 8;  // W rows are padded to a multiple of 128
  }

  auto stage = [&](int kt, int buf) {
    char* sx = smem + buf * STAGE_BYTES;
    char* sw = sx + TILE_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* src;
      if (ALOAD == ALOAD_PLAIN) {
        src = xsrc[i] + k0;
      } else {
        const int k = k0 + xk[i];
        const int ps2 = 2 * p.log2ps;
        const int ch = k >> ps2, rem = k & ((1 << ps2) - 1);
        const int ky = rem >> p.log2ps, kx = rem & ((1 << p.log2ps) - 1);
        src = xsrc[i] + ((long)ch * p.S + ky) * p.S + kx;
      }
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sx + (wave * 4 + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(wsrc[i] + k0), LDS_PTR(sw + (wave * 4 + i) * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets (bytes inside a tile), K-step invariant
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  int xoff[4], xkey[4], woff[4], wkey[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int xr = wm * 64 + i * 16 + l15;
    xoff[i] = xr * 128; xkey[i] = key_x(xr);
    const int wr = wn * 64 + 16 * (l15 >> 2) + i * 4 + (l15 & 3);
    woff[i] = wr * 128; wkey[i] = key_w(wr);
  }

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt has landed for every wave; everyone is done reading the other buffer
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* sx = smem + (kt & 1) * STAGE_BYTES;
    const char* sw = sx + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + g;
      bf16x8 xf[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xf[i] = *reinterpret_cast<const bf16x8*>(sx + xoff[i] + ((c ^ xkey[i]) << 4));
        wf[i] = *reinterpret_cast<const bf16x8*>(sw + woff[i] + ((c ^ wkey[i]) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns row m and columns nb .. nb+15
  const int nb = n0 + wn * 64 + 16 * g;
  float bias[16], gam[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int n = nb + j;
    bias[j] = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
    gam[j] = (EPI == EPI_RESID && p.gamma != nullptr && n < p.N) ? p.gamma[n] : 1.f;
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + l15;
    if (m >= p.M) continue;
    float v[16];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) v[ni * 4 + rg] = acc[ni][mi][rg] + bias[ni * 4 + rg];

    long orow = m;
    if (EPI == EPI_GELU) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = gelu_erf(v[j]);
    } else if (EPI == EPI_RESID) {
      long rrow = m;
      if (p.ridx != nullptr) {
        const int b = m / p.r_np;
        rrow = (long)b * p.r_nsrc + p.ridx[m];
      }
      const long roff = rrow * p.ldr + nb;
      if (nb + 16 <= p.N) {
        float rf[16];
        load16<SF32>(p.R, roff, rf);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = fmaf(gam[j], v[j], rf[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (nb + j < p.N) v[j] = fmaf(gam[j], v[j], load1<SF32>(p.R, roff + j));
      }
    } else if (EPI == EPI_PATCH) {
      const int b = m / p.npatch, pp = m - b * p.npatch;
      orow = (long)b * (p.npatch + 1) + 1 + pp;
      const bf16_t* pr = p.pos + (long)(pp + p.pos_off) * p.ldc + nb;
      if (nb + 16 <= p.N) {
        float pf[16];
        unpack8(*reinterpret_cast<const uint4*>(pr), pf);
        unpack8(*reinterpret_cast<const uint4*>(pr + 8), pf + 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] += pf[j];
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (nb + j < p.N) v[j] += bf2f(pr[j]);
      }
    }
    const long yoff = orow * p.ldc + nb;
    if (nb + 16 <= p.N) {
      store16<SF32>(p.Y, yoff, v);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (nb + j < p.N) store1<SF32>(p.Y, yoff + j, v[j]);
    }
  }
}

// x[b,0,:] = cls + pos[0]  (or cls alone when pos has no CLS row)
template <bool SF32>
__global__ void cls_pos_kernel(const bf16_t* cls, const bf16_t* pos, int pos_has_cls, void* x,
                               long img_stride, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float v = bf2f(cls[c]);
  if (pos_has_cls) v += bf2f(pos[c]);
  store1<SF32>(x, (long)b * img_stride + c, v);
}

template <int EPI, int ALOAD, bool SF32>
int launch_gemm(const GemmParams& p, int kclass, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_tn<EPI, ALOAD, SF32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    if (e != hipSuccess) {
      rajni_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
      return RAJNI_ERR_LAUNCH;
    }
    attr_set = true;
  }
  ProfScope prof(kclass, s, 2.0 * p.M * (double)p.N * p.K,
                 2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N));
  hipLaunchKernelGGL((gemm_bf16_tn<EPI, ALOAD, SF32>), dim3(p.total_tiles), dim3(256), GEMM_LDS, s, p);
  RAJNI_CHECK_LAUNCH("gemm_bf16_tn");
  return RAJNI_OK;
}

}  // namespace

int launch_linear(const rajni_linear_args& a, hipStream_t s) {
  RAJNI_REQUIRE(a.dtype == RAJNI_BF16, RAJNI_ERR_UNSUPPORTED, "rajni_linear: only bf16 is built");
  RAJNI_REQUIRE(a.x && a.w && a.y, RAJNI_ERR_INVALID, "rajni_linear: null pointer");
  RAJNI_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % BK == 0, RAJNI_ERR_INVALID,
                "rajni_linear: M,N>0 and K %% 64 == 0 required (M=%d N=%d K=%d)", a.M, a.N, a.K);
  RAJNI_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 8 == 0, RAJNI_ERR_INVALID,
                "rajni_linear: leading dimensions must be multiples of 8 elements");
  RAJNI_REQUIRE(((uintptr_t)a.x | (uintptr_t)a.w | (uintptr_t)a.y | (uintptr_t)a.resid) % 16 == 0,
                RAJNI_ERR_INVALID, "rajni_linear: pointers must be 16-byte aligned");
  GemmParams p{};
  p.X = (const bf16_t*)a.x; p.lda = a.lda;
  p.W = (const bf16_t*)a.w; p.ldw = a.ldw;
  p.bias = a.bias; p.gamma = a.gamma;
  p.R = a.resid; p.ldr = a.ldr;
  p.ridx = a.r_idx; p.r_np = a.r_np > 0 ? a.r_np : 1; p.r_nsrc = a.r_nsrc;
  p.Y = a.y; p.ldc = a.ldc;
  p.M = a.M; p.N = a.N; p.K = a.K;
  p.tiles_n = (a.N + BN - 1) / BN;
  p.total_tiles = p.tiles_n * ((a.M + BM - 1) / BM);
  switch (a.epilogue) {
    case RAJNI_EPI_BIAS: return launch_gemm<EPI_BIAS, ALOAD_PLAIN, false>(p, KC_GEMM_BIAS, s);
    case RAJNI_EPI_BIAS_GELU: return launch_gemm<EPI_GELU, ALOAD_PLAIN, false>(p, KC_GEMM_GELU, s);
    case RAJNI_EPI_BIAS_RESID:
      RAJNI_REQUIRE(a.resid != nullptr && a.ldr % 8 == 0, RAJNI_ERR_INVALID,
                    "rajni_linear: RESID epilogue needs resid and ldr %% 8 == 0");
      return a.stream_f32 ? launch_gemm<EPI_RESID, ALOAD_PLAIN, true>(p, KC_GEMM_RESID, s)
                          : launch_gemm<EPI_RESID, ALOAD_PLAIN, false>(p, KC_GEMM_RESID, s);
    default:
      rajni_set_error("rajni_linear: unknown epilogue %d", a.epilogue);
      return RAJNI_ERR_INVALID;
  }
}

int launch_patch_embed(const void* images, const void* w, const float* bias, const void* cls,
                       const void* pos, int pos_has_cls, void* x, int out_f32, int B, int Cin, int S,
                       int P, int C, hipStream_t s) {
  RAJNI_REQUIRE(images && w && cls && pos && x, RAJNI_ERR_INVALID, "rajni_patch_embed: null pointer");
  RAJNI_REQUIRE(P >= 8 && (P & (P - 1)) == 0 && S % P == 0 && S % 8 == 0, RAJNI_ERR_UNSUPPORTED,
                "rajni_patch_embed: patch size must be a power of two >= 8 dividing the image (P=%d S=%d)", P, S);
  const int K = Cin * P * P;
  RAJNI_REQUIRE(K % BK == 0 && C % 8 == 0, RAJNI_ERR_UNSUPPORTED,
                "rajni_patch_embed: Cin*P*P %% 64 == 0 and C %% 8 == 0 required");
  int log2ps = 0;
  while ((1 << log2ps) < P) ++log2ps;
  const int gw = S / P, npatch = gw * gw;
  GemmParams p{};
  p.X = (const bf16_t*)images; p.lda = 0;
  p.W = (const bf16_t*)w; p.ldw = K;
  p.bias = bias;
  p.Y = x; p.ldc = C;
  p.M = B * npatch; p.N = C; p.K = K;
  p.tiles_n = (C + BN - 1) / BN;
  p.total_tiles = p.tiles_n * ((p.M + BM - 1) / BM);
  p.cin = Cin; p.S = S; p.log2ps = log2ps; p.gw = gw; p.npatch = npatch;
  p.pos = (const bf16_t*)pos; p.pos_off = pos_has_cls ? 1 : 0;
  int rc = out_f32 ? launch_gemm<EPI_PATCH, ALOAD_PATCH, true>(p, KC_GEMM_PATCH, s)
                   : launch_gemm<EPI_PATCH, ALOAD_PATCH, false>(p, KC_GEMM_PATCH, s);
  if (rc != RAJNI_OK) return rc;
  {
    ProfScope prof(KC_CLS_POS, s, 0.0, 6.0 * B * C);
    const int n = B * C;
    if (out_f32)
      hipLaunchKernelGGL(cls_pos_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, s, (const bf16_t*)cls,
                         (const bf16_t*)pos, pos_has_cls, x, (long)(npatch + 1) * C, B, C);
    else
      hipLaunchKernelGGL(cls_pos_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, s, (const bf16_t*)cls,
                         (const bf16_t*)pos, pos_has_cls, x, (long)(npatch + 1) * C, B, C);
    RAJNI_CHECK_LAUNCH("cls_pos_kernel");
  }
  return RAJNI_OK;
}
