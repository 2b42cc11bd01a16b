// bf16 "TN" GEMM with fused epilogues for the packed-token linears (SURVEY k2,k13,k17,k18,k20).
//
//   Y[m][n] = epi( sum_k X[m][k] * W[n][k] )        X = activations [M,K], W = torch Linear weight [N,K]
//
// Common gfx950 design:
//   * both operands are K-contiguous, so every MFMA fragment is one 16-byte LDS read;
//   * tiles are staged HBM->LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction);
//     the LDS image is XOR-swizzled on the SOURCE address (LDS-DMA writes lane-linear) and the same
//     involution is applied on the fragment reads -> conflict-free ds_read_b128;
//   * v_mfma_f32_16x16x32_bf16 with W as the A operand and X as the B operand, so that a lane owns
//     ONE output row m and - through a permutation of which W row feeds which fragment row - a
//     chosen set of output columns (MAP_SEC / MAP_NAT below): epilogue loads/stores are 16-byte and
//     cover whole 64-byte sectors per row, and per-row epilogue data (gathered residual row,
//     patch->token row remap) is per-lane constant;
//   * workgroup -> tile mapping: every XCD gets a contiguous range of tile ids; ids run (N block of
//     ~1.5 MiB of W, row tile, column in block), see tile_mn.
//
// Tilings (chosen per launch, see launch_gemm):
//   wide  256 x 256 x 64, 8 waves, 2 LDS stages, persistent stream  - wide outputs (qkv, fc1)
//   mid   256 x 128 x 64, 8 waves, 3 LDS stages, persistent stream  - N = 768-class outputs (proj, fc2)
//   small 128 x 128 x 64, 4 waves, 2 stages, 2 workgroups per CU    - M < 1024 (head, tiny batches)
//   f32   128 x 128 x 32(fp32) on v_mfma_f32_16x16x4_f32            - fp32 models
// (Measured in round 1 and removed from this file, numbers in DESIGN.md section 4: 256x128x32 / 128x128x32
//  tilings with 64-byte rows, the wide tile on four waves of 128x128, per-SIMD-partner issue patterns,
//  front-loaded fragment reads / DMA, non-temporal epilogue accesses, cache-policy bits on the DMA loads, row
//  super-blocks in the tile order, round-balanced grids.)
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3,
       EPI_GELU8 = 4 };   // fp8 x fp8 kernel only (gemm_f8.h): bias + GELU, output re-quantised to e4m3 with a per-row scale
enum { ALOAD_PLAIN = 0, ALOAD_PATCH = 1 };

struct GemmParams {
  const void* X; long lda;      // activations (bf16, or fp32 on the fp32 model path)
  const void* W; long ldw;      // weights, same element type (or fp8 e4m3 bytes, see wscale)
  const float* wscale;          // W8 kernels: W is fp8 e4m3 [N,K] and wscale[n] its per-row dequantisation scale
  const float* xscale;          // fp8 x fp8 kernel: X is fp8 e4m3 [M,K] and xscale[m] its per-row dequantisation scale
  const float* yscale;          // ... EPI_GELU8: output row m is stored as e4m3(gelu(.) / yscale[m])
  const float* bias;
  const float* gamma;
  const void* R; long ldr;      // residual stream rows (bf16 or fp32, see SF32)
  const int* ridx; int r_np, r_nsrc;
  void* Y; long ldc;            // output (fp32 when SF32: RESID / PATCH write the residual stream)
  int M, N, K;
  int tiles_n, total_tiles;
  int nblk;                     // persistent tilings: column tiles per N block of the tile order (tile_mn)
  // patch-embed A loader / epilogue
  int cin, S, log2ps, gw, npatch;
  const void* pos; int pos_off; // pos-embed rows, activation dtype
  unsigned long long* stamps;   // diagnostic builds only (RAJNI_GEMM_STAMPS): 4 s_memtime values per block
  int stagger;                  // residual launches: every other workgroup of an XCD sleeps stagger x 8k cycles before its
                                // first tile (see g_resid_stagger); 0 = off
};

// exact-erf GELU (timm nn.GELU()) for the bf16 path, two values per instruction (v_pk_*_f32, no
// transcendentals):  gelu(x) = x*(0.5 + h(x)),  h(x) = 0.5*erf(x/sqrt2) ~= xc*P(xc^2),
// xc = clamp(x, +-3*sqrt2), P = degree-8 least-squares fit on Chebyshev nodes (coefficients from
// tools/fit_gelu.py).  |gelu error| <= 4.3e-5 absolute - 1/50 of a bf16 ulp at |y| ~ 1 - measured
// against fp64 erf over [-8, 8].  The FC1 epilogue is VALU bound: the previous exp+rcp form
// (Abramowitz-Stegun 7.1.26) cost ~20k cycles per 256x256 tile, a third of the tile's time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_pk(f32x2 x) {
  const float X0 = 4.24264069f;
  f32x2 xc = __builtin_elementwise_min(__builtin_elementwise_max(x, f32x2{-X0, -X0}), f32x2{X0, X0});
  const f32x2 u = xc * xc;
  f32x2 q = f32x2{5.405088552e-11f, 5.405088552e-11f};
  q = q * u + f32x2{-5.202485173e-09f, -5.202485173e-09f};
  q = q * u + f32x2{2.215015442e-07f, 2.215015442e-07f};
  q = q * u + f32x2{-5.557312053e-06f, -5.557312053e-06f};
  q = q * u + f32x2{9.274613401e-05f, 9.274613401e-05f};
  q = q * u + f32x2{-1.104852507e-03f, -1.104852507e-03f};
  q = q * u + f32x2{9.805144109e-03f, 9.805144109e-03f};
  q = q * u + f32x2{-6.633033261e-02f, -6.633033261e-02f};
  q = q * u + f32x2{3.988969665e-01f, 3.988969665e-01f};
  const f32x2 h = xc * q;
  return x * h + x * f32x2{0.5f, 0.5f};
}
// the same for two pairs at once, the two Horner chains interleaved statement by statement: a packed fp32 op that reads the
// result of the one issued right before it costs a wait state (hipcc pads with s_nop 0: the single-chain form had one
// behind almost every v_pk_fma_f32 - ~600 per tile and wave, a quarter on top of the epilogue's VALU time); two
// independent chains fill each other's slots.  Same operations per element: bit-identical results.
__device__ __forceinline__ void gelu_pk4(f32x2& a, f32x2& b) {
  const float X0 = 4.24264069f;
  const f32x2 lo = f32x2{-X0, -X0}, hi = f32x2{X0, X0};
  const f32x2 ac = __builtin_elementwise_min(__builtin_elementwise_max(a, lo), hi);
  const f32x2 bc = __builtin_elementwise_min(__builtin_elementwise_max(b, lo), hi);
  const f32x2 ua = ac * ac, ub = bc * bc;
  f32x2 qa = f32x2{5.405088552e-11f, 5.405088552e-11f}, qb = qa;
#define RAJNI_GELU_STEP(c) { const f32x2 k = f32x2{c, c}; qa = qa * ua + k; qb = qb * ub + k; }
  RAJNI_GELU_STEP(-5.202485173e-09f)
  RAJNI_GELU_STEP(2.215015442e-07f)
  RAJNI_GELU_STEP(-5.557312053e-06f)
  RAJNI_GELU_STEP(9.274613401e-05f)
  RAJNI_GELU_STEP(-1.104852507e-03f)
  RAJNI_GELU_STEP(9.805144109e-03f)
  RAJNI_GELU_STEP(-6.633033261e-02f)
  RAJNI_GELU_STEP(3.988969665e-01f)
#undef RAJNI_GELU_STEP
  const f32x2 ha = ac * qa, hb = bc * qb;
  const f32x2 half = f32x2{0.5f, 0.5f};
  const f32x2 ya = a * ha + a * half, yb = b * hb + b * half;
  a = ya; b = yb;
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
__device__ __forceinline__ void load8s(const void* base, long off, float* f) {   // 8 stream elements
  if (F32) load8<float>(reinterpret_cast<const float*>(base) + off, f);
  else load8<bf16_t>(reinterpret_cast<const bf16_t*>(base) + off, f);
}
template <bool F32>
__device__ __forceinline__ void store8s(void* base, long off, const float* v) {
  if (F32) store8<float>(reinterpret_cast<float*>(base) + off, v);
  else store8<bf16_t>(reinterpret_cast<bf16_t*>(base) + off, v);
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

// FC1's hidden activations (the largest tensor of a block: written once, read once by FC2) are stored NON-TEMPORAL:
// measured on the whole forward (tools/ab_libs.sh, round 3) FC1 -2..-3 %, +0.5 % end to end.  The rest of that experiment -
// keeping the fp32 residual stream resident in the 256 MiB Infinity Cache by also moving the QKV output and FC2's X-operand
// DMA past it - was negative: non-temporal X loads cost FC2 +15 % (its row panels are re-read by six column tiles out of L2),
// non-temporal QKV stores cost attention +4 %; neither made the residual epilogues or LayerNorm faster.
#ifndef RAJNI_FC1_NT
#define RAJNI_FC1_NT 1
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void store_u4(void* ptr, const uint4& v) {
  if constexpr (NT) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(ptr));
  else *reinterpret_cast<uint4*>(ptr) = v;
}

// One output row m, columns nb..nb+15 (v = accumulators + bias on entry).
// SF32: the residual-stream tensors this launch touches (R and Y of RESID, Y of PATCH) are fp32.
// v[0..7] are columns nbA..nbA+7 and v[8..15] columns nbB..nbB+7 of output row m (nbB = nbA + 32: the
// sector mapping MAP_SEC below).
template <int EPI, bool SF32>
__device__ __forceinline__ void epilogue_row(const GemmParams& p, int m, int nbA, int nbB, float* v, const float* gam) {
  long orow = m;
  const bool full = nbB + 8 <= p.N;   // both halves inside N (nbA < nbB)
  if (EPI == EPI_GELU) {
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
      f32x2 a = f32x2{v[j], v[j + 1]}, b = f32x2{v[j + 2], v[j + 3]};
      gelu_pk4(a, b);
      v[j] = a[0]; v[j + 1] = a[1]; v[j + 2] = b[0]; v[j + 3] = b[1];
    }
  } else if (EPI == EPI_RESID) {
    long rrow = m;
    if (p.ridx != nullptr) {
      const int b = m / p.r_np;
      rrow = (long)b * p.r_nsrc + p.ridx[m];
    }
    const long rbase = rrow * p.ldr;
    if (full) {
      float rf[16];
      load8s<SF32>(p.R, rbase + nbA, rf);
      load8s<SF32>(p.R, rbase + nbB, rf + 8);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = fmaf(gam[j], v[j], rf[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int n = (j < 8 ? nbA : nbB - 8) + j;
        if (n < p.N) v[j] = fmaf(gam[j], v[j], load1<SF32>(p.R, rbase + n));
      }
    }
  } else if (EPI == EPI_PATCH) {
    const int b = m / p.npatch, pp = m - b * p.npatch;
    orow = (long)b * (p.npatch + 1) + 1 + pp;
    const bf16_t* pr = reinterpret_cast<const bf16_t*>(p.pos) + (long)(pp + p.pos_off) * p.ldc;
    if (full) {
      float pf[16];
      unpack8(*reinterpret_cast<const uint4*>(pr + nbA), pf);
      unpack8(*reinterpret_cast<const uint4*>(pr + nbB), pf + 8);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] += pf[j];
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int n = (j < 8 ? nbA : nbB - 8) + j;
        if (n < p.N) v[j] += bf2f(pr[n]);
      }
    }
  }
  constexpr bool OUT32 = SF32 && (EPI == EPI_RESID || EPI == EPI_PATCH);
  const long ybase = orow * p.ldc;
  if (full) {
    store8s<OUT32>(p.Y, ybase + nbA, v);
    store8s<OUT32>(p.Y, ybase + nbB, v + 8);
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = (j < 8 ? nbA : nbB - 8) + j;
      if (n < p.N) store1<OUT32>(p.Y, ybase + n, v[j]);
    }
  }
}

// ---- fp32 residual stream epilogues (RESID / PATCH with SF32) use the NATURAL column order --------
// With the permuted order a lane owns 16 consecutive columns = 64 bytes of fp32, so each of its four
// 16-byte accesses lands in a different 64-byte sector from its three row neighbours: one wave
// instruction touches 64 sectors.  Measured (tools/proj_probe.py): the fp32-stream proj GEMM took 151 us
// against 72 us with a bias-only bf16 epilogue.  In the natural order (W fragment row r of n-tile ni =
// column 16*ni + r) a lane owns columns 16*ni + 4*g + {0..3}: the four lanes of a row cover one whole
// 64-byte sector per access - 16 sectors per wave instruction, 4x fewer memory transactions.
constexpr bool nat_order(int epi, bool sf32) { return sf32 && (epi == EPI_RESID || epi == EPI_PATCH); }

// Which W tile row feeds fragment row r of n-tile ni decides which output columns a lane owns
// (accumulator element j = ni*4 + rg of a lane in lane group g; n0w = first column of the wave):
//   MAP_NAT  col = 16*ni + 4*g + rg          fp32 stream: 4 lanes of a row = one 64-byte sector (above)
//   MAP_SEC  col = 32*(ni>>1) + 8*g + 4*(ni&1) + rg   bf16 outputs: a lane owns 2 x 8 consecutive columns
//            and the 4 lanes of a row write one whole 64-byte sector per 16-byte store instruction.
//            (With 16 consecutive columns per lane every store instruction wrote 4 quarter sectors
//            per row: the wide tiling's epilogue took 9.5k cycles per 256x256 tile, store-issue bound.)
//   MAP_F8   col = 16*g + 4*ni + rg           fp8 outputs: a lane owns 16 consecutive columns = one 16-byte store,
//            the 4 lanes of a row one whole 64-byte sector
enum { MAP_NAT = 1, MAP_SEC = 2, MAP_F8 = 3 };
constexpr int col_map(int epi, bool sf32) { return epi == EPI_GELU8 ? MAP_F8 : nat_order(epi, sf32) ? MAP_NAT : MAP_SEC; }

template <int MAP>
__device__ __forceinline__ int out_col(int n0w, int g, int j) {
  const int ni = j >> 2, rg = j & 3;
  return MAP == MAP_NAT ? n0w + 16 * ni + 4 * g + rg
       : MAP == MAP_F8 ? n0w + 16 * g + 4 * ni + rg : n0w + 32 * (ni >> 1) + 8 * g + 4 * (ni & 1) + rg;
}
// W tile row (relative to the wave's first W row) read by fragment row l15 of n-tile ni
template <int MAP>
__device__ __forceinline__ int w_frag_row(int l15, int ni) {
  return MAP == MAP_NAT ? ni * 16 + l15
       : MAP == MAP_F8 ? 16 * (l15 >> 2) + 4 * ni + (l15 & 3) : 32 * (ni >> 1) + 8 * (l15 >> 2) + 4 * (ni & 1) + (l15 & 3);
}
// swizzle key of a W tile row for 128-byte-row tilings: the 16 rows one ds_read_b128 lane group
// touches must land in 16 distinct 16-byte slots of the 256-byte bank row
template <int MAP>
__device__ __forceinline__ int w_key(int row) {
  return MAP == MAP_NAT ? (row >> 1) & 7
       : MAP == MAP_F8 ? ((row >> 4) & 3) * 2 + ((row >> 1) & 1) : ((row >> 3) & 3) * 2 + ((row >> 1) & 1);
}

// ---- fp8 (e4m3) weights: 64-byte tile rows (BK = 64 one-byte elements) -----------------------------
// A W fragment of v_mfma_f32_16x16x32_bf16 is 8 consecutive k of one row = 8 BYTES here (ds_read_b64),
// converted to bf16 in registers (v_cvt_scalef32_pk_bf16_fp8, exact); the per-row scale multiplies the
// fp32 accumulator in the epilogue, so the arithmetic is x . (q * s) - a bf16 activation times the
// DEQUANTISED weight - with fp32 accumulation.  LDS-DMA still moves 16-byte units, so the swizzle
// works on the four 16-byte units of a row: a ds_read_b64 lane group (32 lanes: 16 rows x 2 halves)
// is conflict free when the 4 rows of every (row % 4) class sit in 4 distinct units.
template <int MAP>
__device__ __forceinline__ int w_key8(int row) {
  return MAP == MAP_NAT ? (row >> 2) & 3 : (row >> 3) & 3;
}
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(uint2 raw) {
  const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, false);
  const bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, true);
  const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, false);
  const bf16x2 d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, true);
  return bf16x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}
// The 8 fp8 bytes are read from LDS through a __bf16 vector type ON PURPOSE: hipcc's waitcnt insertion
// puts s_waitcnt vmcnt(0) in front of every LDS read that type-based alias analysis cannot separate
// from the LDS-DMA writes - with a uint2 read each W fragment read drained the whole DMA queue (the
// mid tiling ran at 390 instead of 840 TFLOP/s).  Reads typed like the bf16 fragments are left alone.
typedef __attribute__((ext_vector_type(4))) __bf16 fp8x8_raw;
template <bool W8> struct WFragT { typedef bf16x8 type; };
template <> struct WFragT<true> { typedef fp8x8_raw type; };
__device__ __forceinline__ bf16x8 w_frag_bf16(const bf16x8& f) { return f; }
__device__ __forceinline__ bf16x8 w_frag_bf16(const fp8x8_raw& f) { return fp8x8_to_bf16(__builtin_bit_cast(uint2, f)); }

// natural-order epilogue of one output row (fp32 stream): v = accumulators + bias on entry
template <int EPI>
__device__ __forceinline__ void epilogue_row_nat(const GemmParams& p, int m, int n0w, int g, float* v, const float* gam) {
  long orow = m, rrow = m;
  int pp = 0;
  if (EPI == EPI_RESID && p.ridx != nullptr) {
    const int b = m / p.r_np;
    rrow = (long)b * p.r_nsrc + p.ridx[m];
  }
  if (EPI == EPI_PATCH) {
    const int b = m / p.npatch;
    pp = m - b * p.npatch;
    orow = (long)b * (p.npatch + 1) + 1 + pp;
  }
  const float* R = reinterpret_cast<const float*>(p.R);
  const bf16_t* P = reinterpret_cast<const bf16_t*>(p.pos);
  float* Y = reinterpret_cast<float*>(p.Y);
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int c = n0w + 16 * ni + 4 * g;
    float* vv = v + 4 * ni;
    if (c + 4 <= p.N) {
      if (EPI == EPI_RESID) {
        const float4 r = *reinterpret_cast<const float4*>(R + rrow * p.ldr + c);
        vv[0] = fmaf(gam[4 * ni + 0], vv[0], r.x); vv[1] = fmaf(gam[4 * ni + 1], vv[1], r.y);
        vv[2] = fmaf(gam[4 * ni + 2], vv[2], r.z); vv[3] = fmaf(gam[4 * ni + 3], vv[3], r.w);
      } else {
        const uint2 q = *reinterpret_cast<const uint2*>(P + (long)(pp + p.pos_off) * p.ldc + c);
        vv[0] += bf_lo(q.x); vv[1] += bf_hi(q.x); vv[2] += bf_lo(q.y); vv[3] += bf_hi(q.y);
      }
      *reinterpret_cast<float4*>(Y + orow * p.ldc + c) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < p.N) {
          float o = vv[e];
          if (EPI == EPI_RESID) o = fmaf(gam[4 * ni + e], o, R[rrow * p.ldr + c + e]);
          else o += bf2f(P[(long)(pp + p.pos_off) * p.ldc + c + e]);
          Y[orow * p.ldc + c + e] = o;
          vv[e] = o;
        }
    }
  }
}

// Residual rows of a tile, loaded EARLY (during the last K step) for the fp32-stream RESID epilogue:
// that epilogue is a latency-bound HBM stream with only 8 waves per CU, so what sets its rate is how
// many loads are in flight and how much of their latency hides under MFMAs.  Measured on proj
// (tools/proj_probe.py): 155 us -> 135 us with all loads of the tile issued before the first store,
// -> (see profiles) with the loads issued one K step before the epilogue.
// (default cache policy on purpose: non-temporal loads / stores here cost 1-2 % of the forward - the LayerNorm
// launch that reads the stream next profits from the lines the default policy leaves in L2)
template <int MI>
struct ResidPrefetch {
  float4 r[MI][4];
  bool valid;
};
// Round 3 - the ACCESS SHAPE of this epilogue was its bottleneck (tools/pull_probe.hip: 128 KB of loads then 128 KB of stores
// per workgroup, every CU at once / 16 CUs only, cycles per tile): the accumulator layout's shape - a wave instruction = 16 rows
// x 64 bytes, four lanes per row - 14.7 k / 10.8 k; full lines - 8 rows x 128 bytes per instruction - 11.1 k / 3.6 k.  One CU
// pulls 64-byte pieces at a third of the rate of whole 128-byte lines, whatever the other CUs do.  ROWMAJOR (the persistent
// 256 x 128 tiling, which has 16 KiB of LDS to spare): residual rows are loaded, and outputs stored, 8 rows x 128 bytes at a
// time - row group mi, column half hc, instruction j: row 16 mi + 8 j + (lane >> 3), columns 32 hc + 4 (lane & 7) .. + 3 - and
// the accumulators reach that layout through a 2 KiB per-wave LDS transpose (epilogue_tile).
template <int MI>
__device__ __forceinline__ void prefetch_resid_rowmajor(const GemmParams& p, ResidPrefetch<MI>& pre, int m_base, int n0w, int lane) {
  pre.valid = false;
  if (n0w + 64 <= p.N && m_base + MI * 16 <= p.M) {     // interior tile: no guards needed
    // (opaque to loop-invariant code motion: hoisted out of the persistent tile loop, the lane's 64-bit base pointers of
    //  R and Y were two more registers live across the K loop - the fp8-weight instantiation spilled them)
    asm volatile("" : "+v"(lane));
    const float* R = reinterpret_cast<const float*>(p.R) + n0w + 4 * (lane & 7);
    // (the gathered / direct choice is hoisted over the whole block: a per-load "index or load" select makes hipcc
    //  branch around every load and wait vmcnt(0) behind each - eight serialised L2 round trips inside the K loop)
    // (element offsets as 32-bit ints: the host sends tensors of 2^31 elements or more to the 128 x 128 tiling - eight
    //  64-bit offsets held across the last K step were what made the fp8-weight instantiation spill)
    int roff[MI][2];
    if (p.ridx != nullptr) {
      int gi[MI][2];                                      // gathered-row indices first, all in flight together
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) gi[mi][j] = p.ridx[m_base + mi * 16 + j * 8 + (lane >> 3)];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int m = m_base + mi * 16 + j * 8 + (lane >> 3);
          roff[mi][j] = ((m / p.r_np) * p.r_nsrc + gi[mi][j]) * (int)p.ldr;
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j) roff[mi][j] = (m_base + mi * 16 + j * 8 + (lane >> 3)) * (int)p.ldr;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        pre.r[mi][j] = *reinterpret_cast<const float4*>(R + roff[mi][j]);            // column half 0
        pre.r[mi][2 + j] = *reinterpret_cast<const float4*>(R + roff[mi][j] + 32);   // column half 1
      }
    pre.valid = true;
  }
}
template <int EPI, bool SF32, int MI>
__device__ __forceinline__ void prefetch_resid(const GemmParams& p, ResidPrefetch<MI>& pre, int m_base, int n0w,
                                               int l15, int g) {
  pre.valid = false;
  if constexpr (nat_order(EPI, SF32) && EPI == EPI_RESID && MI <= 4) {
    if (n0w + 64 <= p.N && m_base + MI * 16 <= p.M) {   // interior tile: no guards needed
      const float* R = reinterpret_cast<const float*>(p.R);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = m_base + mi * 16 + l15;
        long rrow = m;
        if (p.ridx != nullptr) rrow = (long)(m / p.r_np) * p.r_nsrc + p.ridx[m];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          pre.r[mi][ni] = *reinterpret_cast<const float4*>(R + rrow * p.ldr + n0w + 16 * ni + 4 * g);
      }
      pre.valid = true;
    }
  }
}

// shared tail of every bf16 tiling: bias/gamma for this lane's columns, then one row per m-tile
template <int EPI, bool SF32, int MI, bool W8 = false>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, f32x4 (&acc)[4][MI], int m_base, int n0w,
                                              int l15, int g, ResidPrefetch<MI>& pre, int m_lo = 0,
                                              bool interior = false, char* scratch = nullptr) {
  constexpr int MAP = col_map(EPI, SF32);
  constexpr bool NAT = MAP == MAP_NAT;
  if constexpr (NAT && EPI == EPI_RESID && MI <= 4) {
    if (scratch != nullptr && pre.valid) {
      // ROWMAJOR epilogue of an interior tile (see prefetch_resid_rowmajor): per row group and column half, the wave's 16 x 32
      // accumulator block goes through its 2 KiB LDS scratch - written in the MFMA layout (lane = row l15, columns 4 g..),
      // read back as 8 rows x 128 bytes per instruction (16-byte chunk c of row r in slot c ^ (r & 7): the eight contiguous
      // lanes of a ds_write_b128 group hit eight slots, the sixteen lanes of a ds_read_b128 group sixteen of a 256-byte bank row) - meets the residual rows loaded in that shape, and leaves as whole 128-byte lines.
      int lane = g * 16 + l15;
      asm volatile("" : "+v"(lane));                              // (not hoisted out of the tile loop: see prefetch_resid_rowmajor)
      const int rr = lane >> 3, cc = lane & 7;                    // read-back: row rr (+ 8 j), 16-byte chunk cc of the half
      // column constants of the lane's 2 x 4 columns; fp8 weights: out = resid + gamma * (acc * ws + bias) = resid + (gamma
      // * ws) * acc ... is NOT used - the product gamma * ws would round differently from the other tilings' (acc * ws) +
      // bias, and sub-batches (128 x 128 tiling) must reproduce the full batch bit for bit
      float bias[2][4], gam[2][4], wsc[2][W8 ? 4 : 1];
#pragma unroll
      for (int hc = 0; hc < 2; ++hc) {
        const int n = n0w + 32 * hc + 4 * cc;
        const float4 bq = p.bias != nullptr ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 gq = p.gamma != nullptr ? *reinterpret_cast<const float4*>(p.gamma + n) : make_float4(1.f, 1.f, 1.f, 1.f);
        bias[hc][0] = bq.x; bias[hc][1] = bq.y; bias[hc][2] = bq.z; bias[hc][3] = bq.w;
        gam[hc][0] = gq.x; gam[hc][1] = gq.y; gam[hc][2] = gq.z; gam[hc][3] = gq.w;
        if constexpr (W8) {
          const float4 wq = *reinterpret_cast<const float4*>(p.wscale + n);
          wsc[hc][0] = wq.x; wsc[hc][1] = wq.y; wsc[hc][2] = wq.z; wsc[hc][3] = wq.w;
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every load so far has landed, on every path (see the note below)
      float* Y = reinterpret_cast<float*>(p.Y) + n0w + 4 * cc;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {   // n-tile 2 hc + q: chunk 4 q + g of row l15
            const int slot = (4 * q + g) ^ (l15 & 7);
            *reinterpret_cast<bf16x8*>(scratch + l15 * 128 + slot * 16) = __builtin_bit_cast(bf16x8, acc[2 * hc + q][mi]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = 8 * j + rr, slot = cc ^ (row & 7);
            const f32x4 a = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(scratch + row * 128 + slot * 16));
            const float4 r = pre.r[mi][2 * hc + j];
            float4 o;
            o.x = fmaf(gam[hc][0], W8 ? fmaf(a[0], wsc[hc][0], bias[hc][0]) : a[0] + bias[hc][0], r.x);
            o.y = fmaf(gam[hc][1], W8 ? fmaf(a[1], wsc[hc][1], bias[hc][1]) : a[1] + bias[hc][1], r.y);
            o.z = fmaf(gam[hc][2], W8 ? fmaf(a[2], wsc[hc][2], bias[hc][2]) : a[2] + bias[hc][2], r.z);
            o.w = fmaf(gam[hc][3], W8 ? fmaf(a[3], wsc[hc][3], bias[hc][3]) : a[3] + bias[hc][3], r.w);
            *reinterpret_cast<float4*>(Y + (long)(m_base + mi * 16 + row) * p.ldc + 32 * hc) = o;
          }
          __builtin_amdgcn_wave_barrier();   // the block's reads are issued before the next block's writes (LDS is in order per wave)
        }
      return;
    }
  }
  if constexpr (NAT && EPI == EPI_RESID && MI > 4 && !W8) {
    // the 256 x 256 tiling (fc2 where its tile count fills the rounds better; in place, no gathered rows): the same full-line
    // epilogue, one 16-row group at a time with its residual rows loaded right here (128 accumulators leave no registers
    // to prefetch them under the last K step or to run a group ahead: both forms spilled 196-392 bytes)
    if (scratch != nullptr && interior && p.ridx == nullptr) {
      int lane = g * 16 + l15;
      asm volatile("" : "+v"(lane));
      const int rr = lane >> 3, cc = lane & 7;
      float bias[2][4], gam[2][4];
#pragma unroll
      for (int hc = 0; hc < 2; ++hc) {
        const int n = n0w + 32 * hc + 4 * cc;
        const float4 bq = p.bias != nullptr ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 gq = p.gamma != nullptr ? *reinterpret_cast<const float4*>(p.gamma + n) : make_float4(1.f, 1.f, 1.f, 1.f);
        bias[hc][0] = bq.x; bias[hc][1] = bq.y; bias[hc][2] = bq.z; bias[hc][3] = bq.w;
        gam[hc][0] = gq.x; gam[hc][1] = gq.y; gam[hc][2] = gq.z; gam[hc][3] = gq.w;
      }
      const float* R = reinterpret_cast<const float*>(p.R) + n0w + 4 * cc;
      float* Y = reinterpret_cast<float*>(p.Y) + n0w + 4 * cc;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        float4 r[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const long ro = (long)(m_base + mi * 16 + 8 * j + rr) * p.ldr;
          r[j] = *reinterpret_cast<const float4*>(R + ro);
          r[2 + j] = *reinterpret_cast<const float4*>(R + ro + 32);
        }
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int slot = (4 * q + g) ^ (l15 & 7);
            *reinterpret_cast<bf16x8*>(scratch + l15 * 128 + slot * 16) = __builtin_bit_cast(bf16x8, acc[2 * hc + q][mi]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = 8 * j + rr, slot = cc ^ (row & 7);
            const f32x4 a = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(scratch + row * 128 + slot * 16));
            float4 o;
            o.x = fmaf(gam[hc][0], a[0] + bias[hc][0], r[2 * hc + j].x);
            o.y = fmaf(gam[hc][1], a[1] + bias[hc][1], r[2 * hc + j].y);
            o.z = fmaf(gam[hc][2], a[2] + bias[hc][2], r[2 * hc + j].z);
            o.w = fmaf(gam[hc][3], a[3] + bias[hc][3], r[2 * hc + j].w);
            *reinterpret_cast<float4*>(Y + (long)(m_base + mi * 16 + row) * p.ldc + 32 * hc) = o;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) on every path (see the note below)
      return;
    }
  }
  if constexpr (MAP == MAP_SEC && (EPI == EPI_BIAS || EPI == EPI_GELU)) {
    // interior tile of a bf16-output launch (QKV, FC1 - the bulk of all tiles): no row or column guard, the
    // lane's 2 x 8 columns of bias (and fp8 scale) as four 16-byte loads, two 16-byte stores per row.  The
    // guarded general path below costs ~3x the instructions in exec-mask branches alone.
    if (interior) {
      const int ca = n0w + 8 * g, cb = ca + 32;
      float bs[16], ws[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { bs[j] = 0.f; ws[j] = 1.f; }
      if (p.bias != nullptr) {
        load8<float>(p.bias + ca, bs);
        load8<float>(p.bias + cb, bs + 8);
      }
      if constexpr (W8) {
        load8<float>(p.wscale + ca, ws);
        load8<float>(p.wscale + cb, ws + 8);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see the note in the general path
      bf16_t* Y = reinterpret_cast<bf16_t*>(p.Y);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {   // accumulator (ni, rg) = j>>2, j&3  ->  v[0..7] columns ca.., v[8..15] columns cb..
          const int ni = j >> 2, rg = j & 3, o = (ni >> 1) * 8 + (ni & 1) * 4 + rg;
          v[o] = W8 ? fmaf(acc[ni][mi][rg], ws[o], bs[o]) : acc[ni][mi][rg] + bs[o];
        }
        if (EPI == EPI_GELU) {
#pragma unroll
          for (int j = 0; j < 16; j += 4) {
            f32x2 a = f32x2{v[j], v[j + 1]}, b = f32x2{v[j + 2], v[j + 3]};
            gelu_pk4(a, b);
            v[j] = a[0]; v[j + 1] = a[1]; v[j + 2] = b[0]; v[j + 3] = b[1];
          }
        }
        if (scratch != nullptr) {
          // whole 128-byte lines: the wave's 16 x 64 bf16 block (one line per row) through its 2 KiB LDS scratch - written as
          // the lane's two 16-byte pieces (chunks g and 4 + g of row l15), read back 8 rows x 128 bytes per instruction
          // (same swizzle as the fp32-stream transpose above: conflict free both ways)
          int lane = g * 16 + l15;
          asm volatile("" : "+v"(lane));
          const int rr = lane >> 3, cc = lane & 7;
          const int key = l15 & 7;
          *reinterpret_cast<bf16x8*>(scratch + l15 * 128 + ((g ^ key) << 4)) = __builtin_bit_cast(bf16x8, pack8(v));
          *reinterpret_cast<bf16x8*>(scratch + l15 * 128 + (((4 + g) ^ key) << 4)) = __builtin_bit_cast(bf16x8, pack8(v + 8));
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = 8 * j + rr;
            const uint4 q = __builtin_bit_cast(uint4, *reinterpret_cast<const bf16x8*>(scratch + row * 128 + ((cc ^ (row & 7)) << 4)));
            store_u4<RAJNI_FC1_NT && EPI == EPI_GELU>(Y + (long)(m_base + mi * 16 + row) * p.ldc + n0w + 8 * cc, q);
          }
          __builtin_amdgcn_wave_barrier();
        } else {
          bf16_t* row = Y + (long)(m_base + mi * 16 + l15) * p.ldc;
          store_u4<RAJNI_FC1_NT && EPI == EPI_GELU>(row + ca, pack8(v));
          store_u4<RAJNI_FC1_NT && EPI == EPI_GELU>(row + cb, pack8(v + 8));
        }
      }
      return;
    }
  }
  float bias[16], gam[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int n = out_col<MAP>(n0w, g, j);
    bias[j] = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
    gam[j] = (EPI == EPI_RESID && p.gamma != nullptr && n < p.N) ? p.gamma[n] : 1.f;
  }
  float wsc[W8 ? 16 : 1];
  if constexpr (W8) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = out_col<MAP>(n0w, g, j);
      wsc[j] = n < p.N ? p.wscale[n] : 0.f;
    }
  }
  // Every vector-memory LOAD issued so far (these, and the residual rows prefetched in the last K step) has
  // landed after this explicit wait, on EVERY path - including the ones that skip all uses (a row tile
  // past M).  Without it hipcc's waitcnt pass carries "register X may still be the target of a load" back
  // into the persistent K loop, and the first reuse of X there gets an s_waitcnt vmcnt(0) that drains the
  // DMA queue in every K step (measured: -15 % on the 3-stage tiling).  It costs nothing here: the uses
  // below need the same wait, and vector-memory ops retire in order.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  // W8 dequantises inside the bias add - ONE fused multiply-add fmaf(acc, scale of W row n, bias), the same expression on every
  // epilogue path of every tiling, so that which tiling served a launch never shows in the result
  auto sb = [&](float a, int j) { return W8 ? fmaf(a, wsc[W8 ? j : 0], bias[j]) : a + bias[j]; };
  if constexpr (NAT && EPI == EPI_RESID && MI <= 4) {
    if (pre.valid) {   // interior tile whose residual rows were prefetched during the last K step
      float* Y = reinterpret_cast<float*>(p.Y);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float4 o;
          o.x = fmaf(gam[4 * ni + 0], sb(acc[ni][mi][0], 4 * ni + 0), pre.r[mi][ni].x);
          o.y = fmaf(gam[4 * ni + 1], sb(acc[ni][mi][1], 4 * ni + 1), pre.r[mi][ni].y);
          o.z = fmaf(gam[4 * ni + 2], sb(acc[ni][mi][2], 4 * ni + 2), pre.r[mi][ni].z);
          o.w = fmaf(gam[4 * ni + 3], sb(acc[ni][mi][3], 4 * ni + 3), pre.r[mi][ni].w);
          *reinterpret_cast<float4*>(Y + (long)(m_base + mi * 16 + l15) * p.ldc + n0w + 16 * ni + 4 * g) = o;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m_base + mi * 16 + l15;
    if (m >= p.M || m < m_lo) continue;
    float v[16];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) v[ni * 4 + rg] = sb(acc[ni][mi][rg], ni * 4 + rg);
    if (NAT) epilogue_row_nat<EPI>(p, m, n0w, g, v, gam);
    else epilogue_row<EPI, SF32>(p, m, n0w + 8 * g, n0w + 32 + 8 * g, v, gam);
  }
}
// XCD-aware tile id: blocks b and b+8 share an XCD; give each XCD a contiguous range of tiles
__device__ __forceinline__ int xcd_tile_of(int v, int total) {
  const int q = total >> 3, r = total & 7, xcd = v & 7, loc = v >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}
__device__ __forceinline__ int xcd_tile(int total) { return xcd_tile_of(blockIdx.x, total); }

// global source of the 16-byte chunk (row m, K offset k..) of the X operand; ET = element type
template <int ALOAD, typename ET = bf16_t>
struct XSource {
  static constexpr int CH = 16 / (int)sizeof(ET);   // elements per 16-byte chunk
  const ET* base;  // PLAIN: row pointer (+ swizzled chunk); PATCH: patch origin in the image
  int kofs;        // PATCH: this lane's k offset inside a K step
  __device__ __forceinline__ void init(const GemmParams& p, int m, int chunk) {
    const ET* X = reinterpret_cast<const ET*>(p.X);
    if (ALOAD == ALOAD_PLAIN) {
      base = X + (long)m * p.lda + chunk * CH;
      kofs = 0;
    } else {
      const int b = m / p.npatch, pp = m - b * p.npatch;
      const int py = pp / p.gw, px = pp - py * p.gw;
      base = X + ((long)b * p.cin * p.S + (py << p.log2ps)) * p.S + (px << p.log2ps);
      kofs = chunk * CH;
    }
  }
  __device__ __forceinline__ const ET* at(const GemmParams& p, int k0) const {
    if (ALOAD == ALOAD_PLAIN) return base + k0;
    const int k = k0 + kofs, ps2 = 2 * p.log2ps;
    const int ch = k >> ps2, rem = k & ((1 << ps2) - 1);
    const int ky = rem >> p.log2ps, kx = rem & ((1 << p.log2ps) - 1);
    return base + ((long)ch * p.S + ky) * p.S + kx;
  }
};

// =============================================================================================
// stream tilings: 256(M) x BN x 64(K), 8 waves, ONE persistent workgroup per CU that walks its tiles
// as one continuous stream of K steps.  Measured on the chip (tools/mfma_probe.hip, tools/
// gemm_stamps.py, PMC runs under profiles/): these GEMMs are bound by the bytes a CU can stage per
// second through its load path (35-50 GB/s/CU sustained), not by MFMA issue, LDS or HBM, so the
// levers are bytes per FLOP (tile size, full 128-byte lines) and how far ahead the DMA runs.
//   wide <WM=2,WN=4,MI=8,NS=2>: 256x256, wave tile 128x64, 2 stages x 64 KiB; 16 KiB staged per 2.1
//        MFLOP, DMA one step ahead.  Best for wide outputs (qkv, fc1).
//   mid  <WM=4,WN=2,MI=4,NS=3>: 256x128, wave tile 64x64, 3 stages x 48 KiB; 24 KiB per 2.1 MFLOP
//        but DMA two steps ahead and twice the tiles: for N = 768-class outputs (proj, fc2), where
//        591 tiles of 256x256 make 2.3 rounds on 256 CUs.
// K step j (stage s = j mod NS), registers holding the ks=0 fragments of step j on entry:
//   half 1: MFMAs(j,ks0)  ||  LDS reads of (j,ks1)
//   s_waitcnt lgkmcnt(0) vmcnt((NS-2)*PIECES) ; s_barrier  -> stage s is free, step j+1 has landed
//   half 2: MFMAs(j,ks1)  ||  LDS reads of (j+1,ks0)  ||  LDS-DMA of step j+NS into stage s
// The stream continues ACROSS tile boundaries (the DMA pointers switch to the next tile NS steps
// early), so the next tile's first loads fly during the epilogue.  One raw s_barrier per K step;
// every LDS read and DMA issue sits between MFMAs (sched_group_barrier pins the order).
// =============================================================================================
#define RAJNI_GEMM_NBLK_BYTES (1600 * 1024)
#ifndef RAJNI_TSTORE
#define RAJNI_TSTORE 1
#endif
namespace wide {
constexpr int BM = 256, BK = 64;
constexpr int X_BYTES = BM * BK * 2;            // 32 KiB
__device__ __forceinline__ int key_x(int row) { return (row >> 1) & 7; }

template <int WN_, int NS_, bool W8_ = false, int NW_ = 8> struct Cfg {   // WN_ = 64-column groups of the tile
  static constexpr int BN = WN_ * 64;
  static constexpr int WB = W8_ ? 1 : 2;         // bytes per W element
  static constexpr int W_BYTES = BN * BK * WB;
  static constexpr int STAGE_BYTES = X_BYTES + W_BYTES;
  static constexpr int LDS_BYTES = NS_ * STAGE_BYTES;
  static constexpr int XP = X_BYTES / 1024 / NW_;   // X pieces (1 KiB) per wave per K step: 4 with 8 waves
  static constexpr int PW = W_BYTES / 1024 / NW_;   // W pieces per wave per K step
  static constexpr int PIECES = XP + PW;
  // W pieces 32 tile rows apart share their swizzle key (every map): one pointer per piece of the first 32 rows
  static constexpr int NPW = PW < (W8_ ? 2 : 4) ? PW : (W8_ ? 2 : 4);
};

// issue order of one half step: MI groups of {NI MFMAs, 1-2 fragment reads, DMA pieces}, reads and pieces spread
// evenly over the groups (front-loading either was measured and is slower: the forward 9.03 -> 9.45 ms)
__host__ __device__ constexpr int dma_first(int g, int pieces, int groups) { return g * pieces / groups; }
__host__ __device__ constexpr int rd_first(int g, int mi, int ni) { return g + g * ni / mi; }
template <int G, int MI, bool DMA, int PIECES, int NI = 4>
__device__ __forceinline__ void sched_half() {
  if constexpr (G < MI) {
    __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
    if constexpr (rd_first(G + 1, MI, NI) - rd_first(G, MI, NI) > 0)
      __builtin_amdgcn_sched_group_barrier(0x100, rd_first(G + 1, MI, NI) - rd_first(G, MI, NI), 0);
    if constexpr (DMA) {
      constexpr int nd = dma_first(G + 1, PIECES, MI) - dma_first(G, PIECES, MI);
      if constexpr (nd > 0) __builtin_amdgcn_sched_group_barrier(0x020, nd, 0);
    }
    sched_half<G + 1, MI, DMA, PIECES, NI>();
  }
}
template <int N> __device__ __forceinline__ void wait_step() {   // lgkmcnt(0) + counted vmcnt
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(N) : "memory");
}

// Eight waves of (MI*16) x 64 outputs each.
// TAG does nothing in the body: residual launches with K <= N (the attention projection, bound by its fp32-stream
// epilogue) run an instantiation of their own so that profilers list them apart from fc2, like bench.py's classes.
template <int EPI, int ALOAD, bool SF32, int WM, int WN, int MI, int NS, bool W8 = false, int TAG = 0>
__global__ void __launch_bounds__(WM * WN * 64, 2) gemm_bf16_tn_stream(const GemmParams p) {
  using C = Cfg<WN, NS, W8, WM * WN>;
  constexpr int NI = 4;                        // 16-column n-tiles per wave
  using WFrag = typename WFragT<W8>::type;   // a W fragment as it sits in LDS: 8 bf16, or 8 fp8 bytes
  constexpr int MAP = col_map(EPI, SF32);      // W-row permutation = which output columns a lane owns
  static_assert(WM * WN == 8 && WM * MI * 16 == BM, "8 waves covering 256 rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // fp32-stream RESID launches of the 256 x 128 tiling: full-line residual loads / output stores through a 2 KiB per-wave
  // LDS transpose behind the stages (host: RESID_SCRATCH_BYTES more dynamic LDS) - see prefetch_resid_rowmajor
  // (bf16 weights: with the fp8-weight scale on top the instantiation spills - it keeps the accumulator-layout epilogue)
  constexpr bool ROWMAJOR = nat_order(EPI, SF32) && EPI == EPI_RESID && MI <= 4;
  constexpr bool TSTORE = RAJNI_TSTORE && (EPI == EPI_BIAS || EPI == EPI_GELU);   // bf16 outputs leave as whole lines too
  constexpr bool ROWMAJOR_WIDE = nat_order(EPI, SF32) && EPI == EPI_RESID && MI > 4 && !W8;    // loads in the epilogue, group by group

  // ---- staging: a piece = 1 KiB = 8 rows x 128 B; wave w stages X pieces 4w..4w+3, W pieces PW*w..
  //      (fp8 W: a piece = 16 rows x 64 B, lane -> row lane>>2, 16-byte unit lane&3)
  // NOTE on addressing: the DMA takes a 64-bit per-lane pointer (global_load_lds ... off); scalar base +
  // 32-bit lane offset and buffer_load ... lds were built and measured, neither is faster.
  // Register diet (the RESID instantiations are register bound, and a spilled lane constant is reloaded
  // behind an s_waitcnt vmcnt(0) that drains the DMA queue):
  //   * a ragged last row tile starts at M - 256 instead of clamping its rows (the rows it shares with the
  //     previous tile are recomputed and masked in the epilogue), so every tile's rows are evenly spaced:
  //     X needs TWO pointers per lane (pieces 0/2 and 1/3 differ by 16 rows = a scalar), and
  //   * going to the next tile is `pointer += scalar delta` - no second pointer set, nothing for the
  //     compiler to hoist (it used to precompute the next tile's 6 pointers at the start of every tile).
  auto tile_m0 = [&](int tm_) { return tm_ * BM + BM > p.M ? p.M - BM : tm_ * BM; };   // host: M >= 256
  // Tile order.  An XCD works through a contiguous range of tile ids, its 32 CUs on 32 consecutive ones.
  // With id = (row tile, column tile) column-fastest, those 32 tiles span ALL column tiles: every XCD
  // streams the whole W once per round, and W (4.7 MB for FC1) does not stay in a 4 MB L2 next to the X
  // panels - measured 387 MB fetched per FC1 launch for 63 MB of operands.  Instead the columns are cut
  // into blocks of `nblk` column tiles and the ids run (block, row tile, column in block): an XCD stays in
  // one block for many rounds (its W slice stays L2 resident), at the price of reading X once per block.
  const int tiles_m = p.total_tiles / p.tiles_n;
  auto tile_mn = [&](int t, int& tm_, int& tn_) {
    if (p.nblk >= p.tiles_n) { tm_ = t / p.tiles_n; tn_ = t - tm_ * p.tiles_n; return; }
    const int per = p.nblk * tiles_m, blk = t / per, r = t - blk * per;
    const int left = p.tiles_n - blk * p.nblk, nb = left < p.nblk ? left : p.nblk;
    const int rr = r / nb;
    tm_ = rr; tn_ = blk * p.nblk + (r - rr * nb);
  };
  XSource<ALOAD> xs[ALOAD == ALOAD_PLAIN ? 1 : C::XP];   // fused im2col loader: one source per piece
  const char* xp[2];                                 // plain loader: even and odd pieces (16 rows apart each)
  const char* ws[C::NPW];
  auto point_at = [&](int tile) {   // DMA source pointers of a tile
    int tm, tn;
    tile_mn(tile, tm, tn);
    const int r_in = lane >> 3, pch = lane & 7;
    if constexpr (ALOAD == ALOAD_PLAIN) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (wave * C::XP + i) * 8 + r_in;     // key_x(row + 16) == key_x(row)
        xp[i] = reinterpret_cast<const char*>(p.X) + ((long)(tile_m0(tm) + row) * p.lda + (pch ^ key_x(row)) * 8) * 2;
      }
    } else {
#pragma unroll
      for (int i = 0; i < C::XP; ++i) {
        const int row = (wave * C::XP + i) * 8 + r_in;
        int m = tm * BM + row;
        if (m > p.M - 1) m = p.M - 1;  // clamp: duplicates are computed but never stored
        xs[i].init(p, m, pch ^ key_x(row));
      }
    }
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
      if constexpr (W8) {
        const int row = (wave * C::PW + i) * 16 + (lane >> 2);
        ws[i] = reinterpret_cast<const char*>(p.W) + (long)(tn * C::BN + row) * p.ldw + (((lane & 3) ^ w_key8<MAP>(row)) << 4);
      } else {
        const int row = (wave * C::PW + i) * 8 + r_in;
        ws[i] = reinterpret_cast<const char*>(p.W) + ((long)(tn * C::BN + row) * p.ldw + (pch ^ w_key<MAP>(row)) * 8) * 2;
      }
    }
  };
  auto advance = [&](int from, int to) {   // retarget the DMA from tile `from` to tile `to`
    if constexpr (ALOAD == ALOAD_PLAIN) {
      int tm0, tn0, tm1, tn1;
      tile_mn(from, tm0, tn0);
      tile_mn(to, tm1, tn1);
      const long dx = (long)(tile_m0(tm1) - tile_m0(tm0)) * p.lda * 2;
      const long dw = (long)(tn1 - tn0) * C::BN * p.ldw * C::WB;
      xp[0] += dx; xp[1] += dx;
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) ws[i] += dw;
    } else {
      point_at(to);
    }
  };
  auto dma_piece = [&](int q, int k0, char* dx) {   // piece q of this wave: X 0..3 then W 0..PW-1
    if (q < C::XP) {
      if constexpr (ALOAD == ALOAD_PLAIN)
        __builtin_amdgcn_global_load_lds(GLB_PTR(xp[q & 1] + ((long)(q >> 1) * 16 * p.lda + k0) * 2),
                                         LDS_PTR(dx + (wave * C::XP + q) * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds(GLB_PTR(xs[q].at(p, k0)), LDS_PTR(dx + (wave * C::XP + q) * 1024), 16, 0, 0);
    } else {
      const int i = q - C::XP;   // W piece i: pointer i % NPW, (i / NPW) * 32 tile rows further down
      if constexpr (C::PW <= C::NPW)
        __builtin_amdgcn_global_load_lds(GLB_PTR(ws[i] + k0 * C::WB), LDS_PTR(dx + X_BYTES + (wave * C::PW + i) * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds(GLB_PTR(ws[i % C::NPW] + ((long)(i / C::NPW) * 32 * p.ldw + k0) * C::WB),
                                         LDS_PTR(dx + X_BYTES + (wave * C::PW + i) * 1024), 16, 0, 0);
    }
  };
  auto stage = [&](int kt, int st) {
#pragma unroll
    for (int q = 0; q < C::PIECES; ++q) dma_piece(q, kt * BK, smem + st * C::STAGE_BYTES);
  };

  // ---- fragment addresses: rows base + (lane&15) have key (lane&15)>>1 for every mi, the permuted
  //      W rows have one key for every ni, so each (operand, ks) needs ONE address + immediates
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;
  const int xr0 = wm * (MI * 16) + l15;
  const int wr0 = wn * 64 + w_frag_row<MAP>(l15, 0);
  // byte offset of n-tile ni's W rows from n-tile 0's (lane independent; the swizzle key is the same)
  auto w_ni_off = [](int ni) { return (w_frag_row<MAP>(0, ni) - w_frag_row<MAP>(0, 0)) * (64 * C::WB); };
  int xo[2], wo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    xo[ks] = xr0 * 128 + (((ks * 4 + g) ^ key_x(xr0)) << 4);
    if constexpr (W8)   // 8-byte fragment ks*4+g = half (g&1) of 16-byte unit (ks*4+g)>>1
      wo[ks] = X_BYTES + wr0 * 64 + ((((ks * 4 + g) >> 1) ^ w_key8<MAP>(wr0)) << 4) + ((g & 1) << 3);
    else
      wo[ks] = X_BYTES + wr0 * 128 + (((ks * 4 + g) ^ w_key<MAP>(wr0)) << 4);
  }

  f32x4 acc[4][MI];  // [ni][mi]

  // one half step: MFMAs on (xc,wc) || fragments (stage rst, sub-step rks) -> (xn,wn_), xn[mi] issued
  // right after the group that consumed xc[mi] || if DMA: K-tile dkt of the pointed-at tile -> stage dst
  auto half = [&](auto dma_c, auto read_c, bf16x8 (&xc)[MI], WFrag (&wc)[NI], bf16x8 (&xn)[MI], WFrag (&wn_)[NI],
                  int rst, int rks, int dkt, int dst) {
    constexpr bool DMA = decltype(dma_c)::value;
    constexpr bool READ = decltype(read_c)::value;
    const char* sb = smem + rst * C::STAGE_BYTES;
    char* dx = smem + dst * C::STAGE_BYTES;
    const int k0 = dkt * BK;
    bf16x8 wv[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wv[ni] = w_frag_bf16(wc[ni]);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[ni], xc[mi], acc[ni][mi], 0, 0, 0);
      if constexpr (READ) {
        xn[mi] = *reinterpret_cast<const bf16x8*>(sb + xo[rks] + mi * 2048);
#pragma unroll
        for (int wi = mi * NI / MI; wi < (mi + 1) * NI / MI; ++wi)
          wn_[wi] = *reinterpret_cast<const WFrag*>(sb + wo[rks] + w_ni_off(wi));
      }
      if constexpr (DMA) {
#pragma unroll
        for (int q = dma_first(mi, C::PIECES, MI); q < dma_first(mi + 1, C::PIECES, MI); ++q) dma_piece(q, k0, dx);
      }
    }
    if constexpr (READ) sched_half<0, MI, DMA, C::PIECES, NI>();
  };
  using T = std::true_type; using F = std::false_type;
  const int nk = p.K / BK;        // >= NS + 1 (host checked)
  int v = blockIdx.x;
  int tile = xcd_tile_of(v, p.total_tiles);
  bf16x8 xa[MI], xb[MI];
  WFrag wa[NI], wb[NI];
  if constexpr (EPI == EPI_RESID) {
    if (p.stagger > 0 && ((blockIdx.x >> 3) & 1))
      for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  constexpr int WBASE = (NS - 2) * C::PIECES;
  auto interior = [&](int t) {   // a tile whose 256 rows and BN columns all exist
    int tm_, tn_;
    tile_mn(t, tm_, tn_);
    return tm_ * BM + BM <= p.M && tn_ * C::BN + C::BN <= p.N;
  };
  ResidPrefetch<MI> pre;
  point_at(tile);
#pragma unroll
  for (int j = 0; j < NS; ++j) stage(j, j);
  wait_step<((NS - 1) * C::PIECES > 8) ? 0 : (NS - 1) * C::PIECES>();   // step 0 landed (NS=3: drain all)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < NI; ++i) wa[i] = *reinterpret_cast<const WFrag*>(smem + wo[0] + w_ni_off(i));
#pragma unroll
  for (int i = 0; i < MI; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(smem + xo[0] + i * 2048);
  int st = 0;  // LDS stage of the current K step
  // Output stores of a tile are still in flight when the next tile's K loop starts.  gfx9 counts stores in
  // vmcnt as well and retires vector-memory ops in order, so the first counted wait of a tile may leave
  // exactly the epilogue's stores outstanding instead of waiting for them to reach L2: an interior tile's
  // epilogue issues NSTORE stores per wave after the DMA of K step 1 (a lower bound is all that is needed).
  constexpr int NSTORE_ALL = nat_order(EPI, SF32) ? 4 * MI : 2 * MI;
  constexpr int NSTORE = NSTORE_ALL < 48 ? NSTORE_ALL : 48;   // vmcnt is a 6-bit counter
  bool prev_full = false;          // the previous tile of this workgroup was interior

  while (true) {
#ifdef RAJNI_GEMM_STAMPS
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    int tm, tn;
    tile_mn(tile, tm, tn);
    const int m0 = ALOAD == ALOAD_PLAIN ? tile_m0(tm) : tm * BM, n0 = tn * C::BN;
    const int m_lo = tm * BM;            // rows below it belong to the previous tile (ragged last tile only)
    const bool inter = interior(tile);
    const int vn = v + gridDim.x;
    const bool more = vn < p.total_tiles;
    pre.valid = false;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto kstep = [&](int kt) {   // one K step
      // the DMA of step kt loads K-tile kt+NS; from kt = nk-NS on that is the NEXT tile's K-tile
      // 0.. (when there is no next tile the pointers stay put: harmless re-loads nobody reads)
      if (kt == nk - NS && more) advance(tile, xcd_tile_of(vn, p.total_tiles));
      const int dkt = kt + NS < nk ? kt + NS : kt + NS - nk;
      const int st1 = st + 1 == NS ? 0 : st + 1;
      if (kt == nk - 1 && inter) {
        if constexpr (ROWMAJOR) prefetch_resid_rowmajor<MI>(p, pre, m0 + wm * (MI * 16), n0 + wn * 64, lane);
        else prefetch_resid<EPI, SF32, MI>(p, pre, m0 + wm * (MI * 16), n0 + wn * 64, l15, g);
      }
      half(F{}, T{}, xa, wa, xb, wb, st, 1, 0, 0);
      // my reads of stage st are done and my DMA pieces of step kt+1 have landed ...
      if (NSTORE > 0 && kt == 0 && prev_full) wait_step<WBASE + NSTORE>();
      else wait_step<WBASE>();
      __builtin_amdgcn_s_barrier();   // ... and everyone else's: stage st is free, stage st1 readable
      asm volatile("" ::: "memory");
      half(T{}, T{}, xb, wb, xa, wa, st1, 0, dkt, st);
      st = st1;
    };
    for (int kt = 0; kt < nk; ++kt) kstep(kt);
#ifdef RAJNI_GEMM_STAMPS
    asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[3][MI - 1][3]));
    const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue (the next tile's first loads are in flight)
    epilogue_tile<EPI, SF32, MI, W8>(p, acc, m0 + wm * (MI * 16), n0 + wn * 64, l15, g, pre, m_lo, inter,
                                     (ROWMAJOR || ROWMAJOR_WIDE || TSTORE) ? smem + C::LDS_BYTES + wave * 2048 : nullptr);
#ifdef RAJNI_GEMM_STAMPS
    if (p.stamps != nullptr && wave == 0) {
      const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
      if (lane == 0) {
        unsigned long long* o = p.stamps + (size_t)tile * 4;
        o[0] = ts0; o[1] = ts0; o[2] = ts2; o[3] = ts3;
      }
    }
#endif
    prev_full = inter;
    if (!more) break;
    v = vn;
    tile = xcd_tile_of(v, p.total_tiles);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the trailing (unused) DMA before LDS is released
}
}  // namespace wide

// =============================================================================================
// small tiling: 128 x 128 x 64, 2 stages
// =============================================================================================
namespace small {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // X + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // double buffered: 64 KiB
// 128-byte rows: slot = (row&1)*8 + (chunk ^ key)
__device__ __forceinline__ int key_x(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int key_w(int row) { return ((row >> 4) & 3) * 2 + ((row >> 1) & 1); }

template <int EPI, int ALOAD, bool SF32, bool W8 = false>
__global__ void __launch_bounds__(256, 2) gemm_bf16_tn_128x128(const GemmParams p) {
  constexpr int MAP = col_map(EPI, SF32);      // W-row permutation = which output columns a lane owns
  constexpr int PW = W8 ? 2 : 4;               // W pieces per wave per K step (fp8 tile is 8 KiB)
  using WFrag = typename WFragT<W8>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = xcd_tile(p.total_tiles);
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging: wave w stages pieces 4w..4w+3 of each operand; a piece = 8 rows x 128 B
  const int r_in = lane >> 3, pch = lane & 7;
  XSource<ALOAD> xs[4];
  const char* ws[PW];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + r_in;
    int m = m0 + row;
    if (m > p.M - 1) m = p.M - 1;
    xs[i].init(p, m, pch ^ key_x(row));
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    if constexpr (W8) {
      const int row = (wave * PW + i) * 16 + (lane >> 2);
      ws[i] = reinterpret_cast<const char*>(p.W) + (long)(n0 + row) * p.ldw + (((lane & 3) ^ w_key8<MAP>(row)) << 4);
    } else {
      const int row = (wave * PW + i) * 8 + r_in;
      ws[i] = reinterpret_cast<const char*>(p.W) + ((long)(n0 + row) * p.ldw + (pch ^ w_key<MAP>(row)) * 8) * 2;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* sx = smem + buf * STAGE_BYTES;
    char* sw = sx + TILE_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(xs[i].at(p, k0)), LDS_PTR(sx + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PW; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(ws[i] + k0 * (W8 ? 1 : 2)), LDS_PTR(sw + (wave * PW + i) * 1024), 16, 0, 0);
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  int xoff[4], xkey[4], woff[4], wkey[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int xr = wm * 64 + i * 16 + l15;
    xoff[i] = xr * 128; xkey[i] = key_x(xr);
    const int wr = wn * 64 + w_frag_row<MAP>(l15, i);
    woff[i] = wr * (W8 ? 64 : 128); wkey[i] = W8 ? w_key8<MAP>(wr) : w_key<MAP>(wr);
  }

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  ResidPrefetch<4> pre;
  pre.valid = false;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt has landed for every wave; everyone is done reading the other buffer
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    else prefetch_resid<EPI, SF32, 4>(p, pre, m0 + wm * 64, n0 + wn * 64, l15, g);   // lands under the last MFMAs
    const char* sx = smem + (kt & 1) * STAGE_BYTES;
    const char* sw = sx + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + g;
      bf16x8 xf[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xf[i] = *reinterpret_cast<const bf16x8*>(sx + xoff[i] + ((c ^ xkey[i]) << 4));
        if constexpr (W8)
          wf[i] = w_frag_bf16(*reinterpret_cast<const WFrag*>(sw + woff[i] + (((c >> 1) ^ wkey[i]) << 4) + ((c & 1) << 3)));
        else
          wf[i] = *reinterpret_cast<const bf16x8*>(sw + woff[i] + ((c ^ wkey[i]) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    }
  }

  epilogue_tile<EPI, SF32, 4, W8>(p, acc, m0 + wm * 64, n0 + wn * 64, l15, g, pre);
}
}  // namespace small

#include "gemm_f8.h"   // namespace f8: the fp8 x fp8 persistent kernel (v_mfma_f32_16x16x128_f8f6f4)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): `done` is the calling site's
// per-device flag array (a process may drive several GPUs; one process per GPU is the deployment, but a model on
// cuda:1 in a process whose first launch was on cuda:0 must not inherit that device's "done")
template <typename K>
int set_lds_attr(K kernel, int lds, bool (&done_by_device)[RAJNI_MAX_DEVICES]) {
  bool& done = done_by_device[rajni_current_device()];
  if (done) return RAJNI_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) {
    rajni_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
    return RAJNI_ERR_LAUNCH;
  }
  done = true;
  return RAJNI_OK;
}

// =============================================================================================
// fp32 model path: 128 x 128 x 32(fp32) tile - byte-for-byte the LDS geometry of the small bf16
// tiling (128-byte rows, same swizzle keys, same LDS-DMA staging) - with v_mfma_f32_16x16x4_f32
// (exact fp32 FMA chain, 1/16 of the bf16 rate).  A lane reads 16 bytes = 4 consecutive k of its
// row for BOTH operands and feeds them to 4 MFMAs, element t of every lane group g covering
// k = 4*(4*ks+g)+t: the k order inside a K step is permuted identically for X and W, so the sum is
// the same.  All tensors (x, w, resid, pos, y) are fp32.
// =============================================================================================
namespace f32 {
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE_BYTES = BM * BK * 4;
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
using small::key_w;
using small::key_x;

template <int EPI>
__device__ __forceinline__ void epilogue_row_f32(const GemmParams& p, int m, int nb, float* v, const float* gam) {
  long orow = m;
  if (EPI == EPI_GELU) {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = 0.5f * v[j] * (1.0f + erff(v[j] * 0.70710678118654752f));
  } else if (EPI == EPI_RESID) {
    long rrow = m;
    if (p.ridx != nullptr) {
      const int b = m / p.r_np;
      rrow = (long)b * p.r_nsrc + p.ridx[m];
    }
    const long roff = rrow * p.ldr + nb;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (nb + j < p.N) v[j] = fmaf(gam[j], v[j], reinterpret_cast<const float*>(p.R)[roff + j]);
  } else if (EPI == EPI_PATCH) {
    const int b = m / p.npatch, pp = m - b * p.npatch;
    orow = (long)b * (p.npatch + 1) + 1 + pp;
    const float* pr = reinterpret_cast<const float*>(p.pos) + (long)(pp + p.pos_off) * p.ldc + nb;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (nb + j < p.N) v[j] += pr[j];
  }
  const long yoff = orow * p.ldc + nb;
  if (nb + 16 <= p.N) {
    store16<true>(p.Y, yoff, v);
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (nb + j < p.N) reinterpret_cast<float*>(p.Y)[yoff + j] = v[j];
  }
}

template <int EPI, int ALOAD>
__global__ void __launch_bounds__(256, 2) gemm_f32_tn_128x128(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = xcd_tile(p.total_tiles);
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const float* W = reinterpret_cast<const float*>(p.W);

  const int r_in = lane >> 3, pch = lane & 7;
  XSource<ALOAD, float> xs[4];
  const float* ws[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + r_in;
    int m = m0 + row;
    if (m > p.M - 1) m = p.M - 1;
    xs[i].init(p, m, pch ^ key_x(row));
    ws[i] = W + (long)(n0 + row) * p.ldw + (pch ^ key_w(row)) * 4;
  }
  auto stage = [&](int kt, int buf) {
    char* sx = smem + buf * STAGE_BYTES;
    char* sw = sx + TILE_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(xs[i].at(p, k0)), LDS_PTR(sx + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(ws[i] + k0), LDS_PTR(sw + (wave * 4 + i) * 1024), 16, 0, 0);
  };

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
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* sx = smem + (kt & 1) * STAGE_BYTES;
    const char* sw = sx + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 4 + g;
      f32x4 xf[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xf[i] = *reinterpret_cast<const f32x4*>(sx + xoff[i] + ((c ^ xkey[i]) << 4));
        wf[i] = *reinterpret_cast<const f32x4*>(sw + woff[i] + ((c ^ wkey[i]) << 4));
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni][tt], xf[mi][tt], acc[ni][mi], 0, 0, 0);
    }
  }

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
    epilogue_row_f32<EPI>(p, m, nb, v, gam);
  }
}

template <int EPI, int ALOAD>
int launch(GemmParams p, int kclass, hipStream_t s) {
  static bool attr[RAJNI_MAX_DEVICES] = {};
  p.tiles_n = (p.N + 127) / 128;
  p.total_tiles = p.tiles_n * ((p.M + 127) / 128);
  int rc = set_lds_attr(&gemm_f32_tn_128x128<EPI, ALOAD>, LDS_BYTES, attr);
  if (rc != RAJNI_OK) return rc;
  ProfScope prof(kclass, s, 2.0 * p.M * (double)p.N * p.K,
                 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N));
  hipLaunchKernelGGL((gemm_f32_tn_128x128<EPI, ALOAD>), dim3(p.total_tiles), dim3(256), LDS_BYTES, s, p);
  RAJNI_CHECK_LAUNCH("gemm_f32_tn");
  return RAJNI_OK;
}
}  // namespace f32

// x[b,0,:] = cls + pos[0]  (or cls alone when pos has no CLS row)
template <bool SF32, typename T>
__global__ void cls_pos_kernel(const T* cls, const T* pos, int pos_has_cls, void* x,
                               long img_stride, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float v = ld1(cls + c);
  if (pos_has_cls) v += ld1(pos + c);
  store1<SF32>(x, (long)b * img_stride + c, v);
}

// LDS stages of the wide tiling: 2 for bf16 weights (2 x 64 KiB); fp8 weights leave room for 3 (3 x 48 KiB)
#ifndef RAJNI_W8_WIDE_NS
#define RAJNI_W8_WIDE_NS 2
#endif
#define RAJNI_W8_WIDE_NS_OR(w8) ((w8) ? RAJNI_W8_WIDE_NS : 2)
// test / tuning hooks (include/rajni_hip_debug.h): process-global, unsynchronised - not for concurrent use
int g_force_tiling = 0;  // 0 auto, 1 small (128x128x64, 2 stages), 4 wide 256x256x64, 5 mid 256x128x64 (tests)
int g_nblk_bytes = RAJNI_GEMM_NBLK_BYTES;   // W bytes of one N block (0 = plain order, < 0 = forced block size: tuning)

// column tiles per N block of the persistent tile order (see tile_mn).  Measured on the ViT-B shapes
// (tools/nblk_bench.py): blocks of ~1.5 MiB of W help once every block spans at least two rounds of tiles
// (QKV at 197 tokens 179 -> 161 us, FC1 254 -> 239 us); with fewer rounds W is not re-read often enough
// to pay for reading X once per block, and K = 3072 (fc2) never pays.
inline int n_block(int tiles_n, int tiles_m, int bn, int K, int wbytes, int cus) {
  if (g_nblk_bytes < 0) return -g_nblk_bytes < tiles_n ? -g_nblk_bytes : tiles_n;
  if (g_nblk_bytes == 0) return tiles_n;
  const long per_tile = (long)bn * K * wbytes;
  const int fit = (int)(g_nblk_bytes / per_tile);
  if (fit < 1 || fit >= tiles_n) return tiles_n;
  const int blocks = (tiles_n + fit - 1) / fit;
  const int rounds = tiles_n * tiles_m / cus;
  if (rounds < 2 * blocks) return tiles_n;
  return (tiles_n + blocks - 1) / blocks;
}

// N = 768-class outputs with a long K (fc2): 256x128 tiles are the default, but when the tile count of the
// 256x256 tiling happens to fill its rounds much better, it wins by 6-20 % (tools/fc2_grid_probe.py: 19 of 19
// shapes of ViT-B / ViT-L / ViT-H at batch 64-512 picked right; fc2 at 152 tokens 226 -> 212 us, at batch 512
// 660 -> 537 us).  Cost model: full rounds of 256 tiles plus a partial round whose tiles run faster the emptier
// the chip is (board power cap: DESIGN.md section 4 (10)); one 256x256 round = 1.83 256x128 rounds.  Not used for
// K <= N (proj): there the epilogue dominates a tile and the ratio is larger (measured: 4 misses of 11).
inline double eff_rounds(long tiles, int cus) {
  const long full = tiles / cus;
  const double frac = (double)tiles / (double)cus - (double)full;
  return (double)full + (frac > 0.0 ? 0.6 + 0.4 * frac : 0.0);
}
inline bool wide_wins_on_rounds(int M, int N, int cus) {
  const long rows = (M + 255) / 256;
  return 1.83 * eff_rounds(rows * ((N + 255) / 256), cus) < eff_rounds(rows * ((N + 127) / 128), cus);
}

// persistent grid: one workgroup per CU of the device the launch goes to (hipDeviceProp_t::multiProcessorCount;
// 256 on MI355X).  A grid balanced to whole rounds, as hipBLASLt picks for fc2, was measured: 0.7 % slower over
// the 20 GEMM shapes of the schedule.
inline int stream_grid(int total_tiles, int cus) { return total_tiles <= cus ? total_tiles : cus; }

// Every other workgroup of an XCD starts a residual-epilogue launch `units` x 8192 cycles late, so that the two halves of
// the chip do not hit their prologue loads and epilogue bursts in the same instant.  Round 2 (tools/stagger_probe.py,
// tools/ab_forward.py stagger): proj alone -4 % at 1 unit, fc2 alone flat; in the forward 2 units were best (+0.5-0.8 %).
// Round 3, with the full-line epilogue: stand-alone proj now LOSES with any offset (84.4 us at 0, 88.5 at 1, 97.4 at 2 units;
// four or eight phase groups worse still), in the forward 1 unit is best by 0.1-0.3 % (8.850 / 8.804 ms against 8.872 / 8.825
// at 2 units and 8.878 / 8.838 at 0).  rajni_debug_set_resid_stagger(0) turns it off.
int g_resid_stagger = 1;

template <int EPI, int ALOAD, bool SF32, bool W8 = false>
int launch_gemm(GemmParams p, int kclass, hipStream_t s) {
  p.stamps = rajni_g_stamps;
  p.stagger = g_resid_stagger;
  p.tiles_n = (p.N + 127) / 128;
  const int t256 = p.tiles_n * ((p.M + 255) / 256), t128 = p.tiles_n * ((p.M + 127) / 128);
  int mode = g_force_tiling;
  const int cus = rajni_num_cus();
  if ((mode == 4 || mode == 5) && p.M < 256) mode = 1;   // stream tiles may start at M - 256
  if (mode == 4 && p.K < 192) mode = 1;                   // the persistent streams need >= NS + 1 K steps
  if (mode == 5 && p.K < 256) mode = 1;
  // the 256 x 128 tiling's full-line residual epilogue addresses the residual tensor with 32-bit element offsets
  constexpr bool RESID_NAT = nat_order(EPI, SF32) && EPI == EPI_RESID;
  const long resid_rows = p.ridx != nullptr ? (long)(p.M / p.r_np) * p.r_nsrc : (long)p.M;
  const bool mid_ok = !RESID_NAT || resid_rows * p.ldr < (1L << 31);
  if (mode == 5 && !mid_ok) mode = 1;
  if (mode == 0) {
    // measured on ViT-B shapes (tools/gemm_bench.py, tools/proj_probe.py, profiles/):
    //   wide outputs (qkv, fc1): persistent 256x256 (950 / 870 TFLOP/s vs 750 / 700 for 128x128);
    //   N = 768-class outputs (proj, fc2): 591 tiles of 256x256 make 2.3 rounds on 256 CUs, the
    //   persistent 256x128 3-stage tiling is best (proj 125 us, fc2 284 us vs 127 / 310 for 128x128);
    //   small problems (head, tiny batches): 128x128.
    if (p.M >= 1024 && p.N >= 1536 && p.K >= 192) mode = 4;
    else if (p.M >= 1024 && p.K >= 256) mode = (p.K > p.N && p.K >= 1536 && wide_wins_on_rounds(p.M, p.N, cus)) ? 4 : (mid_ok ? 5 : 1);
    else mode = 1;
  }
  static bool attr[5][RAJNI_MAX_DEVICES] = {};   // [0] small, [1] wide, [2] mid, [3] [4] their K<=N twins; per device
  // algorithmic bytes: X + W + output (+ the residual rows read), fp32 where the residual stream is fp32
  constexpr double ysz = (SF32 && (EPI == EPI_RESID || EPI == EPI_PATCH)) ? 4.0 : 2.0;
  constexpr double rsz = EPI == EPI_RESID ? (SF32 ? 4.0 : 2.0) : 0.0;
  ProfScope prof(kclass, s, 2.0 * p.M * (double)p.N * p.K,
                 2.0 * (double)p.M * p.K + (ysz + rsz) * (double)p.M * p.N + (W8 ? 1.0 : 2.0) * (double)p.N * p.K);
  int rc;
  if (mode == 4) {
    using C = wide::Cfg<4, RAJNI_W8_WIDE_NS_OR(W8), W8>;
    constexpr int NS = RAJNI_W8_WIDE_NS_OR(W8);
    constexpr int lds = C::LDS_BYTES + (((RAJNI_TSTORE && (EPI == EPI_BIAS || EPI == EPI_GELU)) || (nat_order(EPI, SF32) && EPI == EPI_RESID && !W8)) ? 8 * 2048 : 0);
    if ((rc = set_lds_attr(&wide::gemm_bf16_tn_stream<EPI, ALOAD, SF32, 2, 4, 8, NS, W8, 0>, lds, attr[1])) != RAJNI_OK) return rc;
    if constexpr (EPI == EPI_RESID)
      if ((rc = set_lds_attr(&wide::gemm_bf16_tn_stream<EPI, ALOAD, SF32, 2, 4, 8, NS, W8, 1>, lds, attr[3])) != RAJNI_OK) return rc;
    p.tiles_n = (p.N + 255) / 256;
    p.total_tiles = p.tiles_n * ((p.M + 255) / 256);
    p.nblk = n_block(p.tiles_n, (p.M + 255) / 256, 256, p.K, 2, cus);   // fp8 W: same blocks as bf16 (measured)
    const int grid = stream_grid(p.total_tiles, cus);
    if (EPI == EPI_RESID && kclass == KC_GEMM_RESID_SQ) {
      if constexpr (EPI == EPI_RESID)
        hipLaunchKernelGGL((wide::gemm_bf16_tn_stream<EPI, ALOAD, SF32, 2, 4, 8, NS, W8, 1>), dim3(grid), dim3(512), lds, s, p);
    } else {
      hipLaunchKernelGGL((wide::gemm_bf16_tn_stream<EPI, ALOAD, SF32, 2, 4, 8, NS, W8, 0>), dim3(grid), dim3(512), lds, s, p);
    }
  } else if (mode == 5) {
    using C = wide::Cfg<2, 3, W8>;
    // fp32-stream RESID: 2 KiB of LDS per wave behind the three stages for the epilogue's transpose (144 + 16 = 160 KiB)
    constexpr int lds = C::LDS_BYTES + (((nat_order(EPI, SF32) && EPI == EPI_RESID) || (RAJNI_TSTORE && (EPI == EPI_BIAS || EPI == EPI_GELU))) ? 8 * 2048 : 0);
    if ((rc = set_lds_attr(&wide::gemm_bf16_tn_stream<EPI, ALOAD, SF32, 4, 2, 4, 3, W8, 0>, lds, attr[2])) != RAJNI_OK) return rc;
    if constexpr (EPI == EPI_RESID)
      if ((rc = set_lds_attr(&wide::gemm_bf16_tn_stream<EPI, ALOAD, SF32, 4, 2, 4, 3, W8, 1>, lds, attr[4])) != RAJNI_OK) return rc;
    p.total_tiles = t256;
    p.nblk = n_block(p.tiles_n, (p.M + 255) / 256, 128, p.K, 2, cus);
    const int grid = stream_grid(p.total_tiles, cus);
    if (EPI == EPI_RESID && kclass == KC_GEMM_RESID_SQ) {
      if constexpr (EPI == EPI_RESID)
        hipLaunchKernelGGL((wide::gemm_bf16_tn_stream<EPI, ALOAD, SF32, 4, 2, 4, 3, W8, 1>), dim3(grid), dim3(512), lds, s, p);
    } else {
      hipLaunchKernelGGL((wide::gemm_bf16_tn_stream<EPI, ALOAD, SF32, 4, 2, 4, 3, W8, 0>), dim3(grid), dim3(512), lds, s, p);
    }
  } else {
    constexpr int lds = small::LDS_BYTES;
    if ((rc = set_lds_attr(&small::gemm_bf16_tn_128x128<EPI, ALOAD, SF32, W8>, lds, attr[0])) != RAJNI_OK) return rc;
    p.total_tiles = t128;
    hipLaunchKernelGGL((small::gemm_bf16_tn_128x128<EPI, ALOAD, SF32, W8>), dim3(t128), dim3(256), lds, s, p);
  }
  RAJNI_CHECK_LAUNCH("gemm_bf16_tn");
  return RAJNI_OK;
}

int g_force_f8_tiling = 0;   // test hook: 0 by shape, 1 = 256 x 128 always, 2 = 256 x 256 wherever it exists

// fp8 x fp8 launches (gemm_f8.h), persistent: 256 x 256 for wide outputs without a residual epilogue (QKV, FC1),
// 256 x 128 otherwise (FC2, small problems)
template <int EPI, bool SF32>
int launch_gemm_f8(GemmParams p, int kclass, bool tag_sq, hipStream_t s) {
  const int cus = rajni_num_cus();
  p.stamps = rajni_g_stamps;
  if constexpr (EPI == EPI_BIAS || EPI == EPI_GELU8) {
    const bool wide_ok = p.K >= 384;
    // measured on ViT-B at batch 256 (K = 768: 6 K steps per tile, so per-tile costs weigh double against the bf16
    // tilings): FC1 + GELU -> e4m3 132 us on 256 x 256 vs 139 us on 256 x 128.  QKV -> bf16: 256 x 256 since its
    // epilogue stores whole lines (tools/f8_qkv_tiling.py, us at 197 / 173 / 152 / 121 / 87 tokens: 120.5 / 118.8 / 93.7 /
    // 74.3 / 61.6 against 128.1 / 102.0 / 103.4 / 78.6 / 60.8 on 256 x 128; batch 512: 264.5 / 234.4 / 205.6 / 150.0 /
    // 112.3 against 288.3 / 251.6 / 220.1 / 155.8 / 100.4) - except where its tiles end just past a whole number of
    // rounds in a short launch (6.08 and 6.12 rounds above: the nearly empty last round costs 12-16 %)
    auto rounds_ok = [&]() {
      const double rounds = (double)((p.M + 255) / 256) * ((p.N + f8w::BN - 1) / f8w::BN) / cus;
      const double frac = rounds - (long)rounds;
      return !(rounds < 8.0 && frac > 0.0 && frac < 0.2);
    };
    const bool want = g_force_f8_tiling == 2 ||
                      (g_force_f8_tiling == 0 && p.M >= 1024 && p.N >= 1536 && (EPI == EPI_GELU8 || (RAJNI_F8W_LINES && rounds_ok())));
    if (wide_ok && want) {
      p.tiles_n = (p.N + f8w::BN - 1) / f8w::BN;
      const int tiles_m = (p.M + 255) / 256;
      p.total_tiles = p.tiles_n * tiles_m;
      p.nblk = n_block(p.tiles_n, tiles_m, f8w::BN, p.K, 1, cus);
      static bool attrw[RAJNI_MAX_DEVICES] = {};
      ProfScope prof(kclass, s, 2.0 * p.M * (double)p.N * p.K,
                     (double)p.M * p.K + (EPI == EPI_GELU8 ? 1.0 : 2.0) * (double)p.M * p.N + (double)p.N * p.K);
      int rc;
      // bf16 output: 4 KiB of LDS per wave behind the two stages for the epilogue's whole-line transpose (128 + 32 = 160 KiB)
      constexpr int ldsw = f8w::LDS_BYTES + ((EPI == EPI_BIAS && RAJNI_F8W_LINES) ? 8 * 4096 : 0);
      if ((rc = set_lds_attr(&f8w::gemm_f8_tn_wide<EPI>, ldsw, attrw)) != RAJNI_OK) return rc;
      hipLaunchKernelGGL((f8w::gemm_f8_tn_wide<EPI>), dim3(stream_grid(p.total_tiles, cus)), dim3(512), ldsw, s, p);
      RAJNI_CHECK_LAUNCH("gemm_f8_tn_wide");
      return RAJNI_OK;
    }
  }
  p.tiles_n = (p.N + f8::BN - 1) / f8::BN;
  const int tiles_m = (p.M + 255) / 256;
  p.total_tiles = p.tiles_n * tiles_m;
  p.nblk = n_block(p.tiles_n, tiles_m, f8::BN, p.K, 1, cus);
  static bool attr[2][RAJNI_MAX_DEVICES] = {};
  constexpr double ysz = EPI == EPI_GELU8 ? 1.0 : (SF32 && EPI == EPI_RESID) ? 4.0 : 2.0;
  constexpr double rsz = EPI == EPI_RESID ? (SF32 ? 4.0 : 2.0) : 0.0;
  ProfScope prof(kclass, s, 2.0 * p.M * (double)p.N * p.K,
                 (double)p.M * p.K + (ysz + rsz) * (double)p.M * p.N + (double)p.N * p.K);
  int rc;
  const int grid = stream_grid(p.total_tiles, cus);
  // fp32-stream RESID and bf16 outputs: 2 KiB of LDS per wave behind the three stages for the epilogue's transpose (144 + 16 KiB)
  constexpr int lds8 = f8::LDS_BYTES + (((nat_order(EPI, SF32) && EPI == EPI_RESID) || EPI == EPI_BIAS) ? 8 * 2048 : 0);
  if (EPI == EPI_RESID && tag_sq) {
    if constexpr (EPI == EPI_RESID) {
      if ((rc = set_lds_attr(&f8::gemm_f8_tn_stream<EPI, SF32, 1>, lds8, attr[1])) != RAJNI_OK) return rc;
      hipLaunchKernelGGL((f8::gemm_f8_tn_stream<EPI, SF32, 1>), dim3(grid), dim3(512), lds8, s, p);
    }
  } else {
    if ((rc = set_lds_attr(&f8::gemm_f8_tn_stream<EPI, SF32, 0>, lds8, attr[0])) != RAJNI_OK) return rc;
    hipLaunchKernelGGL((f8::gemm_f8_tn_stream<EPI, SF32, 0>), dim3(grid), dim3(512), lds8, s, p);
  }
  RAJNI_CHECK_LAUNCH("gemm_f8_tn");
  return RAJNI_OK;
}

}  // namespace

extern "C" void rajni_debug_force_gemm_tiling(int mode) { g_force_tiling = mode; }
extern "C" void rajni_debug_force_f8_tiling(int mode) { g_force_f8_tiling = mode; }
extern "C" void rajni_debug_set_resid_stagger(int units) { g_resid_stagger = units; }

extern "C" void rajni_debug_set_gemm_nblock_bytes(int bytes) { g_nblk_bytes = bytes; }
// diagnostic builds (-DRAJNI_GEMM_STAMPS): device buffer of 4 x u64 per workgroup, or NULL
extern "C" void rajni_debug_set_gemm_stamps(void* buf) { rajni_g_stamps = (unsigned long long*)buf; }

int launch_linear(const rajni_linear_args& a, hipStream_t s) {
  RAJNI_REQUIRE(a.dtype == RAJNI_BF16 || a.dtype == RAJNI_F32, RAJNI_ERR_INVALID, "rajni_linear: bad dtype %d", a.dtype);
  RAJNI_REQUIRE(a.x && a.w && a.y, RAJNI_ERR_INVALID, "rajni_linear: null pointer");
  RAJNI_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 64 == 0, RAJNI_ERR_INVALID,
                "rajni_linear: M,N>0 and K %% 64 == 0 required (M=%d N=%d K=%d)", a.M, a.N, a.K);
  RAJNI_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 8 == 0, RAJNI_ERR_INVALID,
                "rajni_linear: leading dimensions must be multiples of 8 elements");
  RAJNI_REQUIRE(((uintptr_t)a.x | (uintptr_t)a.w | (uintptr_t)a.y | (uintptr_t)a.resid) % 16 == 0,
                RAJNI_ERR_INVALID, "rajni_linear: pointers must be 16-byte aligned");
  GemmParams p{};
  p.X = a.x; p.lda = a.lda;
  p.W = a.w; p.ldw = a.ldw;
  p.bias = a.bias; p.gamma = a.gamma;
  p.R = a.resid; p.ldr = a.ldr;
  p.ridx = a.r_idx; p.r_np = a.r_np > 0 ? a.r_np : 1; p.r_nsrc = a.r_nsrc;
  p.Y = a.y; p.ldc = a.ldc;
  p.M = a.M; p.N = a.N; p.K = a.K;
  p.wscale = a.w_scale;
  if (a.x_scale != nullptr) {   // fp8 e4m3 activations AND weights on the fp8 matrix pipe
    RAJNI_REQUIRE(a.w_scale != nullptr && a.dtype == RAJNI_BF16, RAJNI_ERR_INVALID,
                  "rajni_linear: fp8 activations (x_scale) need fp8 weights (w_scale) and dtype bf16");
    RAJNI_REQUIRE(a.K % 256 == 0 && a.K >= 512, RAJNI_ERR_UNSUPPORTED,
                  "rajni_linear: the fp8 x fp8 kernel needs K %% 256 == 0 and K >= 512 (K=%d)", a.K);
    RAJNI_REQUIRE(a.lda % 16 == 0 && a.ldw % 16 == 0, RAJNI_ERR_INVALID, "rajni_linear: fp8 operands need lda, ldw %% 16 == 0 (bytes)");
    p.xscale = a.x_scale; p.yscale = a.y_scale;
    switch (a.epilogue) {
      case RAJNI_EPI_BIAS:
        RAJNI_REQUIRE(a.y_scale == nullptr, RAJNI_ERR_UNSUPPORTED, "rajni_linear: an fp8 output (y_scale) exists for the GELU epilogue only");
        return launch_gemm_f8<EPI_BIAS, false>(p, KC_GEMM8_BIAS, false, s);
      case RAJNI_EPI_BIAS_GELU:
        RAJNI_REQUIRE(a.y_scale != nullptr && a.ldc % 16 == 0, RAJNI_ERR_UNSUPPORTED,
                      "rajni_linear: the fp8 x fp8 GELU epilogue writes e4m3 (y_scale required, ldc %% 16 == 0 bytes)");
        return launch_gemm_f8<EPI_GELU8, false>(p, KC_GEMM8_GELU, false, s);
      case RAJNI_EPI_BIAS_RESID:
        RAJNI_REQUIRE(a.resid != nullptr && a.ldr % 8 == 0 && a.y_scale == nullptr, RAJNI_ERR_INVALID,
                      "rajni_linear: RESID epilogue needs resid, ldr %% 8 == 0 and no y_scale");
        return a.stream_f32 ? launch_gemm_f8<EPI_RESID, true>(p, a.K <= a.N ? KC_GEMM8_RESID_SQ : KC_GEMM8_RESID, a.K <= a.N, s)
                            : launch_gemm_f8<EPI_RESID, false>(p, a.K <= a.N ? KC_GEMM8_RESID_SQ : KC_GEMM8_RESID, a.K <= a.N, s);
      default:
        rajni_set_error("rajni_linear: unknown epilogue %d", a.epilogue);
        return RAJNI_ERR_INVALID;
    }
  }
  RAJNI_REQUIRE(a.y_scale == nullptr, RAJNI_ERR_INVALID, "rajni_linear: y_scale without x_scale");
  if (a.w_scale != nullptr) {   // fp8 e4m3 weights, bf16 activations
    RAJNI_REQUIRE(a.dtype == RAJNI_BF16, RAJNI_ERR_UNSUPPORTED, "rajni_linear: fp8 weights need bf16 activations");
    RAJNI_REQUIRE(a.ldw % 16 == 0, RAJNI_ERR_INVALID, "rajni_linear: fp8 weights need ldw %% 16 == 0");
    switch (a.epilogue) {
      case RAJNI_EPI_BIAS: return launch_gemm<EPI_BIAS, ALOAD_PLAIN, false, true>(p, KC_GEMM_BIAS, s);
      case RAJNI_EPI_BIAS_GELU: return launch_gemm<EPI_GELU, ALOAD_PLAIN, false, true>(p, KC_GEMM_GELU, s);
      case RAJNI_EPI_BIAS_RESID:
        RAJNI_REQUIRE(a.resid != nullptr && a.ldr % 8 == 0, RAJNI_ERR_INVALID,
                      "rajni_linear: RESID epilogue needs resid and ldr %% 8 == 0");
        return a.stream_f32 ? launch_gemm<EPI_RESID, ALOAD_PLAIN, true, true>(p, a.K <= a.N ? KC_GEMM_RESID_SQ : KC_GEMM_RESID, s)
                            : launch_gemm<EPI_RESID, ALOAD_PLAIN, false, true>(p, a.K <= a.N ? KC_GEMM_RESID_SQ : KC_GEMM_RESID, s);
      default:
        rajni_set_error("rajni_linear: unknown epilogue %d", a.epilogue);
        return RAJNI_ERR_INVALID;
    }
  }
  if (a.dtype == RAJNI_F32) {
    switch (a.epilogue) {
      case RAJNI_EPI_BIAS: return f32::launch<EPI_BIAS, ALOAD_PLAIN>(p, KC_GEMM_BIAS, s);
      case RAJNI_EPI_BIAS_GELU: return f32::launch<EPI_GELU, ALOAD_PLAIN>(p, KC_GEMM_GELU, s);
      case RAJNI_EPI_BIAS_RESID:
        RAJNI_REQUIRE(a.resid != nullptr, RAJNI_ERR_INVALID, "rajni_linear: RESID epilogue needs resid");
        return f32::launch<EPI_RESID, ALOAD_PLAIN>(p, a.K <= a.N ? KC_GEMM_RESID_SQ : KC_GEMM_RESID, s);
      default:
        rajni_set_error("rajni_linear: unknown epilogue %d", a.epilogue);
        return RAJNI_ERR_INVALID;
    }
  }
  switch (a.epilogue) {
    case RAJNI_EPI_BIAS: return launch_gemm<EPI_BIAS, ALOAD_PLAIN, false>(p, KC_GEMM_BIAS, s);
    case RAJNI_EPI_BIAS_GELU: return launch_gemm<EPI_GELU, ALOAD_PLAIN, false>(p, KC_GEMM_GELU, s);
    case RAJNI_EPI_BIAS_RESID:
      RAJNI_REQUIRE(a.resid != nullptr && a.ldr % 8 == 0, RAJNI_ERR_INVALID,
                    "rajni_linear: RESID epilogue needs resid and ldr %% 8 == 0");
      return a.stream_f32 ? launch_gemm<EPI_RESID, ALOAD_PLAIN, true>(p, a.K <= a.N ? KC_GEMM_RESID_SQ : KC_GEMM_RESID, s)
                          : launch_gemm<EPI_RESID, ALOAD_PLAIN, false>(p, a.K <= a.N ? KC_GEMM_RESID_SQ : KC_GEMM_RESID, s);
    default:
      rajni_set_error("rajni_linear: unknown epilogue %d", a.epilogue);
      return RAJNI_ERR_INVALID;
  }
}

// the fused im2col loader reads 16-byte runs of a patch row with shifts: power-of-two patches of >= 8 pixels
// whose K = Cin*P*P is whole K steps.  Everything else (P = 14: ViT-L/14, ViT-H/14, DINOv2, CLIP) goes through a
// materialised, zero-padded column matrix.
static bool patch_fused(int Cin, int S, int P) {
  return P >= 8 && (P & (P - 1)) == 0 && S % 8 == 0 && (Cin * P * P) % 64 == 0;
}
size_t patch_embed_workspace_bytes(int B, int Cin, int S, int P, int dtype) {
  if (P <= 0 || S <= 0 || S % P != 0 || patch_fused(Cin, S, P)) return 0;
  const size_t kpad = ((size_t)Cin * P * P + 63) / 64 * 64, gw = S / P;
  return (size_t)B * gw * gw * kpad * (dtype == RAJNI_F32 ? 4 : 2);
}
// cols[m, k] = images[b, c, gy*P + py, gx*P + px] for k = (c*P + py)*P + px < Cin*P*P, 0 for the padding;
// one thread = 8 consecutive k of one patch row (one 16-byte store for bf16)
template <typename T>
__global__ void __launch_bounds__(256) im2col_kernel(const T* img, T* cols, int Cin, int S, int P, int gw, int kpad, long total8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int k8 = kpad >> 3;
  const long m = i / k8;
  const int k0 = (int)(i - m * k8) * 8;
  const int npatch = gw * gw;
  const int b = (int)(m / npatch), pp = (int)(m - (long)b * npatch);
  const int gy = pp / gw, gx = pp - gy * gw;
  const int K = Cin * P * P;
  T v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    T e = T(0);
    if (k < K) {
      const int c = k / (P * P), r = k - c * P * P, py = r / P, px = r - py * P;
      e = img[(((long)b * Cin + c) * S + gy * P + py) * S + gx * P + px];
    }
    v[j] = e;
  }
  T* dst = cols + m * kpad + k0;
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = v[j];
}

int launch_patch_embed(const void* images, const void* w, const float* bias, const void* cls,
                       const void* pos, int pos_has_cls, void* x, int out_f32, int B, int Cin, int S,
                       int P, int C, int dtype, void* ws, size_t ws_bytes, hipStream_t s) {
  RAJNI_REQUIRE(images && w && cls && pos && x, RAJNI_ERR_INVALID, "rajni_patch_embed: null pointer");
  RAJNI_REQUIRE(P >= 1 && S % P == 0 && B > 0 && Cin > 0, RAJNI_ERR_INVALID,
                "rajni_patch_embed: the patch size must divide the image (P=%d S=%d)", P, S);
  RAJNI_REQUIRE(C % 8 == 0, RAJNI_ERR_UNSUPPORTED, "rajni_patch_embed: C %% 8 == 0 required");
  RAJNI_REQUIRE(dtype == RAJNI_BF16 || dtype == RAJNI_F32, RAJNI_ERR_INVALID, "rajni_patch_embed: bad dtype %d", dtype);
  const int K = Cin * P * P, kpad = (K + 63) / 64 * 64;
  const int gw = S / P, npatch = gw * gw;
  GemmParams p{};
  p.W = w; p.ldw = kpad;
  p.bias = bias;
  p.Y = x; p.ldc = C;
  p.M = B * npatch; p.N = C; p.K = kpad;
  p.gw = gw; p.npatch = npatch;
  p.pos = pos; p.pos_off = pos_has_cls ? 1 : 0;
  int rc;
  if (patch_fused(Cin, S, P)) {
    int log2ps = 0;
    while ((1 << log2ps) < P) ++log2ps;
    p.X = images; p.lda = 0;
    p.cin = Cin; p.S = S; p.log2ps = log2ps;
    if (dtype == RAJNI_F32) rc = f32::launch<EPI_PATCH, ALOAD_PATCH>(p, KC_GEMM_PATCH, s);
    else rc = out_f32 ? launch_gemm<EPI_PATCH, ALOAD_PATCH, true>(p, KC_GEMM_PATCH, s)
                      : launch_gemm<EPI_PATCH, ALOAD_PATCH, false>(p, KC_GEMM_PATCH, s);
  } else {
    const size_t need = patch_embed_workspace_bytes(B, Cin, S, P, dtype);
    RAJNI_REQUIRE(ws != nullptr && ws_bytes >= need && (uintptr_t)ws % 16 == 0, RAJNI_ERR_INVALID,
                  "rajni_patch_embed: patch size %d needs a %zu-byte, 16-byte aligned column workspace (got %zu)", P, need, ws_bytes);
    {
      ProfScope prof(KC_CLS_POS, s, 0.0, 2.0 * (double)need);
      const long total8 = (long)B * npatch * (kpad / 8);
      const dim3 grid((unsigned)((total8 + 255) / 256)), block(256);
      if (dtype == RAJNI_F32)
        hipLaunchKernelGGL(im2col_kernel<float>, grid, block, 0, s, (const float*)images, (float*)ws, Cin, S, P, gw, kpad, total8);
      else
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)images, (bf16_t*)ws, Cin, S, P, gw, kpad, total8);
      RAJNI_CHECK_LAUNCH("im2col_kernel");
    }
    p.X = ws; p.lda = kpad;
    if (dtype == RAJNI_F32) rc = f32::launch<EPI_PATCH, ALOAD_PLAIN>(p, KC_GEMM_PATCH, s);
    else rc = out_f32 ? launch_gemm<EPI_PATCH, ALOAD_PLAIN, true>(p, KC_GEMM_PATCH, s)
                      : launch_gemm<EPI_PATCH, ALOAD_PLAIN, false>(p, KC_GEMM_PATCH, s);
  }
  if (rc != RAJNI_OK) return rc;
  {
    ProfScope prof(KC_CLS_POS, s, 0.0, 6.0 * B * C);
    const int n = B * C;
    const dim3 grid((n + 255) / 256), block(256);
    const long stride = (long)(npatch + 1) * C;
    if (dtype == RAJNI_F32)
      hipLaunchKernelGGL((cls_pos_kernel<true, float>), grid, block, 0, s, (const float*)cls, (const float*)pos,
                         pos_has_cls, x, stride, B, C);
    else if (out_f32)
      hipLaunchKernelGGL((cls_pos_kernel<true, bf16_t>), grid, block, 0, s, (const bf16_t*)cls, (const bf16_t*)pos,
                         pos_has_cls, x, stride, B, C);
    else
      hipLaunchKernelGGL((cls_pos_kernel<false, bf16_t>), grid, block, 0, s, (const bf16_t*)cls, (const bf16_t*)pos,
                         pos_has_cls, x, stride, B, C);
    RAJNI_CHECK_LAUNCH("cls_pos_kernel");
  }
  return RAJNI_OK;
}
