// Fused softmax attention on the packed (kept) tokens (reference attention.py:42-54):
// tuned kernels for head dim 64 (this header describes them), a general MFMA kernel for every other
// head dim % 8 == 0 up to 128 (attn_bf16_dgen), VALU kernels for fp32 models, and the CLS-row kernel.  The keep_idx row gather is fused into the Q/K/V tile loads: nothing of the
// reference's gathered qkv copy or its [B,H,Np,Np] attention matrix ever reaches HBM.
//
// Work split: grid (ceil(Np/128), H, B); a workgroup = 4 waves, each wave owns 32 query rows.
// K/V of the (image, head) are staged through LDS in chunks of 128 keys (16 KiB + 16 KiB, swizzled
// for conflict-free reads); online softmax over 32-key sub-blocks in fp32.
//
// MFMA orientation ("key/feature on the register axis, query on the lane"):
//   S^T[key][q] = K Q^T        v_mfma_f32_32x32x16_bf16(A = K rows from LDS, B = Q rows from registers)
//   O^T[d][q]  += V^T P^T      A = V^T via ds_read_b64_tr_b16 (hardware transpose of row-major V),
//                              B = the S^T accumulator itself, exponentiated and packed to bf16 -
//                              an accumulator tile is directly the next MFMA's B operand, so P never
//                              goes through LDS or lane shuffles.
// Every per-query quantity (running max, sum, rescale factor, 1/l) is per-lane.
#include "common.h"

namespace {

constexpr int AT_THREADS = 256;
constexpr int AT_QROWS = 128;    // query rows per workgroup
constexpr int KV_CHUNK = 128;    // keys per LDS chunk
constexpr int AT_LDS = 2 * KV_CHUNK * 128;  // K + V tiles, 128-byte rows

struct AttnArgs {
  const bf16_t* qkv;
  const int* idx;   // [B,np] or null
  bf16_t* out;      // [B,np,H*64]
  int n_src, np, H;
  float c;          // scale * log2(e)
  unsigned long long* stamps;   // diagnostic builds (-DRAJNI_ATTN_STAMPS) only
  // fp8 output (rajni_attention_fp8; the O8 instantiations): rows as e4m3_rne_sat(o * oinv) bytes [B,np,H*64], and the
  // dequantisation scale 1 / oinv written to row_scale[b*np + q] (by the head-0 items) for the fp8 proj launch
  unsigned char* out8;
  float oinv, oscale;
  float* row_scale;
};
__device__ __forceinline__ unsigned attn_pack4_e4m3(float a, float b, float c, float d) {   // saturating RNE (gemm_f8.h)
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (unsigned)r;
}

__device__ __forceinline__ int k_off(int row, int chunk) {   // K tile: natural row reads
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ int v_off(int row, int chunk) {   // V tile: 4-row x 64-byte tr blocks
  return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4);
}

__device__ __forceinline__ bf16x8 tr_pair(const char* sv, int row, int col) {
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int o0 = v_off(row, col >> 3) + (col & 7) * 2;
  const int o1 = v_off(row + 8, col >> 3) + (col & 7) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sv + o0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(sv + o1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ void __launch_bounds__(AT_THREADS) attn_bf16_d64(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[AT_LDS];
  char* sk = smem;
  char* sv = smem + KV_CHUNK * 128;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5, g = lane >> 4, l15 = lane & 15;
  const int head = blockIdx.y, b = blockIdx.z;
  const int np = a.np, C = a.H * 64, C3 = 3 * C;
  const bf16_t* img = a.qkv + (long)b * a.n_src * C3;
  const int* idx = a.idx ? a.idx + (long)b * np : nullptr;

  const int qbase = blockIdx.x * AT_QROWS + wave * 32;
  const bool active = qbase < np;

  // ---- Q fragments straight from HBM (B operand: lane holds Q[q = l31][d = 16s + 8h .. +7])
  bf16x8 qf[4];
  {
    int q = qbase + l31;
    if (q > np - 1) q = np - 1;
    const int srow = idx ? idx[q] : q;
    const bf16_t* qp = img + (long)srow * C3 + head * 64 + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  const int st_c = tid & 7, st_r = tid >> 3;  // staging: 8 threads per 128-byte row, 32 rows per pass

  for (int c0 = 0; c0 < np; c0 += KV_CHUNK) {
    // ---- stage K and V rows c0 .. c0+127 (gathered through keep_idx), zero-fill past np
    uint4 kreg[4], vreg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = c0 + st_r + 32 * i;
      if (t < np) {
        const int srow = idx ? idx[t] : t;
        const bf16_t* rp = img + (long)srow * C3 + head * 64 + st_c * 8;
        kreg[i] = *reinterpret_cast<const uint4*>(rp + C);
        vreg[i] = *reinterpret_cast<const uint4*>(rp + 2 * C);
      } else {
        kreg[i] = make_uint4(0, 0, 0, 0);
        vreg[i] = make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();  // previous chunk fully consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = st_r + 32 * i;
      *reinterpret_cast<uint4*>(sk + k_off(row, st_c)) = kreg[i];
      *reinterpret_cast<uint4*>(sv + v_off(row, st_c)) = vreg[i];
    }
    __syncthreads();

    if (active) {
      const int nsub = (np - c0 + 31) >> 5;
      for (int kb = 0; kb < 4 && kb < nsub; ++kb) {
        // ---- S^T tile: 32 keys x 32 queries
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const int krow = kb * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + k_off(krow, 2 * ks + h));
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        const int nvalid = np - c0 - kb * 32;  // keys of this sub-block that exist
        if (nvalid < 32) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kr = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (kr >= nvalid) s[r] = -INFINITY;
          }
        }
        // ---- online softmax; lane = query, registers (+ the other half-wave) = keys
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // Lazy rescale: the running reference m_run only moves when some query's block maximum exceeds it
        // by more than 8 (in log2 units), so p = exp2(s c - m_run) <= 2^8 - exact in fp32, harmless in the
        // bf16 P operand (same relative precision) - and the rescale of O (32 multiplies + an exp per key
        // block) runs for the first block or two only.  One wave-uniform branch per block.
        const float mxc = mx * a.c;
        if (__builtin_amdgcn_ballot_w64(mxc > m_run + 8.0f) != 0) {
          const float m_new = fmaxf(m_run, mxc);
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first block: exp2(-inf) = 0
          l_run *= alpha;
          m_run = m_new;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        float p[16];
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], a.c, -m_run));
          lsum += p[r];
        }
        l_run += lsum;
        // ---- P^T (bf16) as the B operand: registers 8s .. 8s+7 feed k-step s; element j of half h
        //      is key 16s + 8(j>>2) + 4h + (j&3) - V^T below is read in that same key order
        bf16x8 pb[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
          const u32x4 w = {pack2bf(p[8 * s2 + 0], p[8 * s2 + 1]), pack2bf(p[8 * s2 + 2], p[8 * s2 + 3]),
                           pack2bf(p[8 * s2 + 4], p[8 * s2 + 5]), pack2bf(p[8 * s2 + 6], p[8 * s2 + 7])};
          pb[s2] = __builtin_bit_cast(bf16x8, w);
        }
        // ---- O^T += V^T P^T ; lane (d = l31 of the d-tile, half h) needs keys 16s+4h+{0..3} and +8
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int key0 = kb * 32 + 16 * s2 + 4 * h + (l15 >> 2);
          const int col = 16 * (g & 1) + 4 * (l15 & 3);
          const bf16x8 v0 = tr_pair(sv, key0, col);
          const bf16x8 v1 = tr_pair(sv, key0, 32 + col);
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb[s2], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb[s2], o1, 0, 0, 0);
        }
      }
    }
  }

  if (!active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = qbase + l31;
  if (q < np) {
    bf16_t* op = a.out + ((long)b * np + q) * C + head * 64 + 4 * h;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // registers 4t..4t+3 of a d-tile are d = 8t + 4h + {0..3}
      uint2 w0, w1;
      w0.x = pack2bf(o0[4 * t] * inv, o0[4 * t + 1] * inv);
      w0.y = pack2bf(o0[4 * t + 2] * inv, o0[4 * t + 3] * inv);
      w1.x = pack2bf(o1[4 * t] * inv, o1[4 * t + 1] * inv);
      w1.y = pack2bf(o1[4 * t + 2] * inv, o1[4 * t + 3] * inv);
      *reinterpret_cast<uint2*>(op + 8 * t) = w0;
      *reinterpret_cast<uint2*>(op + 32 + 8 * t) = w1;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Np <= 256 (every ViT/DeiT at 224 px, and the late stages of larger models): one 8-wave workgroup
// per (image, head) stages ALL kept K/V rows once (<= 64 KiB), each wave owns 32 query rows and
// keeps the whole S^T row block in registers (NSUB x 16 fp32): exact softmax with ONE max per
// query - no running max, no rescale of O, no alpha exp - i.e. about half the VALU work of the
// online form, which is what bounds attention at these sizes (16 exp + ~110 VALU vs 8 MFMAs per
// 32-key block).  NSUB = ceil(Np/32) is a template parameter so S stays in registers.
// ---------------------------------------------------------------------------------------------
constexpr int ATF_THREADS = 512;
#ifndef RAJNI_ATTN_STAGE_MASK
#define RAJNI_ATTN_STAGE_MASK 0x78   // bit NSUB-1: stage the output tile through LDS (measured: helps NSUB 4..7, not 1..3; 8 has no LDS left)
#endif
constexpr bool stage_o(int nsub) { return (RAJNI_ATTN_STAGE_MASK >> (nsub - 1)) & 1; }

// exact-softmax attention of ONE 32-query tile against all NSUB*32 staged keys (shared by the
// one-shot and the persistent kernels).  sk/sv: swizzled K / V images in LDS; qf: Q fragments.
// so != NULL: a 4 KiB per-wave LDS staging area; the 32 x 64 output tile is written there and leaves as
// whole 128-byte rows (four 16-byte stores per lane-row group).  Stored straight from the accumulator
// layout a lane owns 8 bytes of 8 different rows per instruction: 32 partial cache lines each, which backed
// up the vector-memory queue - the NEXT item's prefetch then stalled at issue (tools/attn_stamps.py:
// prefetch issue 2.5k + stores 1.7k of 11k cycles per item).
template <int NSUB, bool O8 = false>
__device__ __forceinline__ void attn_tile_compute(const char* sk, const char* sv, const bf16x8 (&qf)[4],
                                                  const AttnArgs& a, int b, int head, int qbase, int lane,
                                                  char* so = nullptr) {
  const int l31 = lane & 31, h = lane >> 5, g = lane >> 4, l15 = lane & 15;
  const int np = a.np, C = a.H * 64;
  // ---- S^T = K Q^T for every 32-key block
  f32x16 s[NSUB];
#pragma unroll
  for (int kb = 0; kb < NSUB; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    const int krow = kb * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + k_off(krow, 2 * ks + h));
      s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
    }
  }
#ifdef RAJNI_ATTN_STAMPS
  asm volatile("" :: "v"(s[0][0]), "v"(s[NSUB - 1][15]));
  const unsigned long long tsS = __builtin_amdgcn_s_memtime();
#endif
  // mask the keys past np (only in the last block)
  {
    const int nvalid = np - (NSUB - 1) * 32;
    if (nvalid < 32) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (kr >= nvalid) s[NSUB - 1][r] = -INFINITY;
      }
    }
  }
  // ---- exact softmax: one max per query (lane) over all registers and the other half-wave
  float mx = s[0][0];
#pragma unroll
  for (int kb = 0; kb < NSUB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float mc = mx * a.c;
  float lsum = 0.f;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#pragma unroll
  for (int kb = 0; kb < NSUB; ++kb) {
    float p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], a.c, -mc));
      lsum += p[r];
    }
    bf16x8 pb[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const u32x4 w = {pack2bf(p[8 * s2 + 0], p[8 * s2 + 1]), pack2bf(p[8 * s2 + 2], p[8 * s2 + 3]),
                       pack2bf(p[8 * s2 + 4], p[8 * s2 + 5]), pack2bf(p[8 * s2 + 6], p[8 * s2 + 7])};
      pb[s2] = __builtin_bit_cast(bf16x8, w);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int key0 = kb * 32 + 16 * s2 + 4 * h + (l15 >> 2);
      const int col = 16 * (g & 1) + 4 * (l15 & 3);
      const bf16x8 v0 = tr_pair(sv, key0, col);
      const bf16x8 v1 = tr_pair(sv, key0, 32 + col);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb[s2], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb[s2], o1, 0, 0, 0);
    }
  }
#ifdef RAJNI_ATTN_STAMPS
  asm volatile("" :: "v"(o0[0]), "v"(o1[15]));
  const unsigned long long tsP = __builtin_amdgcn_s_memtime();
  if (a.stamps != nullptr && qbase == 0 && lane == 0) {
    unsigned long long* o = a.stamps + ((size_t)b * a.H + head) * 8;
    o[1] = tsS; o[2] = tsP;
  }
#endif
  const float inv = 1.0f / (lsum + __shfl_xor(lsum, 32, 64));
  const int q = qbase + l31;
  // Everything this wave has in flight (the next item's K/V DMA, its Q and index loads) was issued before
  // this tile's math and has landed by now: wait for it HERE, before the output stores, so that the
  // per-item barrier of the persistent kernel does not have to wait with vmcnt(0) - gfx9 counts stores in
  // vmcnt too, and a wait after them exposes the whole write latency once per item.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  if constexpr (O8) {
    // e4m3 output (so is never NULL here): the 32 x 64-byte tile through 2 KiB of the wave's staging area - the lane's
    // dwords d = 8 t + 4 h .. + 3 of both 32-column halves, 16-byte chunk c of row r at ((c ^ ((r >> 2) & 3)) << 4) -
    // and out as 16 rows x 64 bytes per instruction (a token's head slice is half a line: nothing wider exists)
    const float sc = inv * a.oinv;
    const int key = (l31 >> 2) & 3;
    char* wr = so + l31 * 64 + 4 * (h & 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned w0 = attn_pack4_e4m3(o0[4 * t] * sc, o0[4 * t + 1] * sc, o0[4 * t + 2] * sc, o0[4 * t + 3] * sc);
      const unsigned w1 = attn_pack4_e4m3(o1[4 * t] * sc, o1[4 * t + 1] * sc, o1[4 * t + 2] * sc, o1[4 * t + 3] * sc);
      // byte 8 t + 4 h of the half: chunk t >> 1, dword 2 (t & 1) + h inside it
      *reinterpret_cast<unsigned*>(wr + (((t >> 1) ^ key) << 4) + 8 * (t & 1)) = w0;
      *reinterpret_cast<unsigned*>(wr + (((2 + (t >> 1)) ^ key) << 4) + 8 * (t & 1)) = w1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int u = lane & 3;
    unsigned char* ob = a.out8 + ((long)b * np + qbase) * C + head * 64 + u * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (lane >> 2) + 16 * j;
      const uint4 v = *reinterpret_cast<const uint4*>(so + r * 64 + ((u ^ ((r >> 2) & 3)) << 4));
      if (qbase + r < np) *reinterpret_cast<uint4*>(ob + (long)r * C) = v;
    }
    if (head == 0 && h == 0 && q < np) a.row_scale[(long)b * np + q] = a.oscale;
    return;
  }
  if (so != nullptr) {
    // staging image: row r = query, sixteen 8-byte units per row, 16-byte unit u at ((u ^ key(r)) << 4)
    const int key = (l31 ^ (l31 >> 3)) & 7;
    char* wr = so + l31 * 128 + 8 * h;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint2 w0, w1;
      w0.x = pack2bf(o0[4 * t] * inv, o0[4 * t + 1] * inv);
      w0.y = pack2bf(o0[4 * t + 2] * inv, o0[4 * t + 3] * inv);
      w1.x = pack2bf(o1[4 * t] * inv, o1[4 * t + 1] * inv);
      w1.y = pack2bf(o1[4 * t + 2] * inv, o1[4 * t + 3] * inv);
      *reinterpret_cast<uint2*>(wr + ((t ^ key) << 4)) = w0;          // d = 8t + 4h .. +3
      *reinterpret_cast<uint2*>(wr + (((4 + t) ^ key) << 4)) = w1;    // d = 32 + 8t + 4h .. +3
    }
    // same wave wrote and reads: only the LDS counter orders them
    const int u = lane & 7;
    bf16_t* ob = a.out + ((long)b * np + qbase) * C + head * 64 + u * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (lane >> 3) + 8 * j;
      const uint4 v = *reinterpret_cast<const uint4*>(so + r * 128 + ((u ^ ((r ^ (r >> 3)) & 7)) << 4));
      if (qbase + r < np) *reinterpret_cast<uint4*>(ob + (long)r * C) = v;
    }
    return;
  }
  if (q < np) {
    bf16_t* op = a.out + ((long)b * np + q) * C + head * 64 + 4 * h;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint2 w0, w1;
      w0.x = pack2bf(o0[4 * t] * inv, o0[4 * t + 1] * inv);
      w0.y = pack2bf(o0[4 * t + 2] * inv, o0[4 * t + 3] * inv);
      w1.x = pack2bf(o1[4 * t] * inv, o1[4 * t + 1] * inv);
      w1.y = pack2bf(o1[4 * t + 2] * inv, o1[4 * t + 3] * inv);
      *reinterpret_cast<uint2*>(op + 8 * t) = w0;
      *reinterpret_cast<uint2*>(op + 32 + 8 * t) = w1;
    }
  }
}

template <int NSUB>
__global__ void __launch_bounds__(ATF_THREADS, 2) attn_bf16_d64_full(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = NSUB * 32;
  char* sk = smem;
  char* sv = smem + ROWS * 128;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5, g = lane >> 4, l15 = lane & 15;
  const int head = blockIdx.x, b = blockIdx.y;
  const int np = a.np, C = a.H * 64, C3 = 3 * C;
  const bf16_t* img = a.qkv + (long)b * a.n_src * C3;
  const int* idx = a.idx ? a.idx + (long)b * np : nullptr;
  const int qbase = wave * 32;
  const bool active = qbase < np;

  // ---- Q fragments from HBM while K/V are staged
  bf16x8 qf[4];
  {
    int q = qbase + l31;
    if (q > np - 1) q = np - 1;
    const int srow = idx ? idx[q] : q;
    const bf16_t* qp = img + (long)srow * C3 + head * 64 + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }
  // ---- stage all K and V rows (gathered through keep_idx), zero-fill up to ROWS
  {
    const int st_c = tid & 7, st_r = tid >> 3;  // 64 rows per pass
#pragma unroll
    for (int i = 0; i < (ROWS + 63) / 64; ++i) {
      const int t = st_r + 64 * i;
      if (t < ROWS) {
        uint4 kr = make_uint4(0, 0, 0, 0), vr = make_uint4(0, 0, 0, 0);
        if (t < np) {
          const int srow = idx ? idx[t] : t;
          const bf16_t* rp = img + (long)srow * C3 + head * 64 + st_c * 8;
          kr = *reinterpret_cast<const uint4*>(rp + C);
          vr = *reinterpret_cast<const uint4*>(rp + 2 * C);
        }
        *reinterpret_cast<uint4*>(sk + k_off(t, st_c)) = kr;
        *reinterpret_cast<uint4*>(sv + v_off(t, st_c)) = vr;
      }
    }
  }
  __syncthreads();
  if (!active) return;
  attn_tile_compute<NSUB>(sk, sv, qf, a, b, head, qbase, lane);
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the kernel above: workgroups walk (image, head) items; K/V of item i+1 are
// LDS-DMA'd (row gather through keep_idx on the per-lane SOURCE address, swizzle folded into it)
// into the other LDS buffer while item i is computed; Q fragments and the keep_idx entries are
// prefetched one / two items ahead.  One barrier per item.  Keys past np re-load row np-1 (finite)
// and are masked in the softmax, so no zero fill is needed.
// ---------------------------------------------------------------------------------------------
// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB at the wave-uniform LDS address `dst`) issued from inline
// asm so that hipcc does NOT see it: with the builtin, hipcc (ROCm 7.2) drains the DMA (vmcnt(0)) before
// every ds_read_b64_tr_b16 (possible alias) and before any use of an ordinary global load issued near it,
// which serialised K/V prefetch with compute (tools/attn_stamps.py: 9.8k of 16.4k cycles per item).  Its
// completion is counted by hand: s_waitcnt vmcnt(0) before the per-item barrier.  M0 is saved/restored
// inside the statement (M0 is compiler-reserved); recipe: cdna_hip_programming.md section 5.7.
// Scalar base + 32-bit lane byte offset: nothing but one v_add per piece on the vector side.
__device__ __forceinline__ void dma16(const char* base, unsigned off, char* dst) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(dst));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(off), "s"(lds), "s"(base) : "memory");
}

// G: rows are gathered through keep_idx (a.idx != NULL) - a template flag, so that the per-item prefetch
// carries no pointer tests (they were 10 scalar branches per item).
template <int NSUB, bool G, bool O8 = false>
__global__ void __launch_bounds__(ATF_THREADS, 2) attn_bf16_d64_stream(const AttnArgs a, int n_items) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = NSUB * 32, BUF = ROWS * 256, NP8 = ROWS / 8;   // pieces (8 rows) per operand
  constexpr bool STAGE_O = O8 || stage_o(NSUB);      // (e4m3 output always leaves through its 2 KiB-per-wave staging area)
  constexpr int STAGE_BYTES = O8 ? 2048 : 4096;
  constexpr int PER_WAVE = (NP8 + 7) / 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int np = a.np, C = a.H * 64, C3 = 3 * C;
  const int qbase = wave * 32;
  const bool active = qbase < np;
  const int r_in = lane >> 3, pos = lane & 7;

  // token (packed index) of the rows this lane stages, and of its query row
  int trow[PER_WAVE];
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int t = (wave + 8 * i) * 8 + r_in;
    trow[i] = t < np ? t : np - 1;
  }
  const int tq = qbase + l31 < np ? qbase + l31 : np - 1;

  // swizzled 16-byte chunk of this lane inside its K / V row.  Piece q = wave + 8 i covers rows 8 q + r_in:
  // (row >> 1) & 7 = (4 (q & 1) + (r_in >> 1)) & 7 and q & 1 = wave & 1, so both are per-lane constants.
  const unsigned ck = (unsigned)(pos ^ ((4 * (wave & 1) + (r_in >> 1)) & 7)) << 4;
  const unsigned cv = (unsigned)(pos ^ (((r_in >> 1) & 1) << 2)) << 4;
  const unsigned row_bytes = (unsigned)C3 * 2;

  auto load_rows = [&](int b, int (&srow)[PER_WAVE], int& sq) {   // keep_idx entries of image b
    if constexpr (G) {
      const int* idx = a.idx + (long)b * np;
#pragma unroll
      for (int i = 0; i < PER_WAVE; ++i) srow[i] = idx[trow[i]];
      sq = idx[tq];
    } else {
#pragma unroll
      for (int i = 0; i < PER_WAVE; ++i) srow[i] = trow[i];
      sq = tq;
    }
  };
  auto dma_item = [&](int b, int head, const int (&srow)[PER_WAVE], int buf) {
    const char* bk = reinterpret_cast<const char*>(a.qkv) + ((size_t)b * a.n_src * C3 + head * 64 + C) * 2;
    const char* bv = bk + (size_t)C * 2;
    char* sk = smem + buf * BUF;
    char* sv = sk + ROWS * 128;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int q = wave + 8 * i;            // piece index, wave uniform
      if (q < NP8) {
        const unsigned ro = (unsigned)srow[i] * row_bytes;
        dma16(bk, ro + ck, sk + q * 1024);
        dma16(bv, ro + cv, sv + q * 1024);
      }
    }
  };
  auto load_q = [&](int b, int head, int sq, bf16x8 (&qf)[4]) {
    const char* bq = reinterpret_cast<const char*>(a.qkv) + ((size_t)b * a.n_src * C3 + head * 64) * 2;
    const unsigned qo = (unsigned)sq * row_bytes + 16 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(bq + (size_t)(qo + 32 * s));
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  // Software pipeline (i = item being computed):  K/V(i+1) by asm DMA, Q(i+1) and keep_idx(i+2) by
  // ordinary loads, all issued right after the per-item barrier and left in flight during compute(i).
  // hipcc must not wait for any of them inside compute: every ordinary-load result is "touched" by an
  // empty asm right after the barrier, where the queue has just been drained by hand, so that is where
  // hipcc places its (then free) waits.
  int srow_a[PER_WAVE], sq_a;     // keep_idx entries of item i+1 (then i+2)
  bf16x8 qf[4], qn[4];
  int cb = item / a.H, chead = item - cb * a.H;   // (image, head) of the item being computed
  load_rows(cb, srow_a, sq_a);
  dma_item(cb, chead, srow_a, 0);
  load_q(cb, chead, sq_a, qn);
  int nxt = item + gridDim.x;
  load_rows((nxt < n_items ? nxt : n_items - 1) / a.H, srow_a, sq_a);
  int buf = 0;
  // the builtin (not asm) on purpose: hipcc's waitcnt pass must know that nothing is pending at the loop
  // header, or it waits with vmcnt(0) for the prefetched Q registers after the barrier - behind the stores
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): K/V, Q and indices of the first item
  while (true) {
    // my DMA pieces landed (waited before the previous item's output stores; before the loop for the first
    // item; just below for a wave without queries), my LDS reads are done
    if (!active) __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef RAJNI_ATTN_STAMPS
    const unsigned long long ts_arrive = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();                                  // ... everyone's: buffer `buf` is complete
#ifdef RAJNI_ATTN_STAMPS
    const unsigned long long ts_bar = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) { asm volatile("" : "+v"(qn[s])); qf[s] = qn[s]; }
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) asm volatile("" : "+v"(srow_a[i]));
    asm volatile("" : "+v"(sq_a));
    // prefetch for item i+1 / i+2, UNCONDITIONALLY (clamped to the last item: harmless re-loads that
    // nobody reads) - a conditional load would need a register copy, and hipcc would wait for it here
    const int last = n_items - 1;
    const int nx = nxt < last ? nxt : last;
    const int nn = nxt + (int)gridDim.x < last ? nxt + (int)gridDim.x : last;
    int srow_use[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) srow_use[i] = srow_a[i];
    const int nb = nx / a.H, nhead = nx - nb * a.H;
    load_q(nb, nhead, sq_a, qn);               // Q of item i+1
    load_rows(nn / a.H, srow_a, sq_a);         // keep_idx entries of item i+2
    dma_item(nb, nhead, srow_use, buf ^ 1);    // K/V of item i+1 -> the other buffer
#ifdef RAJNI_ATTN_STAMPS
    const unsigned long long ts_issue = __builtin_amdgcn_s_memtime();
#endif
    if (active) {
      const int b = cb, head = chead;
      attn_tile_compute<NSUB, O8>(smem + buf * BUF, smem + buf * BUF + ROWS * 128, qf, a, b, head, qbase, lane,
                                  STAGE_O ? smem + 2 * BUF + (qbase >> 5) * STAGE_BYTES : nullptr);
    }
#ifdef RAJNI_ATTN_STAMPS
    if (a.stamps != nullptr && qbase == 0 && lane == 0) {   // wave 0: [0] barrier arrival, [3] released, [4] prefetch issued, [5] tile done
      unsigned long long* o = a.stamps + (size_t)item * 8;
      o[0] = ts_arrive; o[3] = ts_bar; o[4] = ts_issue; o[5] = __builtin_amdgcn_s_memtime();
    }
#endif
    if (nxt >= n_items) break;
    item = nxt;
    cb = nb; chead = nhead;                    // nx == nxt here
    nxt += gridDim.x;
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing (unused) DMA before LDS is released
}

// ---------------------------------------------------------------------------------------------
// fp32 model path (accuracy path: "logits within 1e-3 of the reference's fp32 run").  No MFMA: four
// lanes share a query row (16 of the 64 head dims each), keys are staged in 64-key fp32 LDS chunks
// (read as broadcasts), online softmax per key in fp32.  ~50x the bf16 kernel's time, by design simple.
// ---------------------------------------------------------------------------------------------
struct AttnArgsF32 {
  const float* qkv; const int* idx; float* out;
  int n_src, np, H;
  float c;
};

__global__ void __launch_bounds__(256) attn_f32_d64(const AttnArgsF32 a) {
  __shared__ __attribute__((aligned(16))) float sk[64 * 64];
  __shared__ __attribute__((aligned(16))) float sv[64 * 64];
  const int tid = threadIdx.x;
  const int qrow = tid >> 2, pt = tid & 3;           // 64 query rows x 4 lanes
  const int head = blockIdx.y, b = blockIdx.z;
  const int np = a.np, C = a.H * 64, C3 = 3 * C;
  const float* img = a.qkv + (long)b * a.n_src * C3;
  const int* idx = a.idx ? a.idx + (long)b * np : nullptr;
  int q = blockIdx.x * 64 + qrow;
  const bool valid = q < np;
  if (!valid) q = np - 1;
  float qv[16], o[16];
  {
    const int srow = idx ? idx[q] : q;
    const float* qp = img + (long)srow * C3 + head * 64 + pt * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = reinterpret_cast<const float4*>(qp)[j];
      qv[4 * j] = t.x; qv[4 * j + 1] = t.y; qv[4 * j + 2] = t.z; qv[4 * j + 3] = t.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) o[j] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  for (int c0 = 0; c0 < np; c0 += 64) {
    __syncthreads();
    // stage 64 keys x 64 dims of K and V: 1024 float4 each, 4 per thread
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;           // float4 index
      const int r = e >> 4, c4 = e & 15;
      const int t = c0 + r;
      float4 kq = make_float4(0, 0, 0, 0), vq = make_float4(0, 0, 0, 0);
      if (t < np) {
        const int srow = idx ? idx[t] : t;
        const float* rp = img + (long)srow * C3 + head * 64 + c4 * 4;
        kq = *reinterpret_cast<const float4*>(rp + C);
        vq = *reinterpret_cast<const float4*>(rp + 2 * C);
      }
      reinterpret_cast<float4*>(sk)[e] = kq;
      reinterpret_cast<float4*>(sv)[e] = vq;
    }
    __syncthreads();
    const int nk = np - c0 < 64 ? np - c0 : 64;
    for (int j = 0; j < nk; ++j) {
      const float* kr = sk + j * 64 + pt * 16;
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) d = fmaf(qv[e], kr[e], d);
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      const float sc = d * a.c;
      const float m_new = fmaxf(m_run, sc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      const float pj = __builtin_amdgcn_exp2f(sc - m_new);
      l_run = fmaf(l_run, alpha, pj);
      m_run = m_new;
      const float* vr = sv + j * 64 + pt * 16;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = fmaf(o[e], alpha, pj * vr[e]);
    }
  }
  if (valid) {
    const float inv = 1.0f / l_run;
    float* op = a.out + ((long)b * np + q) * C + head * 64 + pt * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      reinterpret_cast<float4*>(op)[j] = make_float4(o[4 * j] * inv, o[4 * j + 1] * inv, o[4 * j + 2] * inv, o[4 * j + 3] * inv);
  }
}


// ---------------------------------------------------------------------------------------------
// Any head dim D with D % 8 == 0, D <= 128 (ViT-H 80, ViT-g 88, EVA/SigLIP 72...128, small heads 32/48): the
// general path beside the tuned D = 64 kernels above.  Same orientation - S^T = K Q^T, O^T += V^T P^T, the
// exponentiated S^T accumulator is the next MFMA's B operand - on v_mfma_f32_16x16x16_bf16, whose contraction
// step of 16 fits every such D after zero-padding to DP = 16 * ceil(D / 16) in LDS.  Workgroup = 4 waves x 32
// query rows of one (image, head); keys in chunks of 64: K row-major, V transposed, both padded to
// conflict-free strides; online softmax per chunk in fp32 (per-query state is per-lane, replicated over
// the four lane groups).  Operand layouts (lane l: r = l % 16, g = l / 16, j = 0..3):
//   A[i = r][k = 4g + j]   B[k = 4g + j][n = r]   C[i = 4g + j][n = r]
// ---------------------------------------------------------------------------------------------
constexpr int AG_KSTRIDE = 136;   // bf16 per K row in LDS (128 + 8): 16 rows x 2 lane groups hit 64 distinct banks
constexpr int AG_VSTRIDE = 68;    // bf16 per V^T row (64 keys + 4)
constexpr int AG_QROWS = 128;     // query rows per workgroup: 4 waves x 2 blocks of 16
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <int NW, int NB>   // waves per workgroup, blocks of 16 query rows per wave
__global__ void __launch_bounds__(NW * 64, 2) attn_bf16_dgen(const AttnArgs a, int D) {
  constexpr int NT = NW * 64, QROWS = NW * NB * 16, KI = 1024 / NT;   // K staging items per thread (<= 1024 per chunk)
  __shared__ __attribute__((aligned(16))) bf16_t sk[64 * AG_KSTRIDE];
  __shared__ __attribute__((aligned(16))) bf16_t svt[128 * AG_VSTRIDE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const int np = a.np, C = a.H * D, C3 = 3 * C;
  const int ND = (D + 15) >> 4;                 // 16-wide d blocks (last one zero-padded)
  const bf16_t* img = a.qkv + (long)b * a.n_src * C3 + head * D;
  const int* idx = a.idx ? a.idx + (long)b * np : nullptr;

  // a wave owns two blocks of 16 query rows: every K / V^T fragment read from LDS feeds two MFMAs
  int q[NB];
  bool valid[NB];
  s16x4_t qf[NB][8];                              // B operand of S^T: Q[q][16 kd + 4g + j]
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    q[u] = blockIdx.x * QROWS + (wave * NB + u) * 16 + r;
    valid[u] = q[u] < np;
    if (!valid[u]) q[u] = np - 1;
    const int srow = idx ? idx[q[u]] : q[u];
    const bf16_t* qp = img + (long)srow * C3;
#pragma unroll
    for (int kd = 0; kd < 8; ++kd) {
      qf[u][kd] = s16x4_t{0, 0, 0, 0};
      const int d0 = 16 * kd + 4 * g;
      if (kd < ND && d0 < D) qf[u][kd] = *reinterpret_cast<const s16x4_t*>(qp + d0);   // D % 4 == 0: whole or nothing
    }
  }
  f32x4 o[NB][8];
#pragma unroll
  for (int u = 0; u < NB; ++u)
#pragma unroll
    for (int db = 0; db < 8; ++db) o[u][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[NB], l_part[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) { m_run[u] = -INFINITY; l_part[u] = 0.f; }
  const int chunks8 = 2 * ND;                   // 8-element pieces per staged row (covering DP)

  // Staging is software-pipelined through registers: the global loads of chunk c+1 are issued before chunk c is
  // computed (up to 4 K pieces + 4 V pieces of 16 bytes per thread), so their latency hides under the MFMAs.
  uint4 kreg[KI], vreg[4];
  const int n_kitems = 64 * chunks8, n_vitems = 16 * chunks8;   // <= 1024 / <= 256
  auto fetch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int item = tid + NT * i;
      kreg[i] = make_uint4(0, 0, 0, 0);
      if (item < n_kitems) {
        const int row = item / chunks8, d0 = (item - row * chunks8) * 8;
        const int t = c0 + row;
        if (t < np && d0 < D) {                 // D % 8 == 0: a piece is whole or padding
          const int srow = idx ? idx[t] : t;
          kreg[i] = *reinterpret_cast<const uint4*>(img + (long)srow * C3 + C + d0);
        }
      }
    }
    if (tid < n_vitems) {                       // 8 dims of FOUR consecutive keys
      const int r4 = tid / chunks8, d0 = (tid - r4 * chunks8) * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = c0 + 4 * r4 + i;
        vreg[i] = make_uint4(0, 0, 0, 0);
        if (t < np && d0 < D) {
          const int srow = idx ? idx[t] : t;
          vreg[i] = *reinterpret_cast<const uint4*>(img + (long)srow * C3 + 2 * C + d0);
        }
      }
    }
  };
  fetch(0);

  for (int c0 = 0; c0 < np; c0 += 64) {
    __syncthreads();                            // everyone is done reading the previous chunk
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int item = tid + NT * i;
      if (item < n_kitems) {
        const int row = item / chunks8, d0 = (item - row * chunks8) * 8;
        *reinterpret_cast<uint4*>(sk + row * AG_KSTRIDE + d0) = kreg[i];
      }
    }
    if (tid < n_vitems) {                       // V transposed: eight 8-byte stores of 4 keys each
      const int r4 = tid / chunks8, d0 = (tid - r4 * chunks8) * 8;
      const unsigned w[4][4] = {{vreg[0].x, vreg[0].y, vreg[0].z, vreg[0].w}, {vreg[1].x, vreg[1].y, vreg[1].z, vreg[1].w},
                                {vreg[2].x, vreg[2].y, vreg[2].z, vreg[2].w}, {vreg[3].x, vreg[3].y, vreg[3].z, vreg[3].w}};
#pragma unroll
      for (int j = 0; j < 8; ++j) {             // element j of key i = half (j & 1) of dword j >> 1
        unsigned e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = (j & 1) ? (w[i][j >> 1] >> 16) : (w[i][j >> 1] & 0xFFFFu);
        *reinterpret_cast<uint2*>(svt + (d0 + j) * AG_VSTRIDE + 4 * r4) = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
      }
    }
    __syncthreads();
    if (c0 + 64 < np) fetch(c0 + 64);

    f32x4 st[NB][4];
    float mloc[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) mloc[u] = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int u = 0; u < NB; ++u) st[u][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kd = 0; kd < 8; ++kd)
        if (kd < ND) {
          const s16x4_t ka = *reinterpret_cast<const s16x4_t*>(sk + (16 * kb + r) * AG_KSTRIDE + 16 * kd + 4 * g);
#pragma unroll
          for (int u = 0; u < NB; ++u) st[u][kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ka, qf[u][kd], st[u][kb], 0, 0, 0);
        }
#pragma unroll
      for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = c0 + 16 * kb + 4 * g + j;
          const float v = key < np ? st[u][kb][j] * a.c : -INFINITY;   // log2 domain
          st[u][kb][j] = v;
          mloc[u] = fmaxf(mloc[u], v);
        }
    }
    float alpha[NB];
    s16x4_t pf[NB][4];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      float ml = fmaxf(mloc[u], __shfl_xor(mloc[u], 16, 64));
      ml = fmaxf(ml, __shfl_xor(ml, 32, 64));
      const float m_new = fmaxf(m_run[u], ml);      // finite: every chunk holds at least one real key
      alpha[u] = __builtin_amdgcn_exp2f(m_run[u] - m_new);
      m_run[u] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        float pv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pv[j] = __builtin_amdgcn_exp2f(st[u][kb][j] - m_new);
          psum += pv[j];
        }
        pf[u][kb] = __builtin_bit_cast(s16x4_t, make_uint2(pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])));
      }
      l_part[u] = fmaf(l_part[u], alpha[u], psum);
    }
#pragma unroll
    for (int db = 0; db < 8; ++db)
      if (db < ND) {
#pragma unroll
        for (int u = 0; u < NB; ++u) o[u][db] *= alpha[u];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const s16x4_t va = *reinterpret_cast<const s16x4_t*>(svt + (16 * db + r) * AG_VSTRIDE + 16 * kb + 4 * g);
#pragma unroll
          for (int u = 0; u < NB; ++u) o[u][db] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va, pf[u][kb], o[u][db], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    float l = l_part[u] + __shfl_xor(l_part[u], 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (valid[u]) {
      const float inv = 1.0f / l;
      bf16_t* op = a.out + ((long)b * np + q[u]) * C + head * D;
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        const int d0 = 16 * db + 4 * g;
        if (db < ND && d0 < D)
          *reinterpret_cast<uint2*>(op + d0) =
              make_uint2(pack2bf(o[u][db][0] * inv, o[u][db][1] * inv), pack2bf(o[u][db][2] * inv, o[u][db][3] * inv));
      }
    }
  }
}

// fp32 models, any head dim D % 4 == 0, D <= 128: the VALU kernel of attn_f32_d64 with DQ = D / 4 dims per lane
// (DQM = compile-time bound of DQ) and 32-key chunks.
template <int DQM>
__global__ void __launch_bounds__(256) attn_f32_dgen(const AttnArgsF32 a, int D) {
  __shared__ __attribute__((aligned(16))) float sk[32 * 4 * DQM];
  __shared__ __attribute__((aligned(16))) float sv[32 * 4 * DQM];
  const int tid = threadIdx.x;
  const int qrow = tid >> 2, pt = tid & 3;           // 64 query rows x 4 lanes
  const int head = blockIdx.y, b = blockIdx.z;
  const int np = a.np, C = a.H * D, C3 = 3 * C, DQ = D >> 2, D4 = D >> 2;
  const float* img = a.qkv + (long)b * a.n_src * C3 + head * D;
  const int* idx = a.idx ? a.idx + (long)b * np : nullptr;
  int q = blockIdx.x * 64 + qrow;
  const bool valid = q < np;
  if (!valid) q = np - 1;
  float qv[DQM], o[DQM];
  {
    const int srow = idx ? idx[q] : q;
    const float* qp = img + (long)srow * C3 + pt * DQ;
#pragma unroll
    for (int e = 0; e < DQM; ++e) {
      qv[e] = e < DQ ? qp[e] : 0.f;
      o[e] = 0.f;
    }
  }
  float m_run = -INFINITY, l_run = 0.f;
  for (int c0 = 0; c0 < np; c0 += 32) {
    __syncthreads();
    for (int e = tid; e < 32 * D4; e += 256) {       // float4 pieces of 32 K rows and 32 V rows
      const int row = e / D4, c4 = e - row * D4;
      const int t = c0 + row;
      float4 kq = make_float4(0, 0, 0, 0), vq = make_float4(0, 0, 0, 0);
      if (t < np) {
        const int srow = idx ? idx[t] : t;
        const float* rp = img + (long)srow * C3 + c4 * 4;
        kq = *reinterpret_cast<const float4*>(rp + C);
        vq = *reinterpret_cast<const float4*>(rp + 2 * C);
      }
      reinterpret_cast<float4*>(sk)[e] = kq;
      reinterpret_cast<float4*>(sv)[e] = vq;
    }
    __syncthreads();
    const int nk = np - c0 < 32 ? np - c0 : 32;
    for (int j = 0; j < nk; ++j) {
      const float* kr = sk + j * D + pt * DQ;
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < DQM; ++e)
        if (e < DQ) d = fmaf(qv[e], kr[e], d);
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      const float sc = d * a.c;
      const float m_new = fmaxf(m_run, sc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      const float pj = __builtin_amdgcn_exp2f(sc - m_new);
      l_run = fmaf(l_run, alpha, pj);
      m_run = m_new;
      const float* vr = sv + j * D + pt * DQ;
#pragma unroll
      for (int e = 0; e < DQM; ++e)
        if (e < DQ) o[e] = fmaf(o[e], alpha, pj * vr[e]);
    }
  }
  if (valid) {
    const float inv = 1.0f / l_run;
    float* op = a.out + ((long)b * np + q) * C + head * D + pt * DQ;
#pragma unroll
    for (int e = 0; e < DQM; ++e)
      if (e < DQ) op[e] = o[e] * inv;
  }
}

int g_force_attn = 0;  // 0 auto (persistent), 1 chunked online-softmax kernel, 2 one-shot full-row kernel (tests)

template <int NSUB>
int launch_full_o8(const AttnArgs& a, int B, hipStream_t s) {   // e4m3 output: persistent kernel only, 2 KiB staging per wave
  constexpr int lds = NSUB * 32 * 256 * 2 + 8 * 2048;
  static bool attr_by_device[RAJNI_MAX_DEVICES] = {};
  bool& attr = attr_by_device[rajni_current_device()];
  if (!attr && lds > 64 * 1024) {
    for (const void* fn : {reinterpret_cast<const void*>(&attn_bf16_d64_stream<NSUB, false, true>),
                           reinterpret_cast<const void*>(&attn_bf16_d64_stream<NSUB, true, true>)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) { rajni_set_error("hipFuncSetAttribute(attn): %s", hipGetErrorString(e)); return RAJNI_ERR_LAUNCH; }
    }
    attr = true;
  }
  const int per_cu = (160 * 1024) / lds >= 2 ? 2 : 1;
  const int items = a.H * B;
  const int cus = rajni_num_cus();
  const int grid = items < cus * per_cu ? items : cus * per_cu;
  if (a.idx != nullptr)
    hipLaunchKernelGGL((attn_bf16_d64_stream<NSUB, true, true>), dim3(grid), dim3(ATF_THREADS), lds, s, a, items);
  else
    hipLaunchKernelGGL((attn_bf16_d64_stream<NSUB, false, true>), dim3(grid), dim3(ATF_THREADS), lds, s, a, items);
  return RAJNI_OK;
}

template <int NSUB>
int launch_full(const AttnArgs& a, int B, hipStream_t s) {
  if (g_force_attn == 2) {
    constexpr int lds = NSUB * 32 * 128 * 2;
    hipLaunchKernelGGL(attn_bf16_d64_full<NSUB>, dim3(a.H, B), dim3(ATF_THREADS), lds, s, a);
    return RAJNI_OK;
  }
  constexpr int lds = NSUB * 32 * 256 * 2 + (stage_o(NSUB) ? 8 * 4096 : 0);   // two K+V buffers (+ output staging)
  static bool attr_by_device[RAJNI_MAX_DEVICES] = {};
  bool& attr = attr_by_device[rajni_current_device()];
  if (!attr && lds > 64 * 1024) {
    for (const void* fn : {reinterpret_cast<const void*>(&attn_bf16_d64_stream<NSUB, false>),
                           reinterpret_cast<const void*>(&attn_bf16_d64_stream<NSUB, true>)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) { rajni_set_error("hipFuncSetAttribute(attn): %s", hipGetErrorString(e)); return RAJNI_ERR_LAUNCH; }
    }
    attr = true;
  }
  const int per_cu = (160 * 1024) / lds >= 2 ? 2 : 1;   // 512-thread workgroups resident per CU
  const int items = a.H * B;
  const int cus = rajni_num_cus();
  const int grid = items < cus * per_cu ? items : cus * per_cu;
  if (a.idx != nullptr)
    hipLaunchKernelGGL((attn_bf16_d64_stream<NSUB, true>), dim3(grid), dim3(ATF_THREADS), lds, s, a, items);
  else
    hipLaunchKernelGGL((attn_bf16_d64_stream<NSUB, false>), dim3(grid), dim3(ATF_THREADS), lds, s, a, items);
  return RAJNI_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// CLS-query attention: out[b, h*D + d] = softmax_n(q[b,0,h] . k[b,n,h] * scale) . v[b,n,h,d] over all
// N tokens - the only attention row the classifier head can see in the LAST block (model.py:65-66 reads
// x[:, 0]).  One wave per (image, head): lanes over keys for the logits, lanes over d for the output.
// fp32 math on the stored activations (P is not rounded to bf16 here).  D % 8 == 0, D <= 128.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) attn_cls_kernel(const T* qkv, T* out, int N, int H, int D, float c, float q_scale) {
  extern __shared__ __attribute__((aligned(16))) float cls_sm[];
  float* qs = cls_sm;                // [D]
  float* prob = cls_sm + D;          // [N]
  const int lane = threadIdx.x, head = blockIdx.x, b = blockIdx.y;
  const int C = H * D, C3 = 3 * C;
  const T* img = qkv + (long)b * N * C3 + head * D;
  for (int d = lane; d < D; d += 64) qs[d] = ld1(img + d);
  __syncthreads();
  float mx = -INFINITY;
  for (int n = lane; n < N; n += 64) {
    const T* kr = img + (long)n * C3 + C;
    float dot = 0.f;
    for (int i = 0; i < (D >> 3); ++i) {
      float kf[8];
      load8<T>(kr + 8 * i, kf);
#pragma unroll
      for (int j = 0; j < 8; ++j) dot = fmaf(qs[8 * i + j], kf[j], dot);
    }
    dot *= c;                                                             // log2 domain
    prob[n] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int n = lane; n < N; n += 64) {
    const float pr = __builtin_amdgcn_exp2f(prob[n] - mx);
    prob[n] = pr;
    sum += pr;
  }
  sum = wave_sum(sum);
  __syncthreads();
  for (int d = lane; d < D; d += 64) {                                    // lane = d
    float acc = 0.f;
    const T* vc = img + 2 * C + d;
    for (int n = 0; n < N; ++n) acc = fmaf(prob[n], ld1(vc + (long)n * C3), acc);
    float o = acc / sum;
    if (q_scale > 0.f) {   // the rounding rajni_attention_fp8 applies to this row in the all-rows form of the block
      const float t = __builtin_amdgcn_fmed3f(o * (1.0f / q_scale), -448.f, 448.f);
      o = __builtin_amdgcn_cvt_f32_fp8(__builtin_amdgcn_cvt_pk_fp8_f32(t, 0.f, 0, false), 0) * q_scale;
    }
    st1(out + (long)b * C + head * D + d, o);
  }
}

int launch_attention_cls(const void* qkv, void* out, int B, int N, int H, int D, float scale, int dtype,
                         hipStream_t s, float q_scale) {
  RAJNI_REQUIRE(qkv && out && D >= 8 && D <= 128 && D % 8 == 0 && B > 0 && H > 0 && N > 0 && B <= 65535 &&
                (size_t)(N + D) * sizeof(float) <= 64 * 1024, RAJNI_ERR_INVALID, "attention_cls: bad arguments");
  const float c = scale * 1.4426950408889634f;
  const size_t lds = (size_t)(N + D) * sizeof(float);
  ProfScope prof(KC_ATTENTION, s, 4.0 * B * H * (double)N * D, 2.0 * B * (double)N * H * D * (dtype == RAJNI_F32 ? 4.0 : 2.0));
  if (dtype == RAJNI_F32)
    hipLaunchKernelGGL(attn_cls_kernel<float>, dim3(H, B), dim3(64), lds, s, (const float*)qkv, (float*)out, N, H, D, c, 0.f);
  else
    hipLaunchKernelGGL(attn_cls_kernel<bf16_t>, dim3(H, B), dim3(64), lds, s, (const bf16_t*)qkv, (bf16_t*)out, N, H, D, c, q_scale);
  RAJNI_CHECK_LAUNCH("attn_cls_kernel");
  return RAJNI_OK;
}

int launch_attention(const void* qkv, const int32_t* keep_idx, void* out, int B, int n_src, int np,
                     int H, int D, float scale, int dtype, hipStream_t s) {
  RAJNI_REQUIRE(qkv && out, RAJNI_ERR_INVALID, "rajni_attention: null pointer");
  RAJNI_REQUIRE(D >= 8 && D <= 128 && D % 8 == 0, RAJNI_ERR_UNSUPPORTED,
                "rajni_attention: head dim %d not supported (multiples of 8 up to 128)", D);
  RAJNI_REQUIRE(B > 0 && H > 0 && np > 0 && n_src >= np, RAJNI_ERR_INVALID,
                "rajni_attention: bad shape B=%d H=%d np=%d n_src=%d", B, H, np, n_src);
  RAJNI_REQUIRE(keep_idx != nullptr || np == n_src, RAJNI_ERR_INVALID,
                "rajni_attention: identity selection needs np == n_src");
  RAJNI_REQUIRE(H <= 65535 && B <= 65535, RAJNI_ERR_UNSUPPORTED, "rajni_attention: grid too large");
  if (dtype == RAJNI_F32) {
    AttnArgsF32 f{};
    f.qkv = (const float*)qkv; f.idx = keep_idx; f.out = (float*)out;
    f.n_src = n_src; f.np = np; f.H = H; f.c = scale * 1.4426950408889634f;
    ProfScope prof(KC_ATTENTION, s, 4.0 * B * H * (double)np * np * D, 4.0 * B * (double)np * H * D * 4.0);
    const dim3 grid((np + 63) / 64, H, B);
    if (D == 64) hipLaunchKernelGGL(attn_f32_d64, grid, dim3(256), 0, s, f);
    else if (D < 64) hipLaunchKernelGGL(attn_f32_dgen<16>, grid, dim3(256), 0, s, f, D);
    else hipLaunchKernelGGL(attn_f32_dgen<32>, grid, dim3(256), 0, s, f, D);
    RAJNI_CHECK_LAUNCH("attn_f32");
    return RAJNI_OK;
  }
  RAJNI_REQUIRE(dtype == RAJNI_BF16, RAJNI_ERR_INVALID, "rajni_attention: bad dtype %d", dtype);
  AttnArgs a{};
  a.qkv = (const bf16_t*)qkv; a.idx = keep_idx; a.out = (bf16_t*)out;
  a.n_src = n_src; a.np = np; a.H = H;
  a.c = scale * 1.4426950408889634f;
  a.stamps = rajni_g_stamps;
  const double flops = 4.0 * B * H * (double)np * np * D;
  const double bytes = 2.0 * B * (double)np * H * D * 4.0;
  ProfScope prof(KC_ATTENTION, s, flops, bytes);
  const int nsub = (np + 31) / 32;
  if (D != 64) {
    #ifndef RAJNI_ATTN_DGEN_NW
#define RAJNI_ATTN_DGEN_NW 4
#endif
#ifndef RAJNI_ATTN_DGEN_NB
#define RAJNI_ATTN_DGEN_NB 2
#endif
    constexpr int qrows = RAJNI_ATTN_DGEN_NW * RAJNI_ATTN_DGEN_NB * 16;
    hipLaunchKernelGGL((attn_bf16_dgen<RAJNI_ATTN_DGEN_NW, RAJNI_ATTN_DGEN_NB>), dim3((np + qrows - 1) / qrows, H, B),
                       dim3(RAJNI_ATTN_DGEN_NW * 64), 0, s, a, D);
  } else if (nsub <= 8 && g_force_attn != 1) {
    int rc = RAJNI_OK;
    switch (nsub) {
      case 1: rc = launch_full<1>(a, B, s); break;
      case 2: rc = launch_full<2>(a, B, s); break;
      case 3: rc = launch_full<3>(a, B, s); break;
      case 4: rc = launch_full<4>(a, B, s); break;
      case 5: rc = launch_full<5>(a, B, s); break;
      case 6: rc = launch_full<6>(a, B, s); break;
      case 7: rc = launch_full<7>(a, B, s); break;
      default: rc = launch_full<8>(a, B, s); break;
    }
    if (rc != RAJNI_OK) return rc;
  } else {
    hipLaunchKernelGGL(attn_bf16_d64, dim3((np + AT_QROWS - 1) / AT_QROWS, H, B), dim3(AT_THREADS), 0,
                       s, a);
  }
  RAJNI_CHECK_LAUNCH("attn_bf16_d64");
  return RAJNI_OK;
}

// attention with e4m3 output rows (rajni_attention_fp8): bf16 qkv, head dim 64, np <= 224 (the persistent kernel with
// LDS to spare for the output staging)
int launch_attention_fp8(const void* qkv, const int32_t* keep_idx, void* out_q, float out_scale, float* row_scale,
                         int B, int n_src, int np, int H, int D, float scale, hipStream_t s) {
  RAJNI_REQUIRE(qkv && out_q && row_scale, RAJNI_ERR_INVALID, "rajni_attention_fp8: null pointer");
  RAJNI_REQUIRE(D == 64 && np <= 224, RAJNI_ERR_UNSUPPORTED,
                "rajni_attention_fp8: head dim 64 and at most 224 tokens (D=%d np=%d)", D, np);
  RAJNI_REQUIRE(B > 0 && H > 0 && np > 0 && n_src >= np, RAJNI_ERR_INVALID,
                "rajni_attention_fp8: bad shape B=%d H=%d np=%d n_src=%d", B, H, np, n_src);
  RAJNI_REQUIRE(keep_idx != nullptr || np == n_src, RAJNI_ERR_INVALID,
                "rajni_attention_fp8: identity selection needs np == n_src");
  RAJNI_REQUIRE(out_scale > 0.f && out_scale < INFINITY, RAJNI_ERR_INVALID, "rajni_attention_fp8: out_scale must be positive and finite");
  AttnArgs a{};
  a.qkv = (const bf16_t*)qkv; a.idx = keep_idx; a.out = nullptr;
  a.n_src = n_src; a.np = np; a.H = H;
  a.c = scale * 1.4426950408889634f;
  a.stamps = nullptr;
  a.out8 = (unsigned char*)out_q; a.oscale = out_scale; a.oinv = 1.0f / out_scale; a.row_scale = row_scale;
  ProfScope prof(KC_ATTENTION, s, 4.0 * B * H * (double)np * np * D, 2.0 * B * (double)np * H * D * 3.5);
  int rc = RAJNI_OK;
  switch ((np + 31) / 32) {
    case 1: rc = launch_full_o8<1>(a, B, s); break;
    case 2: rc = launch_full_o8<2>(a, B, s); break;
    case 3: rc = launch_full_o8<3>(a, B, s); break;
    case 4: rc = launch_full_o8<4>(a, B, s); break;
    case 5: rc = launch_full_o8<5>(a, B, s); break;
    case 6: rc = launch_full_o8<6>(a, B, s); break;
    default: rc = launch_full_o8<7>(a, B, s); break;
  }
  if (rc != RAJNI_OK) return rc;
  RAJNI_CHECK_LAUNCH("attn_bf16_d64_stream<e4m3 out>");
  return RAJNI_OK;
}

extern "C" void rajni_debug_force_attention(int mode) { g_force_attn = mode; }
