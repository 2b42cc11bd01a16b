// RAJNI importance score + per-image top-k selection + order-preserving compaction
// (SURVEY k3-k9,k14; reference importance.py:4-34, attention.py:31-39,58) in ONE launch per stage.
//
// One 1024-thread workgroup per image (16 waves: the two read passes are bound by loads in flight per CU).  HBM traffic = the K and V thirds of qkv, read once with
// 16-byte loads; logits, head-mean V, norms and scores live in LDS; selection is a rank count
// (N <= 577 scores, broadcast LDS reads) followed by wave ballot + popcount prefix compaction, which
// yields the ascending index list directly (the reference's topk -> sort).
//
// Numerics: fp32 throughout; the final score is rounded to the I/O dtype (what the reference
// returns, importance.py:34) and ranking is done on the rounded values - like the reference ranks
// its own dtype-rounded scores - with the DEFINED tie rule: larger first, then lower index; NaN = +inf.
// All reductions use fixed trees, so the same input always selects the same tokens.
#include "common.h"

namespace {

#ifndef RAJNI_SS_THREADS
#define RAJNI_SS_THREADS 1024
#endif
constexpr int SS_THREADS = RAJNI_SS_THREADS;
// threads (and LDS floats) of the token-mean partial sums: 512 like the 8-wave version of the kernel, so that the
// 16-wave one needs no more LDS (N = 589 with 16 heads still fits 160 KiB) and sums in the same fixed order
constexpr int SS_PART = 512;
#ifndef RAJNI_SS_KU
#define RAJNI_SS_KU 8   // K-pass chunks in flight per lane
#endif
#ifndef RAJNI_SS_KUM
#define RAJNI_SS_KUM 4  // ... in the one-pass form, next to RAJNI_SS_VUM V head rows of the lane's V item
#endif
#ifndef RAJNI_SS_VUM
#define RAJNI_SS_VUM 6
#endif

struct ScoreArgs {          // T = activation dtype (bf16_t or float)
  const void* qkv;          // T [B,N,3C] or null (select-only)
  const void* scores_in;    // T [B,N] (select-only)
  int N, H, D;
  float eps;
  int keep;                 // 0: scores only
  void* scores_out;         // T [B,N] or null
  int* keep_idx;            // [B,keep+1]
  void* next_scores;        // T [B,keep+1] or null
  unsigned long long* stamps;   // diagnostic builds (-DRAJNI_SS_STAMPS): 16 x u64 s_memtime per workgroup, or null
};
#ifdef RAJNI_SS_STAMPS
#define SS_STAMP(i) do { if (a.stamps != nullptr && threadIdx.x == 0) a.stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SS_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ float rank_key(float s) { return (s != s) ? INFINITY : s; }

// sum over an aligned group of `width` (4, 8 or 16) consecutive lanes, every lane gets the total; DPP adds in
// a fixed tree (bitwise reproducible): xor 1, xor 2, mirror within 8, mirror within 16
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group_sum(float v, int width) {
  v = dpp_add<0xB1>(v);                    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                    // quad_perm [2,3,0,1]
  if (width >= 8) v = dpp_add<0x141>(v);   // row_half_mirror
  if (width >= 16) v = dpp_add<0x140>(v);  // row_mirror
  return v;
}

// MERGED: logits and vbar have LDS regions of their own and K and V rows are read in ONE pass (a token's K and V
// thirds are 3072 contiguous bytes of its 4608-byte qkv row, and no V load has to wait for the softmax statistics);
// otherwise (N = 577 with 16 heads: 185 KiB would be needed) vbar reuses the logits' region after the statistics.
template <bool COMPUTE, typename T, bool MERGED>
__global__ void __launch_bounds__(SS_THREADS) score_select_kernel(const ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x;
  const int N = a.N, H = a.H, D = a.D, C = H * D;
  const int lg_sz = (H * N + 3) & ~3;
  const int region_sz = MERGED ? lg_sz + N * D : (((H * N > N * D) ? H * N : N * D) + 3) & ~3;
  float* qcls = sm;                    // [C]
  float* region = qcls + C;            // logits [H][N]  (then / followed by)  vbar [N][D]
  float* vbar = MERGED ? region + lg_sz : region;
  float* acls = region + region_sz;    // [N]
  float* sc = acls + N;                // vnorm [N] then scores [N]
  float* hstat = sc + N;               // [2H]
  float* part = hstat + 2 * H;         // [SS_PART]
  float* mean = part + SS_PART;        // [D]
  float* misc = mean + D;              // [16]
  int* wcount = reinterpret_cast<int*>(misc + 16);  // [SS_THREADS / 64]

  SS_STAMP(0);
  if (COMPUTE) {
    const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * N * 3 * C;
    const int LP = D >> 3;             // lanes per 2*D-byte head row
    // head dims 32 / 64 / 128: LP is 4 / 8 / 16 and the lanes of a head row are summed by DPP adds; any other
    // D % 8 == 0 (80, 96, 72 ...) takes the plain forms below (one lane per (token, head) dot product, serial norms)
    const bool pow2 = LP == 4 || LP == 8 || LP == 16;
    const int sub = tid % LP;
    const int grp = tid / LP, ngrp = SS_THREADS / LP;   // threads past ngrp * LP sit the head-row passes out

    // ---- CLS query row -> LDS (importance.py:18)
    for (int c = tid; c < (C >> 3); c += SS_THREADS) {
      float f[8];
      load8<T>(base + c * 8, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) qcls[c * 8 + j] = f[j];
    }
    __syncthreads();

    // ---- logits[h][n] = q_cls[h] . k[n,h] / sqrt(D)   (importance.py:19)
    // Work item = one 16-byte chunk c of K row n (C/8 chunks per row, lane-contiguous -> coalesced); the LP
    // chunks of a head sit in LP consecutive lanes and are summed with DPP adds (ds_bpermute shuffles and an
    // integer division per item made this loop 34 of the kernel's 60 us).  U loads in flight per thread.
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    const float inv_h = 1.0f / (float)H;
    constexpr int U = RAJNI_SS_KU;
    const int CP = C >> 3;
    const int dn = SS_THREADS / CP, dc = SS_THREADS - dn * CP;   // K item index += SS_THREADS
    // one K chunk: dot with the CLS query chunk, summed over the head's LP lanes, lane 0 of the group stores
    auto k_item = [&](const float (&kf)[8], int n, int c) {
      const float4 q0 = *reinterpret_cast<const float4*>(qcls + c * 8);
      const float4 q1 = *reinterpret_cast<const float4*>(qcls + c * 8 + 4);
      float dot = kf[0] * q0.x;
      dot = fmaf(kf[1], q0.y, dot); dot = fmaf(kf[2], q0.z, dot); dot = fmaf(kf[3], q0.w, dot);
      dot = fmaf(kf[4], q1.x, dot); dot = fmaf(kf[5], q1.y, dot); dot = fmaf(kf[6], q1.z, dot);
      dot = fmaf(kf[7], q1.w, dot);
      dot = group_sum(dot, LP);
      if ((c & (LP - 1)) == 0) region[(c / LP) * N + n] = dot * inv_sqrt_d;
    };
    auto k_pass = [&]() {
      if (!pow2) {
        for (int item = tid; item < N * H; item += SS_THREADS) {   // consecutive lanes = consecutive heads of a row
          const int n = item / H, h = item - n * H;
          const T* kp = base + (long)n * 3 * C + C + h * D;
          float dot = 0.f;
          for (int c = 0; c < LP; ++c) {
            float kf[8];
            load8<T>(kp + c * 8, kf);
#pragma unroll
            for (int j = 0; j < 8; ++j) dot = fmaf(kf[j], qcls[h * D + c * 8 + j], dot);
          }
          region[h * N + n] = dot * inv_sqrt_d;
        }
      } else {
        int n = tid / CP, c = tid - n * CP;
        while (n < N) {
          float kf[U][8];
          int nn[U], cc[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            nn[u] = n; cc[u] = c;
            if (n < N) load8<T>(base + (long)n * 3 * C + C + c * 8, kf[u]);
            n += dn; c += dc;
            if (c >= CP) { c -= CP; ++n; }
          }
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (nn[u] < N) k_item(kf[u], nn[u], cc[u]);
        }
      }
    };
    // ---- vbar[n][:] = mean_h v[n,h,:]   (importance.py:24): thread (token n, 16-byte slice `sub` of the head dim),
    //      12 head rows in flight, summed in head order (fixed tree)
    auto v_item = [&](int n) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const T* vp = base + (long)n * 3 * C + 2 * C + sub * 8;
      for (int h0 = 0; h0 < H; h0 += 12) {
        float vf[12][8];
#pragma unroll
        for (int u = 0; u < 12; ++u)
          if (h0 + u < H) load8<T>(vp + (h0 + u) * D, vf[u]);
#pragma unroll
        for (int u = 0; u < 12; ++u)
          if (h0 + u < H) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += vf[u][j];
          }
      }
      float* dst = vbar + n * D + sub * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = acc[j] * inv_h;
    };
    auto v_pass = [&]() {
      for (int n = grp < ngrp ? grp : N; n < N; n += ngrp) v_item(n);
    };
    // ---- per-head softmax statistics over ALL N tokens (importance.py:20), then A_cls[n] = mean_h softmax_h[n]
    //      (importance.py:21)
    auto softmax_stats = [&]() {
      for (int h = wave; h < H; h += SS_THREADS / 64) {
        float mx = -INFINITY;
        for (int n = lane; n < N; n += 64) mx = fmaxf(mx, region[h * N + n]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int n = lane; n < N; n += 64) se += __expf(region[h * N + n] - mx);
        se = wave_sum(se);
        if (lane == 0) { hstat[h] = mx; hstat[H + h] = se; }
      }
      __syncthreads();
      // A_cls[n] = (1/H) sum_h exp(l[h][n] - max_h) / sum_h: FOUR lanes per token, lane q sums heads q, q + 4, ... in head
      // order, the four partial sums join in a fixed quad tree (one lane per token left 80 % of the workgroup idle
      // through 12 dependent exp / divide steps: 2.5k of the kernel's 79k cycles at 197 tokens)
      for (int n0 = 0; n0 < N; n0 += SS_THREADS / 4) {
        const int n = n0 + (tid >> 2), q = tid & 3;
        float s = 0.f;
        if (n < N)
          for (int h = q; h < H; h += 4) s += __expf(region[h * N + n] - hstat[h]) / hstat[H + h];
        s = dpp_add<0xB1>(s);
        s = dpp_add<0x4E>(s);
        if (n < N && q == 0) acls[n] = s / (float)H;
      }
    };

    if (MERGED && pow2) {
      // ONE pass over the K and V thirds: per iteration a thread has U K chunks and the H head rows of one V
      // item in flight (every result is the same fixed-order sum as in the two-pass form: bit-identical scores)
      constexpr int UM = RAJNI_SS_KUM, VU = RAJNI_SS_VUM;
      int n = tid / CP, c = tid - n * CP;
      int nv = grp < ngrp ? grp : N;
      while (n < N || nv < N) {
        float kf[UM][8];
        int nn[UM], cc[UM];
#pragma unroll
        for (int u = 0; u < UM; ++u) {
          nn[u] = n; cc[u] = c;
          if (n < N) load8<T>(base + (long)n * 3 * C + C + c * 8, kf[u]);
          n += dn; c += dc;
          if (c >= CP) { c -= CP; ++n; }
        }
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const bool vdo = nv < N;
        const T* vp = base + (long)(vdo ? nv : 0) * 3 * C + 2 * C + sub * 8;
        float vf[VU][8];
#pragma unroll
        for (int u = 0; u < VU; ++u)
          if (vdo && u < H) load8<T>(vp + u * D, vf[u]);
#pragma unroll
        for (int u = 0; u < UM; ++u)
          if (nn[u] < N) k_item(kf[u], nn[u], cc[u]);
        if (vdo) {
#pragma unroll
          for (int u = 0; u < VU; ++u)
            if (u < H) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] += vf[u][j];
            }
          for (int h0 = VU; h0 < H; h0 += VU) {       // further rounds of VU head rows, summed in head order
#pragma unroll
            for (int u = 0; u < VU; ++u)
              if (h0 + u < H) load8<T>(vp + (h0 + u) * D, vf[u]);
#pragma unroll
            for (int u = 0; u < VU; ++u)
              if (h0 + u < H) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += vf[u][j];
              }
          }
          float* dst = vbar + nv * D + sub * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) dst[j] = acc[j] * inv_h;
          nv += ngrp;
        }
      }
      __syncthreads();
      SS_STAMP(1);
      softmax_stats();
      __syncthreads();
      SS_STAMP(2);
    } else {
      k_pass();
      __syncthreads();
      SS_STAMP(1);
      softmax_stats();
      __syncthreads();  // logits are dead: region becomes vbar (two-pass layout)
      SS_STAMP(2);
      v_pass();
      __syncthreads();
    }
    SS_STAMP(3);
    // ---- token mean of vbar, fixed-order two-level sum (importance.py:25)
    {
      const int d = tid % D, prt = tid / D, nparts = SS_PART / D;      // threads past nparts * D idle
      // four independent chains (tokens prt, prt + nparts, ...: chain = step mod 4), joined in a fixed tree: one
      // dependent LDS-read + add chain of N / nparts = 25 steps was 3k of the kernel's cycles
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int n = prt < nparts ? prt : N;
      for (; n + 3 * nparts < N; n += 4 * nparts) {
        s0 += vbar[n * D + d]; s1 += vbar[(n + nparts) * D + d];
        s2 += vbar[(n + 2 * nparts) * D + d]; s3 += vbar[(n + 3 * nparts) * D + d];
      }
      if (n < N) s0 += vbar[n * D + d];
      if (n + nparts < N) s1 += vbar[(n + nparts) * D + d];
      if (n + 2 * nparts < N) s2 += vbar[(n + 2 * nparts) * D + d];
      const float s = (s0 + s1) + (s2 + s3);
      if (prt < nparts) part[prt * D + d] = s;
      __syncthreads();
      if (tid < D) {
        float t = 0.f;
        for (int q = 0; q < nparts; ++q) t += part[q * D + tid];
        mean[tid] = t / (float)N;
      }
      __syncthreads();
    }
    // ---- ||vbar[n] - mean||_2   (importance.py:27)
    if (!pow2) {
      for (int n = tid; n < N; n += SS_THREADS) {
        float ss = 0.f;
        for (int d = 0; d < D; ++d) {
          const float dlt = vbar[n * D + d] - mean[d];
          ss = fmaf(dlt, dlt, ss);
        }
        sc[n] = sqrtf(ss);
      }
    } else
    for (int n = grp; n < N; n += ngrp) {
      const float* src = vbar + n * D + sub * 8;
      const float* mp = mean + sub * 8;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dlt = src[j] - mp[j];
        ss = fmaf(dlt, dlt, ss);
      }
      ss = group_sum(ss, LP);
      if (sub == 0) sc[n] = sqrtf(ss);
    }
    __syncthreads();
    SS_STAMP(4);
    // ---- mu, unbiased std + eps over tokens   (importance.py:28-29)
    //      (every wave reducing the norms itself - one barrier and the broadcast fewer - was measured: +400 cycles)
    if (wave == 0) {
      float s = 0.f;
      for (int n = lane; n < N; n += 64) s += sc[n];
      const float mu = wave_sum(s) / (float)N;
      float ss = 0.f;
      for (int n = lane; n < N; n += 64) {
        const float dlt = sc[n] - mu;
        ss = fmaf(dlt, dlt, ss);
      }
      ss = wave_sum(ss);
      if (lane == 0) {
        misc[0] = mu;
        misc[1] = sqrtf(ss / (float)(N - 1)) + a.eps;
      }
    }
    __syncthreads();
    // ---- score = A_cls * sigmoid(z), rounded to the I/O dtype   (importance.py:31-34)
    {
      const float mu = misc[0], sd = misc[1];
      for (int n = tid; n < N; n += SS_THREADS) {
        const float z = (sc[n] - mu) / sd;
        const float sig = 1.0f / (1.0f + __expf(-z));
        const float sv = round_to<T>(acls[n] * sig);   // what the reference returns: qkv's dtype
        if (a.scores_out != nullptr) st1(reinterpret_cast<T*>(a.scores_out) + (long)b * N + n, sv);
        sc[n] = sv;
      }
    }
    __syncthreads();
  } else {
    for (int n = tid; n < N; n += SS_THREADS) sc[n] = ld1(reinterpret_cast<const T*>(a.scores_in) + (long)b * N + n);
    __syncthreads();
  }

  SS_STAMP(5);
  if (a.keep <= 0) return;

  // ---- rank patch tokens 1..N-1, keep rank < keep, compact in ascending index order
  //      (attention.py:34-39: topk -> sort -> +1 -> prepend CLS; attention.py:58: carried scores)
  const int keep = a.keep;
  int* kout = a.keep_idx + (long)b * (keep + 1);
  T* nout = a.next_scores ? reinterpret_cast<T*>(a.next_scores) + (long)b * (keep + 1) : nullptr;
  int running = 0;
  // tpt = 1, 2, 4 or 8 lanes share a token's rank count (each a slice of the j range, summed by shuffles): with
  // 197 tokens one lane per token left 60 % of the workgroup idle through a 196-step loop
  const int tpt = (N - 1) * 8 <= SS_THREADS ? 8 : (N - 1) * 4 <= SS_THREADS ? 4 : (N - 1) * 2 <= SS_THREADS ? 2 : 1;
  const int per_iter = SS_THREADS / tpt, sl = tid & (tpt - 1);
  // The count loop is VALU bound (N^2 compares over 4 SIMDs: stamps put it at 14k of the kernel's 88k cycles at 197
  // tokens, 95k of 297k at 577, when a compare was ~8 instructions: two range tests, >, ==, index test, or / and / add).
  // 16-bit scores (the bf16 path): ONE unsigned compare per pair on a packed key
  //     key32[j] = sortable16(score_j) << 16 | (0xFFFF - j)        (NaN = +inf, -0 = +0; N <= 65535)
  // "j beats i" (larger score, or equal score and lower index - the defined tie rule) <=> key32[j] > key32[i], and the
  // CLS slot and the padding hold 0 (below every real key), so slices need no range tests: v_cmp + add-with-carry.
  // fp32 scores keep the float compare (the accuracy path).
  constexpr bool PACKED = sizeof(T) == 2;
  float* keys = acls;                                         // fp32 path: ranking keys (NaN = +inf), A_cls is dead
  unsigned* keys32 = reinterpret_cast<unsigned*>(region);     // packed path: the logits' region is dead (>= N + 4 * tpt + 4 words)
  const int chunks = (N + 3) >> 2, cps = (chunks + tpt - 1) / tpt;   // 16-byte chunks of keys; per slice
  const int cstride = cps | 1;     // slices an ODD number of chunks apart: the tpt broadcast reads of a step hit distinct banks
  if constexpr (PACKED) {
    for (int w = tid; w < cstride * tpt * 4; w += SS_THREADS) {
      const int slw = w / (cstride * 4), off = w - slw * cstride * 4;
      const int n = off < cps * 4 ? slw * cps * 4 + off : N;     // token of word w (padding words: none)
      unsigned k = 0;
      if (n >= 1 && n < N) {
        unsigned u = __float_as_uint(rank_key(sc[n]));
        u = (u == 0x80000000u) ? 0u : u;                            // -0 ranks as +0
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);             // monotone float -> unsigned
        k = (u & 0xFFFF0000u) | (unsigned)(0xFFFF - n);
      }
      keys32[w] = k;
    }
  } else {
    for (int n = tid; n < N; n += SS_THREADS) keys[n] = rank_key(sc[n]);
  }
  __syncthreads();
  SS_STAMP(7);
  const int slice = (N - 1 + tpt - 1) / tpt;
  for (int base_i = 1; base_i < N; base_i += per_iter) {
    const int i = base_i + tid / tpt;
    const bool valid = i < N;
    float si = 0.f;
    int rank = 0;
    if (valid) {
      si = sc[i];
      if constexpr (PACKED) {
        const int si_ = i / (cps * 4);                                  // slice and word that hold token i's key
        const unsigned ki = keys32[si_ * cstride * 4 + (i - si_ * cps * 4)];
        const uint4* kp = reinterpret_cast<const uint4*>(keys32) + sl * cstride;
#pragma unroll 4
        for (int c = 0; c < cps; ++c) {
          const uint4 k4 = kp[c];
          rank += (k4.x > ki) + (k4.y > ki) + (k4.z > ki) + (k4.w > ki);
        }
      } else {
        const float ki = keys[i];
        const int j0 = 1 + sl * slice, j1 = j0 + slice < N ? j0 + slice : N;
        // j beats i when its key is larger, or equal with a lower index (the defined tie rule)
        auto beats = [&](float kj, int j) { return (j >= j0 && j < j1 && (kj > ki || (kj == ki && j < i))) ? 1 : 0; };
#pragma unroll 4
        for (int j = j0 & ~3; j < j1; j += 4) {    // aligned 16-byte reads; entries outside [j0, j1) are masked
          const float4 k4 = *reinterpret_cast<const float4*>(keys + j);   // may run 3 floats into `sc`: masked
          rank += beats(k4.x, j) + beats(k4.y, j + 1) + beats(k4.z, j + 2) + beats(k4.w, j + 3);
        }
      }
    }
    SS_STAMP(8);
    if (tpt >= 2) rank += __shfl_xor(rank, 1, 64);
    if (tpt >= 4) rank += __shfl_xor(rank, 2, 64);
    if (tpt >= 8) rank += __shfl_xor(rank, 4, 64);
    const bool kept = valid && sl == 0 && rank < keep;
    const unsigned long long bal = __ballot(kept);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    SS_STAMP(9);
    int prefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SS_THREADS / 64; ++w) {
      const int cnt = wcount[w];
      prefix += (w < wave) ? cnt : 0;
      total += cnt;
    }
    if (kept) {
      const int pos = running + prefix + __popcll(bal & ((1ull << lane) - 1ull));
      kout[1 + pos] = i;
      if (nout) st1(nout + 1 + pos, si);
    }
    running += total;
    __syncthreads();
    SS_STAMP(10);
  }
  if (tid == 0) {
    kout[0] = 0;
    if (nout) st1(nout, sc[0]);
  }
  SS_STAMP(6);
}

size_t ss_lds_bytes(int N, int H, int D, bool merged) {
  const size_t C = (size_t)H * D;
  const size_t region = merged ? (size_t)((H * N + 3) & ~3) + (size_t)N * D
                               : (size_t)((((H * N > N * D) ? H * N : N * D) + 3) & ~3);
  return (C + region + 2 * (size_t)N + 2 * (size_t)H + SS_PART + D + 16 + SS_THREADS / 64) * sizeof(float);
}

int g_ss_force_two_pass = 0;   // test hook (rajni_hip_debug.h): 1 = the two-pass layout even when the merged one fits

}  // namespace

extern "C" void rajni_debug_force_score_two_pass(int on) { g_ss_force_two_pass = on; }

// qkv != null: compute scores (and select when keep > 0); qkv == null: select from scores_in.
int launch_score_select(const void* qkv, const void* scores_in, int B, int N, int H, int D,
                        float eps, int keep, void* scores_out, int32_t* keep_idx,
                        void* next_scores, int dtype, hipStream_t s) {
  RAJNI_REQUIRE(dtype == RAJNI_BF16 || dtype == RAJNI_F32, RAJNI_ERR_INVALID, "score/select: bad dtype %d", dtype);
  RAJNI_REQUIRE(B > 0 && N >= 2, RAJNI_ERR_INVALID, "score/select: need B > 0 and N >= 2 (B=%d N=%d)", B, N);
  RAJNI_REQUIRE(keep >= 0 && keep <= N - 1, RAJNI_ERR_INVALID,
                "score/select: keep=%d out of range for N=%d (keep_ratio must be <= 1)", keep, N);
  RAJNI_REQUIRE(keep == 0 || keep_idx != nullptr, RAJNI_ERR_INVALID, "score/select: keep_idx is null");
  ScoreArgs a{};
  a.N = N; a.keep = keep; a.eps = eps;
  a.scores_out = scores_out; a.keep_idx = keep_idx; a.next_scores = next_scores;
  a.stamps = rajni_g_stamps;
  size_t lds;
  bool merged = false;
  if (qkv != nullptr) {
    RAJNI_REQUIRE(D >= 8 && D <= 128 && D % 8 == 0, RAJNI_ERR_UNSUPPORTED,
                  "importance: head dim %d not supported (multiples of 8 up to 128)", D);
    RAJNI_REQUIRE(H > 0, RAJNI_ERR_INVALID, "importance: H must be positive");
    a.qkv = qkv; a.H = H; a.D = D;
    merged = ss_lds_bytes(N, H, D, true) <= 160 * 1024 && g_ss_force_two_pass == 0;   // else vbar reuses the logits' region
    lds = ss_lds_bytes(N, H, D, merged);
  } else {
    RAJNI_REQUIRE(scores_in != nullptr, RAJNI_ERR_INVALID, "select: scores is null");
    a.scores_in = scores_in; a.H = 1; a.D = 32;
    lds = ss_lds_bytes(N, 1, 32, false);
  }
  RAJNI_REQUIRE(lds <= 160 * 1024, RAJNI_ERR_UNSUPPORTED,
                "score/select: N=%d H=%d D=%d needs %zu B of LDS (> 160 KiB)", N, H, D, lds);
  const bool f32 = dtype == RAJNI_F32;
  typedef void (*kern_t)(const ScoreArgs);
  const kern_t kern = qkv ? (merged ? (f32 ? &score_select_kernel<true, float, true> : &score_select_kernel<true, bf16_t, true>)
                                    : (f32 ? &score_select_kernel<true, float, false> : &score_select_kernel<true, bf16_t, false>))
                          : (f32 ? &score_select_kernel<false, float, false> : &score_select_kernel<false, bf16_t, false>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      rajni_set_error("hipFuncSetAttribute(score_select, %zu): %s", lds, hipGetErrorString(e));
      return RAJNI_ERR_LAUNCH;
    }
  }
  const double es = f32 ? 4.0 : 2.0;
  const double bytes = qkv ? (2.0 * N * H * D + H * D) * es * B + 4.0 * N * B : 6.0 * N * B;
  ProfScope prof(qkv ? (keep > 0 ? KC_SCORE_SELECT : KC_IMPORTANCE) : KC_SELECT, s, 0.0, bytes);
  hipLaunchKernelGGL(kern, dim3(B), dim3(SS_THREADS), lds, s, a);
  RAJNI_CHECK_LAUNCH("score_select_kernel");
  return RAJNI_OK;
}
