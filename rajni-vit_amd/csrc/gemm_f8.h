// fp8 x fp8 "TN" GEMM on the CDNA4 fp8 matrix pipe (BASELINE.json configs[4], "CDNA4 fp8 MFMA").
// Included by gemm.hip inside its anonymous namespace (same GemmParams, column maps, epilogue helpers).
//
//   Y[m][n] = epi( xscale[m] * wscale[n] * sum_k Xq[m][k] * Wq[n][k] )     Xq, Wq: OCP e4m3 bytes, K-contiguous
//
// v_mfma_f32_16x16x128_f8f6f4 (unscaled form: both scale arguments 0; tools/f8_mfma_probe.hip checks the semantics
// with exact integer data and measures 4.7 PFLOP/s for the bare instruction stream) does 4x the K of the bf16
// instruction in 2x its cycles.  A K step is 128 fp8 = 128 BYTES per row, i.e. byte for byte the LDS tile geometry
// of the bf16 tilings (128-byte rows, 1 KiB LDS-DMA pieces of 8 rows, XOR swizzle on 16-byte chunks), so staging,
// swizzles and the tile order are the bf16 kernel's; only the fragments differ: a lane's operand is 32 bytes =
// chunks 2g and 2g+1 of its row (lane group g = lane >> 4) - for BOTH operands, which is all the dot product needs.
// Per-row activation scales and per-row weight scales multiply the fp32 accumulator in the epilogue (block scales of
// the instruction stay 1).
//
// Tiling: 256(M) x 128(N) x 128(K bytes), 8 waves of 64 x 64, 3 LDS stages of 48 KiB, ONE persistent workgroup per
// CU.  Both fragment sets of a K step are live (MFMAs of step j run while step j+1's 16 ds_read_b128 land in the
// other set): 2 x 64 fragment VGPRs + 64 accumulators; the 256 x 256 tile would need 320.  K step j:
//   s_waitcnt lgkmcnt(0) vmcnt(PIECES) ; s_barrier      step j's fragments are in registers, tile j+1 has landed,
//                                                        everyone is done reading stage j
//   16 MFMAs(j)  ||  16 LDS reads of tile j+1  ||  6 DMA pieces of tile j+3 into stage j
// X row indices are clamped to M-1 (rows past M are computed and never stored), so any M >= 1 works.
// Host contract: K % 256 == 0 (even number of K steps: the two fragment sets alternate by unrolling), K >= 512.
namespace f8 {
using wide::BM;
using wide::X_BYTES;
using wide::key_x;
using wide::wait_step;
constexpr int BN = 128, NS = 3, MI = 4, NI = 4, WM = 4, WN = 2;
constexpr int KB = 128;                           // bytes (= fp8 elements) of K per step
constexpr int W_BYTES = BN * KB;                  // 16 KiB
constexpr int STAGE_BYTES = X_BYTES + W_BYTES;    // 48 KiB
constexpr int LDS_BYTES = NS * STAGE_BYTES;       // 144 KiB
constexpr int XP = 4, PW = 2, PIECES = XP + PW;   // 1 KiB pieces per wave per K step
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

struct Frag { bf16x8 lo, hi; };                   // read through __bf16 vectors: see the TBAA note at fp8x8_raw
__device__ __forceinline__ v8i frag_bits(const Frag& f) {
  return __builtin_shufflevector(__builtin_bit_cast(v4i, f.lo), __builtin_bit_cast(v4i, f.hi), 0, 1, 2, 3, 4, 5, 6, 7);
}
// issue order of a step: MI groups of {NI MFMAs, 4 fragment reads, 1-2 DMA pieces}
template <int G, bool READ>
__device__ __forceinline__ void sched_step() {
  if constexpr (G < MI) {
    __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
    if constexpr (READ) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI) / MI, 0);
    constexpr int nd = (G + 1) * PIECES / MI - G * PIECES / MI;
    if constexpr (nd > 0) __builtin_amdgcn_sched_group_barrier(0x020, nd, 0);
    sched_step<G + 1, READ>();
  }
}
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
  // saturating RNE conversion (the hardware convert alone would produce NaN past 448)
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (unsigned)r;
}

template <int EPI, bool SF32, int TAG = 0>
__global__ void __launch_bounds__(512, 2) gemm_f8_tn_stream(const GemmParams p) {
  constexpr int MAP = col_map(EPI, SF32);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = p.total_tiles / p.tiles_n;
  auto tile_mn = [&](int t, int& tm_, int& tn_) {   // (N block, row tile, column in block): see gemm_bf16_tn_stream
    if (p.nblk >= p.tiles_n) { tm_ = t / p.tiles_n; tn_ = t - tm_ * p.tiles_n; return; }
    const int per = p.nblk * tiles_m, blk = t / per, r = t - blk * per;
    const int left = p.tiles_n - blk * p.nblk, nb = left < p.nblk ? left : p.nblk;
    const int rr = r / nb;
    tm_ = rr; tn_ = blk * p.nblk + (r - rr * nb);
  };
  // ---- staging: a piece = 8 rows x 128 B; wave w stages X pieces 4w..4w+3 and W pieces 2w, 2w+1
  const char* xp[XP];
  const char* wp[PW];
  const int r_in = lane >> 3, pch = lane & 7;
  auto point_at = [&](int tile) {
    int tm, tn;
    tile_mn(tile, tm, tn);
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int row = (wave * XP + i) * 8 + r_in;
      int m = tm * BM + row;
      if (m > p.M - 1) m = p.M - 1;   // clamp: duplicates are computed but never stored
      xp[i] = reinterpret_cast<const char*>(p.X) + (long)m * p.lda + ((pch ^ key_x(row)) << 4);
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int row = (wave * PW + i) * 8 + r_in;
      wp[i] = reinterpret_cast<const char*>(p.W) + (long)(tn * BN + row) * p.ldw + ((pch ^ w_key<MAP>(row)) << 4);
    }
  };
  auto dma_piece = [&](int q, int k0, char* dx) {
    if (q < XP)
      __builtin_amdgcn_global_load_lds(GLB_PTR(xp[q] + k0), LDS_PTR(dx + (wave * XP + q) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(GLB_PTR(wp[q - XP] + k0), LDS_PTR(dx + X_BYTES + (wave * PW + q - XP) * 1024), 16, 0, 0);
  };
  // ---- fragment addresses (chunks 2g, 2g+1 of the lane's row; one swizzle key per operand, see gemm_bf16_tn_stream)
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;
  const int xr0 = wm * (MI * 16) + l15;
  const int wr0 = wn * 64 + w_frag_row<MAP>(l15, 0);
  auto w_ni_off = [](int ni) { return (w_frag_row<MAP>(0, ni) - w_frag_row<MAP>(0, 0)) * KB; };
  int xo[2], wo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    xo[h] = xr0 * KB + (((2 * g + h) ^ key_x(xr0)) << 4);
    wo[h] = X_BYTES + wr0 * KB + (((2 * g + h) ^ w_key<MAP>(wr0)) << 4);
  }
  auto read_frags = [&](Frag (&xf)[MI], Frag (&wf)[NI], int st) {
    const char* sb = smem + st * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      xf[i].lo = *reinterpret_cast<const bf16x8*>(sb + xo[0] + i * 16 * KB);
      xf[i].hi = *reinterpret_cast<const bf16x8*>(sb + xo[1] + i * 16 * KB);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      wf[i].lo = *reinterpret_cast<const bf16x8*>(sb + wo[0] + w_ni_off(i));
      wf[i].hi = *reinterpret_cast<const bf16x8*>(sb + wo[1] + w_ni_off(i));
    }
  };

  f32x4 acc[4][MI];   // [ni][mi]
  // one K step: MFMAs on (xc, wc)  ||  fragments of the next K tile (stage rst) -> (xn, wn_)  ||  DMA of K-tile dkt
  // of the pointed-at tile into stage dst
  auto step = [&](auto read_c, Frag (&xc)[MI], Frag (&wc)[NI], Frag (&xn)[MI], Frag (&wn_)[NI], int rst, int dkt, int dst) {
    constexpr bool READ = decltype(read_c)::value;
    const char* sb = smem + rst * STAGE_BYTES;
    char* dx = smem + dst * STAGE_BYTES;
    const int k0 = dkt * KB;
    v8i wv[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wv[ni] = frag_bits(wc[ni]);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const v8i xv = frag_bits(xc[mi]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv[ni], xv, acc[ni][mi], 0, 0, 0, 0, 0, 0);
      if constexpr (READ) {
        xn[mi].lo = *reinterpret_cast<const bf16x8*>(sb + xo[0] + mi * 16 * KB);
        xn[mi].hi = *reinterpret_cast<const bf16x8*>(sb + xo[1] + mi * 16 * KB);
        wn_[mi].lo = *reinterpret_cast<const bf16x8*>(sb + wo[0] + w_ni_off(mi));     // MI == NI: one W fragment per group
        wn_[mi].hi = *reinterpret_cast<const bf16x8*>(sb + wo[1] + w_ni_off(mi));
      }
#pragma unroll
      for (int q = mi * PIECES / MI; q < (mi + 1) * PIECES / MI; ++q) dma_piece(q, k0, dx);
    }
    sched_step<0, READ>();
  };
  static_assert(MI == NI, "step() pairs X fragment mi with W fragment mi");

  const int nk = p.K / KB;        // even, >= NS + 1 (host checked)
  int v = blockIdx.x;
  int tile = xcd_tile_of(v, p.total_tiles);
  Frag xa[MI], wa[NI], xb[MI], wb[NI];
  constexpr int WBASE = (NS - 2) * PIECES;
  // stores of the previous tile's epilogue may still be in flight at a tile's first counted wait (gfx9 counts stores
  // in vmcnt and retires vector-memory ops in order): a lower bound on their number keeps that wait from draining them
  constexpr int NSTORE = EPI == EPI_GELU8 ? MI : (nat_order(EPI, SF32) ? 4 * MI : 2 * MI);
  bool prev_full = false;
  point_at(tile);
#pragma unroll
  for (int j = 0; j < NS; ++j) {
#pragma unroll
    for (int q = 0; q < PIECES; ++q) dma_piece(q, j * KB, smem + j * STAGE_BYTES);
  }
  wait_step<(NS - 1) * PIECES>();   // K tile 0 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(xa, wa, 0);
  int st = 0;   // LDS stage of the current K step

  while (true) {
#ifdef RAJNI_GEMM_STAMPS
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    int tm, tn;
    tile_mn(tile, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const bool inter = m0 + BM <= p.M && n0 + BN <= p.N;
    const int vn = v + gridDim.x;
    const bool more = vn < p.total_tiles;
    const int m_base = m0 + wm * (MI * 16), n0w = n0 + wn * 64;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float xsr[MI];   // dequantisation scale of each of the lane's rows
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xsr[mi] = 0.f;

    // one K step of the stream; the LAST step of a tile reads no fragments (the next tile's first fragments are read
    // after the epilogue instead: one exposed LDS round trip per tile) - with both fragment sets live through the
    // epilogue its operands do not fit 256 VGPRs, and the registers freed in the last step hold the early
    // residual-row loads of the fp32-stream epilogue
    auto kstep = [&](auto last_c, auto odd_c, int k) {
      constexpr bool LAST = decltype(last_c)::value, ODD = decltype(odd_c)::value;
      // the DMA of step k loads K-tile k+NS; from k = nk-NS on that is the NEXT tile's K-tile 0..
      if (k == nk - NS && more) point_at(xcd_tile_of(vn, p.total_tiles));
      const int dkt = k + NS < nk ? k + NS : k + NS - nk;
      const int st1 = st + 1 == NS ? 0 : st + 1;
      if constexpr (LAST) {   // the epilogue's per-row data, loaded under the MFMAs
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int m = m_base + mi * 16 + l15;
          xsr[mi] = p.xscale[m < p.M ? m : p.M - 1];
        }
      }
      if (k == 0 && prev_full) wait_step<WBASE + NSTORE>();
      else wait_step<WBASE>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      using RD = std::integral_constant<bool, !LAST>;
      if constexpr (!ODD) step(RD{}, xa, wa, xb, wb, st1, dkt, st);
      else step(RD{}, xb, wb, xa, wa, st1, dkt, st);
      st = st1;
    };
    using T = std::true_type; using F = std::false_type;
    for (int kt = 0; kt < nk - 2; kt += 2) {
      kstep(F{}, F{}, kt);
      kstep(F{}, T{}, kt + 1);
    }
    kstep(F{}, F{}, nk - 2);
    kstep(T{}, T{}, nk - 1);
#ifdef RAJNI_GEMM_STAMPS
    asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[3][MI - 1][3]));
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue (the next tile's first loads are in flight): dequantise rows, then the shared tile epilogues
    // (the fence keeps hipcc from hoisting the epilogue's loads among the last step's MFMAs, where their targets
    // overlap the fragments still being read: 84 bytes of spills, each a serialised load-wait-store)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni][mi] *= xsr[mi];
    if constexpr (EPI == EPI_GELU8) {
      // lane = one row per mi, 16 consecutive columns c0..c0+15 (MAP_F8); y = e4m3(gelu(acc * ws + bias) / yscale[m])
      const int c0 = n0w + 16 * g;
      float bs[16], ws[16];
      const bool full = c0 + 16 <= p.N;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int n = c0 + j;
        bs[j] = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
        ws[j] = n < p.N ? p.wscale[n] : 0.f;
      }
      float inv[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = m_base + mi * 16 + l15;
        inv[mi] = 1.0f / p.yscale[m < p.M ? m : p.M - 1];
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) on every path: see epilogue_tile
      unsigned char* Y = reinterpret_cast<unsigned char*>(p.Y);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = m_base + mi * 16 + l15;
        if (m >= p.M) continue;
        float y[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) y[j] = fmaf(acc[j >> 2][mi][j & 3], ws[j], bs[j]);
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
          f32x2 ta = f32x2{y[j], y[j + 1]}, tb = f32x2{y[j + 2], y[j + 3]};
          gelu_pk4(ta, tb);
          y[j] = ta[0] * inv[mi]; y[j + 1] = ta[1] * inv[mi]; y[j + 2] = tb[0] * inv[mi]; y[j + 3] = tb[1] * inv[mi];
        }
        uint4 q;
        q.x = pack4_e4m3(y[0], y[1], y[2], y[3]);   q.y = pack4_e4m3(y[4], y[5], y[6], y[7]);
        q.z = pack4_e4m3(y[8], y[9], y[10], y[11]); q.w = pack4_e4m3(y[12], y[13], y[14], y[15]);
        unsigned char* row = Y + (long)m * p.ldc + c0;
        if (full) {
          *reinterpret_cast<uint4*>(row) = q;
        } else {
          const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (c0 + j < p.N) row[j] = (unsigned char)(w[j >> 2] >> (8 * (j & 3)));
        }
      }
    } else {
      // fp32-stream RESID: the residual rows of an interior tile are loaded first thing in the epilogue - not in
      // the last K step like the bf16 tiling does: two fragment sets + accumulators + 16 float4 do not fit 256
      // VGPRs, and every variant that held part of them across the K loop spilled (100-270 bytes, each spill a
      // serialised load-wait-store)
      ResidPrefetch<MI> pre;
      pre.valid = false;
      char* scratch = nullptr;
      if constexpr (EPI == EPI_BIAS) scratch = smem + LDS_BYTES + wave * 2048;   // bf16 output (QKV) as whole lines: epilogue_tile
      if constexpr (nat_order(EPI, SF32) && EPI == EPI_RESID) {
        // loaded on EVERY path (addresses clamped into the tensor; edge tiles ignore the values): a conditional load
        // leaves `pre` half-defined and hipcc then keeps its registers reserved around the whole tile loop.
        // Round 3: in FULL 128-byte LINES - row group mi, instruction j: row 16 mi + 8 j + (lane >> 3), columns 4 (lane & 7)..
        // of each 32-column half - the layout epilogue_tile's LDS transpose brings the accumulators to (one CU pulls 64-byte
        // pieces at a third of the rate of whole lines: prefetch_resid_rowmajor in gemm.hip)
        const float* R = reinterpret_cast<const float*>(p.R);
        const int cbase = n0w + 64 <= p.N ? n0w + 4 * (lane & 7) : 0;
        if (p.ridx != nullptr) {
          int gi[MI][2], mm[MI][2];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              int m = m_base + mi * 16 + j * 8 + (lane >> 3);
              mm[mi][j] = m > p.M - 1 ? p.M - 1 : m;
              gi[mi][j] = p.ridx[mm[mi][j]];
            }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float* rp = R + ((long)(mm[mi][j] / p.r_np) * p.r_nsrc + gi[mi][j]) * p.ldr + cbase;
              pre.r[mi][j] = *reinterpret_cast<const float4*>(rp);
              pre.r[mi][2 + j] = *reinterpret_cast<const float4*>(rp + (n0w + 64 <= p.N ? 32 : 0));
            }
        } else {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              int m = m_base + mi * 16 + j * 8 + (lane >> 3);
              if (m > p.M - 1) m = p.M - 1;
              const float* rp = R + (long)m * p.ldr + cbase;
              pre.r[mi][j] = *reinterpret_cast<const float4*>(rp);
              pre.r[mi][2 + j] = *reinterpret_cast<const float4*>(rp + (n0w + 64 <= p.N ? 32 : 0));
            }
        }
        pre.valid = inter;
        scratch = smem + LDS_BYTES + wave * 2048;     // host: 16 KiB more dynamic LDS for these instantiations
      }
      epilogue_tile<EPI, SF32, MI, true>(p, acc, m_base, n0w, l15, g, pre, 0, inter, scratch);
    }
    __builtin_amdgcn_sched_barrier(0);    // keep the fragment reads below the epilogue (hoisted, they cost it 64 VGPRs)
#ifdef RAJNI_GEMM_STAMPS
    const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
    if (more) read_frags(xa, wa, st);     // the next tile's K-tile 0 (landed and barrier-passed in the last step)
#ifdef RAJNI_GEMM_STAMPS
    if (p.stamps != nullptr && wave == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
      if (lane == 0) {
        unsigned long long* o = p.stamps + (size_t)tile * 4;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
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
}  // namespace f8

// =============================================================================================
// fp8 x fp8, 256(M) x 256(N) x 128(K bytes) - for wide outputs (QKV, FC1).  The 256 x 128 kernel above stages
// 48 KB per 8.4 MFLOP and is bound by that (a K step takes ~2.1k cycles where its MFMAs need 1.0k); this tile stages
// 64 KB per 16.8 MFLOP.  Its register budget is the bf16 wide tiling's: v_mfma_f32_32x32x64_f8f6f4 consumes 64 BYTES
// of K per instruction, so a K step is two half steps of 64 bytes whose fragment sets (4 X + 2 W tiles x 32 bytes per
// lane = 48 VGPRs) alternate exactly like the bf16 kernel's ks = 0 / 1 fragments; 128 accumulator registers.
//   lane (r = lane & 31, h = lane >> 5): operand = chunks 4*ks + 2h, 4*ks + 2h + 1 of row r of the 32-row tile, for
//   BOTH operands; C/D: column = r (the X row = output row m), row i = (reg & 3) + 8 (reg >> 2) + 4h (the W row).
//   W tile row feeding A-row i is 16 ((i>>2)&1) + (i&3) + 4 (i>>3), which makes a lane own the 16 CONSECUTIVE
//   output columns 16h + reg of the 32-column tile: 32-byte bf16 or 16-byte e4m3 stores per row.
// K step j (2 LDS stages of 64 KB), registers holding the ks = 0 fragments of step j on entry:
//   half 1: 8 MFMAs(j, ks0)  ||  12 LDS reads of (j, ks1)
//   s_waitcnt lgkmcnt(0) vmcnt(0) ; s_barrier
//   half 2: 8 MFMAs(j, ks1)  ||  12 LDS reads of (j+1, ks0)  ||  8 DMA pieces of step j+2 into stage j
// Epilogues: bias -> bf16 (QKV) and bias + GELU -> e4m3 (FC1).  Host contract: K % 128 == 0, K >= 384.
// =============================================================================================
#ifndef RAJNI_F8W_LINES
#define RAJNI_F8W_LINES 1
#endif
namespace f8w {
using f8::Frag;
using f8::frag_bits;
using f8::pack4_e4m3;
using f8::v8i;
using wide::key_x;
using wide::wait_step;
constexpr int BM = 256, BN = 256, KB = 128, NS = 2;
constexpr int X_BYTES = BM * KB, W_BYTES = BN * KB, STAGE_BYTES = X_BYTES + W_BYTES, LDS_BYTES = NS * STAGE_BYTES;
constexpr int XP = 4, PW = 4, PIECES = XP + PW;
constexpr int MT = 4, NT = 2;     // 32 x 32 tiles per wave: 128 rows x 64 columns
// W tile row (of 32) read by A-row i, and the swizzle key of a W tile row: the 16 rows one ds_read_b128 lane group
// touches ({0-3, 16-19, 4-7, 20-23} + 8 for the second group) must land in 16 distinct 16-byte slots of 256 bytes
__device__ __forceinline__ int w_row32(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
__device__ __forceinline__ int key_w(int row) { return ((row >> 4) & 1) * 4 + ((row >> 1) & 3); }

template <int G, bool READ, bool DMA>
__device__ __forceinline__ void sched_half() {
  if constexpr (G < MT) {
    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
    if constexpr (READ) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    if constexpr (DMA) __builtin_amdgcn_sched_group_barrier(0x020, PIECES / MT, 0);
    sched_half<G + 1, READ, DMA>();
  }
}

template <int EPI>
__global__ void __launch_bounds__(512, 2) gemm_f8_tn_wide(const GemmParams p) {
  static_assert(EPI == EPI_BIAS || EPI == EPI_GELU8, "wide fp8 tile: bias (bf16 out) or GELU (e4m3 out)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = p.total_tiles / p.tiles_n;
  auto tile_mn = [&](int t, int& tm_, int& tn_) {   // (N block, row tile, column in block): see gemm_bf16_tn_stream
    if (p.nblk >= p.tiles_n) { tm_ = t / p.tiles_n; tn_ = t - tm_ * p.tiles_n; return; }
    const int per = p.nblk * tiles_m, blk = t / per, r = t - blk * per;
    const int left = p.tiles_n - blk * p.nblk, nb = left < p.nblk ? left : p.nblk;
    const int rr = r / nb;
    tm_ = rr; tn_ = blk * p.nblk + (r - rr * nb);
  };
  const char* xp[XP];
  const char* wp[PW];
  const int r_in = lane >> 3, pch = lane & 7;
  auto point_at = [&](int tile) {
    int tm, tn;
    tile_mn(tile, tm, tn);
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int row = (wave * XP + i) * 8 + r_in;
      int m = tm * BM + row;
      if (m > p.M - 1) m = p.M - 1;   // clamp: duplicates are computed but never stored
      xp[i] = reinterpret_cast<const char*>(p.X) + (long)m * p.lda + ((pch ^ key_x(row)) << 4);
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int row = (wave * PW + i) * 8 + r_in;    // W is allocated with its rows padded to a multiple of 256
      wp[i] = reinterpret_cast<const char*>(p.W) + (long)(tn * BN + row) * p.ldw + ((pch ^ key_w(row)) << 4);
    }
  };
  auto dma_piece = [&](int q, int k0, char* dx) {
    if (q < XP)
      __builtin_amdgcn_global_load_lds(GLB_PTR(xp[q] + k0), LDS_PTR(dx + (wave * XP + q) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(GLB_PTR(wp[q - XP] + k0), LDS_PTR(dx + X_BYTES + (wave * PW + q - XP) * 1024), 16, 0, 0);
  };
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int xr0 = wm * 128 + r;                    // + 32 * mi: same key
  const int wr0 = wn * 64 + w_row32(r);            // + 32 * ni: same key
  int xo[2][2], wo[2][2];                          // [ks][chunk of the pair]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      xo[ks][c] = xr0 * KB + (((4 * ks + 2 * h + c) ^ key_x(xr0)) << 4);
      wo[ks][c] = X_BYTES + wr0 * KB + (((4 * ks + 2 * h + c) ^ key_w(wr0)) << 4);
    }

  f32x16 acc[NT][MT];
  // one half step: MFMAs on (xc, wc)  ||  if READ: fragments (stage rst, half rks) -> (xn, wn_)  ||  if DMA: K-tile dkt
  // of the pointed-at tile -> stage dst
  auto half = [&](auto read_c, auto dma_c, Frag (&xc)[MT], Frag (&wc)[NT], Frag (&xn)[MT], Frag (&wn_)[NT],
                  int rst, int rks, int dkt, int dst) {
    constexpr bool READ = decltype(read_c)::value, DMA = decltype(dma_c)::value;
    const char* sb = smem + rst * STAGE_BYTES;
    char* dx = smem + dst * STAGE_BYTES;
    const int k0 = dkt * KB;
    v8i wv[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) wv[ni] = frag_bits(wc[ni]);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const v8i xv = frag_bits(xc[mi]);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv[ni], xv, acc[ni][mi], 0, 0, 0, 0, 0, 0);
      if constexpr (READ) {
        xn[mi].lo = *reinterpret_cast<const bf16x8*>(sb + xo[rks][0] + mi * 32 * KB);
        xn[mi].hi = *reinterpret_cast<const bf16x8*>(sb + xo[rks][1] + mi * 32 * KB);
        if (mi < NT) wn_[mi].lo = *reinterpret_cast<const bf16x8*>(sb + wo[rks][0] + mi * 32 * KB);
        else wn_[mi - NT].hi = *reinterpret_cast<const bf16x8*>(sb + wo[rks][1] + (mi - NT) * 32 * KB);
      }
      if constexpr (DMA) {
#pragma unroll
        for (int q = mi * PIECES / MT; q < (mi + 1) * PIECES / MT; ++q) dma_piece(q, k0, dx);
      }
    }
    sched_half<0, READ, DMA>();
  };
  using T = std::true_type; using F = std::false_type;

  const int nk = p.K / KB;   // >= NS + 1 (host checked)
  int v = blockIdx.x;
  int tile = xcd_tile_of(v, p.total_tiles);
  Frag xa[MT], wa[NT], xb[MT], wb[NT];
  auto read_ks0 = [&](int st) {
    const char* sb = smem + st * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      xa[i].lo = *reinterpret_cast<const bf16x8*>(sb + xo[0][0] + i * 32 * KB);
      xa[i].hi = *reinterpret_cast<const bf16x8*>(sb + xo[0][1] + i * 32 * KB);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      wa[i].lo = *reinterpret_cast<const bf16x8*>(sb + wo[0][0] + i * 32 * KB);
      wa[i].hi = *reinterpret_cast<const bf16x8*>(sb + wo[0][1] + i * 32 * KB);
    }
  };
  constexpr int NSTORE = EPI == EPI_GELU8 ? MT * NT : 2 * MT * NT;   // stores per wave of an interior tile's epilogue
  bool prev_full = false;
  point_at(tile);
#pragma unroll
  for (int j = 0; j < NS; ++j) {
#pragma unroll
    for (int q = 0; q < PIECES; ++q) dma_piece(q, j * KB, smem + j * STAGE_BYTES);
  }
  wait_step<PIECES>();   // K tile 0 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_ks0(0);
  int st = 0;

  while (true) {
#ifdef RAJNI_GEMM_STAMPS
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    int tm, tn;
    tile_mn(tile, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const bool inter = m0 + BM <= p.M && n0 + BN <= p.N;
    const int vn = v + gridDim.x;
    const bool more = vn < p.total_tiles;
    const int m_base = m0 + wm * 128, n0w = n0 + wn * 64;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    float xsr[MT];

    // the last K step reads no (j+1, ks0) fragments: the next tile's first fragments are read after the epilogue
    // (one exposed LDS round trip per tile; with a fragment set live through it the epilogue's operands do not fit)
    auto kstep = [&](auto last_c, int kt) {
      constexpr bool LAST = decltype(last_c)::value;
      if (kt == nk - NS && more) point_at(xcd_tile_of(vn, p.total_tiles));
      const int dkt = kt + NS < nk ? kt + NS : kt + NS - nk;
      const int st1 = st ^ 1;
      if constexpr (LAST) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {   // on every path (clamped): see the note on conditional loads above
          const int m = m_base + mi * 32 + r;
          xsr[mi] = p.xscale[m < p.M ? m : p.M - 1];
        }
      }
      half(T{}, F{}, xa, wa, xb, wb, st, 1, 0, 0);
      if (kt == 0 && prev_full) wait_step<NSTORE>();
      else wait_step<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      using RD = std::integral_constant<bool, !LAST>;
      half(RD{}, T{}, xb, wb, xa, wa, st1, 0, dkt, st);
      st = st1;
    };
    for (int kt = 0; kt < nk - 1; ++kt) kstep(F{}, kt);
    kstep(T{}, nk - 1);
#ifdef RAJNI_GEMM_STAMPS
    asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[NT - 1][MT - 1][15]));
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue
    __builtin_amdgcn_sched_barrier(0);
    float inv[MT];
    if constexpr (EPI == EPI_GELU8) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int m = m_base + mi * 32 + r;
        inv[mi] = 1.0f / p.yscale[m < p.M ? m : p.M - 1];
      }
    }
    bool lines_done = false;
    if constexpr (EPI == EPI_BIAS && RAJNI_F8W_LINES) {
      if (inter) {
        // interior tile, bf16 output as whole 128-byte lines: per 32-row group the wave's 32 x 64 block (two 16-byte pieces
        // per lane and column tile: chunks 4 ni + 2 h, + 1 of row r) goes through 4 KiB of LDS behind the stages - slot =
        // chunk ^ (row & 7), conflict free both ways - and leaves as 8 rows x 128 bytes per instruction.  Stored
        // straight from the accumulator layout an instruction touches 32 rows x 64 bytes (two lanes per row): half lines,
        // which a CU moves at a third of the rate (tools/pull_probe.hip) - the reason QKV used the 256 x 128 kernel.
        char* scratch = smem + LDS_BYTES + wave * 4096;
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int rr = lane_v >> 3, cc = lane_v & 7;
        float bs[NT][16], ws[NT][16];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const int c0 = n0w + 32 * ni + 16 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 wq = *reinterpret_cast<const float4*>(p.wscale + c0 + 4 * q);
            const float4 bq = p.bias != nullptr ? *reinterpret_cast<const float4*>(p.bias + c0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            ws[ni][4 * q] = wq.x; ws[ni][4 * q + 1] = wq.y; ws[ni][4 * q + 2] = wq.z; ws[ni][4 * q + 3] = wq.w;
            bs[ni][4 * q] = bq.x; bs[ni][4 * q + 1] = bq.y; bs[ni][4 * q + 2] = bq.z; bs[ni][4 * q + 3] = bq.w;
          }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) on every path: see epilogue_tile
        bf16_t* Y = reinterpret_cast<bf16_t*>(p.Y) + n0w + 8 * cc;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int key = r & 7;
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) {
            float y[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) y[j] = fmaf(acc[ni][mi][j] * xsr[mi], ws[ni][j], bs[ni][j]);
            *reinterpret_cast<bf16x8*>(scratch + r * 128 + (((4 * ni + 2 * h) ^ key) << 4)) = __builtin_bit_cast(bf16x8, pack8(y));
            *reinterpret_cast<bf16x8*>(scratch + r * 128 + (((4 * ni + 2 * h + 1) ^ key) << 4)) = __builtin_bit_cast(bf16x8, pack8(y + 8));
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = 8 * j + rr;
            const uint4 q = __builtin_bit_cast(uint4, *reinterpret_cast<const bf16x8*>(scratch + row * 128 + ((cc ^ (row & 7)) << 4)));
            *reinterpret_cast<uint4*>(Y + (long)(m_base + mi * 32 + row) * p.ldc) = q;
          }
          __builtin_amdgcn_wave_barrier();   // the group's reads are issued before the next group's writes
        }
        lines_done = true;
      }
    }
    if (!lines_done)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int c0 = n0w + 32 * ni + 16 * h;       // this lane's 16 consecutive columns of the tile
      const bool full = c0 + 16 <= p.N;
      float bs[16], ws[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int n = c0 + j < p.N ? c0 + j : p.N - 1;
        bs[j] = p.bias != nullptr ? p.bias[n] : 0.f;
        ws[j] = p.wscale[n];
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) on every path: see epilogue_tile
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int m = m_base + mi * 32 + r;
        if (m >= p.M) continue;
        float y[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) y[j] = fmaf(acc[ni][mi][j] * xsr[mi], ws[j], bs[j]);
        if constexpr (EPI == EPI_GELU8) {
#pragma unroll
          for (int j = 0; j < 16; j += 4) {
            f32x2 ta = f32x2{y[j], y[j + 1]}, tb = f32x2{y[j + 2], y[j + 3]};
            gelu_pk4(ta, tb);
            y[j] = ta[0] * inv[mi]; y[j + 1] = ta[1] * inv[mi]; y[j + 2] = tb[0] * inv[mi]; y[j + 3] = tb[1] * inv[mi];
          }
          uint4 q;
          q.x = pack4_e4m3(y[0], y[1], y[2], y[3]);   q.y = pack4_e4m3(y[4], y[5], y[6], y[7]);
          q.z = pack4_e4m3(y[8], y[9], y[10], y[11]); q.w = pack4_e4m3(y[12], y[13], y[14], y[15]);
          unsigned char* row = reinterpret_cast<unsigned char*>(p.Y) + (long)m * p.ldc + c0;
          if (full) {
            *reinterpret_cast<uint4*>(row) = q;
          } else {
            const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 16; ++j)
              if (c0 + j < p.N) row[j] = (unsigned char)(w4[j >> 2] >> (8 * (j & 3)));
          }
        } else {
          bf16_t* row = reinterpret_cast<bf16_t*>(p.Y) + (long)m * p.ldc + c0;
          if (full) {
            *reinterpret_cast<uint4*>(row) = pack8(y);
            *reinterpret_cast<uint4*>(row + 8) = pack8(y + 8);
          } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
              if (c0 + j < p.N) row[j] = f2bf(y[j]);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef RAJNI_GEMM_STAMPS
    const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
    if (more) read_ks0(st);      // the next tile's (K-tile 0, ks0): landed and barrier-passed in the last step
#ifdef RAJNI_GEMM_STAMPS
    if (p.stamps != nullptr && wave == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
      if (lane == 0) {
        unsigned long long* o = p.stamps + (size_t)tile * 4;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
      }
    }
#endif
    prev_full = inter;
    if (!more) break;
    v = vn;
    tile = xcd_tile_of(v, p.total_tiles);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
}  // namespace f8w
