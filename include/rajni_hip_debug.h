/* rajni_hip_debug.h - test and tuning hooks of librajni_hip.so.  NOT part of the drop-in boundary
 * (include/rajni_hip.h): nothing on the product path calls these.  They set process-global, unsynchronised
 * switches, so they must not be flipped while another thread is inside a rajni_* call; the tests and the
 * probes under tools/ use them from one thread, before the launch they want to steer. */
#ifndef RAJNI_HIP_DEBUG_H
#define RAJNI_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* attention kernel choice: 0 = by np (default: persistent full-row kernel for np <= 256), 1 = chunked
 * online-softmax kernel, 2 = one-shot full-row kernel (np <= 256) */
void rajni_debug_force_attention(int mode);

/* GEMM tiling: 0 = by shape (default), 1 = 128x128x64 (4 waves), 4 = 256x256x64 persistent,
 * 5 = 256x128x64 3-stage persistent */
void rajni_debug_force_gemm_tiling(int mode);

/* fp8 x fp8 GEMM tiling: 0 = by shape (default), 1 = 256x128x128 always, 2 = 256x256x128 wherever it exists (bias and
 * GELU epilogues) */
void rajni_debug_force_f8_tiling(int mode);

/* every other workgroup of an XCD sleeps `units` x 8192 cycles before its first tile of a residual-epilogue GEMM, so
 * that the two halves of the chip do not burst in the same instant; default 2 (+0.5-0.8 % on the forward), 0 = off */
void rajni_debug_set_resid_stagger(int units);

/* W bytes one N block of the persistent tile order may occupy (default 1600 KiB); 0 = the plain column-fastest
 * order; -k = blocks of k column tiles regardless of size.  Results are bit-identical for every value (tested). */
void rajni_debug_set_gemm_nblock_bytes(int bytes);

/* score+select: 1 = read K and V in two passes with vbar reusing the logits' LDS region (what N = 577 x 16 heads
 * needs) even when the one-pass layout fits; 0 = default.  Scores are bit-identical either way (tested). */
void rajni_debug_force_score_two_pass(int on);

/* diagnostic builds (-DRAJNI_GEMM_STAMPS / -DRAJNI_ATTN_STAMPS / -DRAJNI_SS_STAMPS) only: device buffer receiving
 * 4 x uint64 s_memtime stamps per workgroup; NULL disables */
void rajni_debug_set_gemm_stamps(void* buf);

#ifdef __cplusplus
}
#endif
#endif
