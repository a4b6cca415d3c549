/*
 * isr_hip.h — C ABI of libisr_hip.so, the MI355X (gfx950) implementation of the
 * image-sequence-registration hot path.
 *
 * The reference (Kudo510/ImageSequenceRegistrationfor6DPoseEstimationLabeling) has no FFI:
 * its boundary is a handful of Python functions that call torch / OpenCV / Open3D / sklearn.
 * Each entry point below replaces one of those library call sites; the citation is the
 * reference file:line whose arithmetic the entry point takes over.  The Python mirror in
 * imagesequenceregistrationfor6dposeestimationlabeling_amd/registration.py binds these with
 * ctypes and keeps the reference's function names and signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *   - nothing is allocated for the caller: scratch comes from `ws` (size from *_workspace_bytes);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises;
 *   - return value: ISR_OK or a negative ISR_ERR_*; isr_last_error() has the text;
 *   - counts that are produced on the device (M, number of inliers, status) stay on the device
 *     so a whole per-image pipeline can be enqueued without a host round trip.
 */
#ifndef ISR_HIP_H
#define ISR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* v4 (round 4): isr_corr_argmax accepts idx = logp = NULL (lse-only call), f32 rows run as f16 planes by default
 * (ISR_TUNE_K1_F32_CHAIN values 0-4), new isr_corr_topk / isr_corr_topk_workspace_bytes.
 * v5 (round 5): ISR_DTYPE_BF16_LOG2_SCREENED, isr_corr_argmax_screen_redone; isr_corr_topk_workspace_bytes sized by the key
 * ranges a call uses; isr_corr_argmax_digits + isr_select_top_batch_digits (the cut's first histogram from K1's epilogue); isr_corr_argmax_phase (a call's closing kernels on another stream). */
#define ISR_ABI_VERSION 5

#define ISR_OK 0
#define ISR_ERR_ARG (-1)         /* bad shape / null pointer / unsupported value */
#define ISR_ERR_WORKSPACE (-2)   /* ws_bytes smaller than *_workspace_bytes says */
#define ISR_ERR_HIP (-3)         /* a HIP runtime call or launch failed */
#define ISR_ERR_UNSUPPORTED (-4) /* valid request this build does not implement */

#define ISR_DTYPE_BF16 0 /* bf16 inputs, v_mfma_f32_32x32x16_bf16, f32 accumulate */
#define ISR_DTYPE_F32 1  /* f32 inputs; the index is the arg-max of the k-ordered f32 fmaf-chain logits (lowest key on ties).
                            D <= 128 (default): the rows run on the 16-bit matrix cores as f16 planes (x1 | x2s | x1s, three plane
                            pairs per 16-wide block, f32-accurate sums), margin test + recheck by the f32 chain of the ORIGINAL rows;
                            a call holding an |element| >= 65 000 falls through, on the device, to v_mfma_f32_32x32x2_f32, whose
                            accumulation IS that chain.  ISR_TUNE_K1_F32_CHAIN selects the other routes (same indices). */
#define ISR_DTYPE_BF16_LOG2 2 /* bf16 inputs whose QUERIES were multiplied by log2(e) before their one
                                rounding to bf16: logits are in log2 units inside the kernel (exp2 + add
                                per element, nothing else) — the fastest path; outputs stay natural-log */

#define ISR_DTYPE_BF16_LOG2_SCREENED 3 /* ISR_DTYPE_BF16_LOG2 rows behind a rigorous low-precision screen (round 5; D = 64, N <= 262 144 —
                                other shapes run the ISR_DTYPE_BF16_LOG2 kernels).  The rows are ALSO held as block-scaled FP6; one
                                v_mfma_scale_f32_32x32x64_f8f6f4 per 32 x 32 tile gives approximate logits with a proven error bound,
                                and a (16-key lane group, 32-key tile) PIECE enters the log-sum-exp only when its largest exact logit
                                reaches L_q - T:  L_q a true logit of the query found by a first FP6 pass (its maximum wherever the
                                softmax has a clear winner), T = 21 + ceil(log2 N) log2 units.  Pieces the screen proves to lie below
                                that are never formed; the others are formed on the bf16 matrix cores from the original rows.
                                idx: as ISR_DTYPE_BF16_LOG2 (exact).  logp / lse: the left-out pieces sum to < 2^-21 of the total (5e-7
                                in lse); a function of (query, keys) only, bit-identical whatever else is in the launch.  For data
                                whose softmax is peaked (descriptor matching); on flat logits nothing can be skipped and the call
                                costs about 1.5 x the unscreened one (a first pass, then the dense kernel with the same rule). */

typedef void* isr_stream_t; /* hipStream_t */

int isr_abi_version(void);
const char* isr_last_error(void);
/* Number of HIP devices visible to the library's runtime (0 on a CPU-only host, no error). */
int isr_device_count(void);

/* Tuning knobs (experiments and tests only; every knob at its default selects the shipped plan).  Process-wide
 * atomics, read by the entry points when they plan a launch: no environment look-ups on the call path.  The
 * initial values are read ONCE, when the library is loaded, from the ISR_* environment variables named below.
 * isr_tuning_set returns ISR_OK (ISR_ERR_ARG for an unknown knob), isr_tuning_get the value (0 for an unknown
 * knob); all choices give bit-identical results (the NN paths are interchangeable by construction and tested to be). */
#define ISR_TUNE_NN_PATH 0       /* -1 auto | 0 brute force | 1 per-lane grid | 2 block-cooperative grid   ISR_NN_GRID */
#define ISR_TUNE_NN_FILTER 1     /* -1 auto | 0 plain loop | 1 filter loop (cold searches)                  ISR_NN_FILTER */
#define ISR_TUNE_ICP_WARM 2      /* 1 warm-started ICP passes (default) | 0 every pass cold                 ISR_ICP_WARM */
#define ISR_TUNE_NN_PLAN_RQ 3    /* 0 auto | 1 | 4 queries per lane                                         ISR_NN_PLAN="rq,blocks" */
#define ISR_TUNE_NN_PLAN_BLOCKS 4 /* 0 auto | wanted workgroups per launch */
#define ISR_TUNE_NN_TILE_ST 5    /* 0 default | target cell scale x 1000                                    ISR_NN_TILE="st,sq,tb" */
#define ISR_TUNE_NN_TILE_SQ 6    /* 0 default | query cell scale x 1000 */
#define ISR_TUNE_NN_TILE_TB 7    /* 0 default | 64 | 128 | 256 threads per workgroup */
#define ISR_TUNE_EP_WSUM_VALU 8  /* 0 (default) the sampler's chunk sums form their logits on the f32 MFMA | 1 on VALU fma chains (same bits) */
#define ISR_TUNE_K1_F32_CHAIN 9  /* f32 queries: 0 / 3 (default) f16 planes on the matrix cores, exact f32-chain recheck, falls through to the chain kernel when a descriptor does not fit f16 | 1 the f32-MFMA chain kernel | 2 three bf16 planes (D <= 64; chain kernel above) | 4 round 3 96-wide rows (D <= 16) */
#define ISR_TUNE_K1_SPLIT 10     /* 0 (default) K1 picks its number of key ranges | n > 0 forced (capped); results do not depend on it */
#define ISR_TUNE_COUNT 11
int isr_tuning_set(int knob, int value);
int isr_tuning_get(int knob);

/* ------------------------------------------------------------------------------------------
 * K1  feature correlation: getCors(queries, feats, leaves=1)
 * replaces  torch.log_softmax(queries @ feats.T, -1) + torch.topk(k=1)
 *           inference.py:142-149 (= finalposes.py:38-45 = choosePose.py:35-42),
 *           and the logsumexp-only use at pose_refine.py:56 (ask for `lse`).
 * Q (P, ldq) and K (N, ldk) row-major, element type by `dtype`; the first D columns are used.
 * D: 16, 32, 64 or 128 (bf16: zero-pad the columns) / any D <= 128 (f32); ldq, ldk >= D and (bf16) rows
 * 16-byte aligned.
 * Outputs, per query row p — functions of (Q[p], K) only, bit-identical whatever else is in the launch:
 *   idx[p]  = argmax_n <Q[p],K[n]>   (lowest n on ties).  f32: of the k-ordered fmaf-chain logits; bf16:
 *             of the EXACT logits (products of bf16 values summed without error: queries whose top-2
 *             margin is inside the f32 accumulation error bound are re-decided in exact arithmetic)
 *   logp[p] = max_n logit - logsumexp_n logit   (= the top-1 value of log_softmax)
 *   lse[p]  = logsumexp_n logit      (nullable)
 * idx = logp = NULL with lse given: an lse-only call (pose_refine.py:56, poseEstSurf.py:68-71) — no maxima tracked, no index
 * certified; lse carries the bits of a full call's.
 * f32 (round 4): the rows run on the 16-bit matrix cores as f16 planes (x1 | x2s | x1s, three plane pairs per
 * 16-wide block) with a margin test and a recheck by the k-ordered f32 fmaf chain of the ORIGINAL rows — idx is the chain's
 * arg-max bit for bit, logp / lse are accurate to f32 (2^-21 |q||k|); descriptors must be finite, and a call holding an
 * |element| >= 65 000 runs the f32-MFMA chain kernels instead (decided on the device).  ISR_TUNE_K1_F32_CHAIN selects the others.
 * The (P x N) matrix is never materialised.
 */
size_t isr_corr_argmax_workspace_bytes(int P, int N, int D, int dtype);
int isr_corr_argmax(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk, int dtype,
                    int32_t* idx, float* logp, float* lse, void* ws, size_t ws_bytes,
                    isr_stream_t stream);

/* getCors for a GROUP of images whose top-80 % cut follows (inference.py:142-149 then :282-290): the same
 * call, the same outputs bit for bit, plus — formed where each logp is written, so the cut does not read logp once more for it —
 * the first histogram of the cut's radix select.  The P rows are P / rows_per_image images of rows_per_image rows each
 * (P % rows_per_image == 0); of image b the first n_rows[b] rows count (n_rows NULL: all of them; the padding rows of a
 * capacity-sized crop batch do not).  digit_hist (P / rows_per_image, 2048) int32 is ZEROED by the call, then
 *     digit_hist[b][u(logp[q]) >> 21] += 1   for every counted row q of image b,
 * u = the order-preserving unsigned image of a float (sign bit flipped for x >= 0, all bits for x < 0).  Integer atomics:
 * the histogram is a function of logp alone.  idx, logp and digit_hist are required; every dtype and route of
 * isr_corr_argmax.  Hand digit_hist to isr_select_top_batch_digits. */
int isr_corr_argmax_digits(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk, int dtype,
                           int32_t* idx, float* logp, float* lse, int rows_per_image, const int32_t* n_rows,
                           int32_t* digit_hist, void* ws, size_t ws_bytes, isr_stream_t stream);

/* isr_corr_argmax in two halves, for callers that launch it back to back (one call per group of images).
 * phase 1 ("open"):  pre-processing, key norms and the chip-filling kernel(s);
 * phase 2 ("close"): the closing kernels — fallback for out-of-range queries, finalize, exact recheck, merge: ~170 us of small
 *                    launches at low occupancy;
 * phase 3: both (= isr_corr_argmax / isr_corr_argmax_digits).
 * Both halves take the SAME arguments and the SAME workspace; the close must be ordered behind its open (same stream, or an
 * event), the outputs are complete when the close has run, and the workspace must not be opened again before its close has
 * finished (alternate two workspaces).  Putting the close on another stream lets the next group's chip-filling kernel follow
 * this one's directly instead of behind the closing kernels (bench: +0.4 %, profiles/r05_k1_call_split_ab.txt).  Results do not depend on how a call is split.
 * digit_hist may be NULL (then rows_per_image / n_rows are ignored). */
int isr_corr_argmax_phase(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk, int dtype,
                          int32_t* idx, float* logp, float* lse, int rows_per_image, const int32_t* n_rows,
                          int32_t* digit_hist, int phase, void* ws, size_t ws_bytes, isr_stream_t stream);

/* Diagnostics: number of queries the last isr_corr_argmax call on this workspace (same P, N, dtype)
 * decided by the exact recheck; -1 for ISR_DTYPE_F32.  count_host is a HOST pointer; synchronises. */
int isr_corr_argmax_recheck_count(const void* ws, size_t ws_bytes, int P, int N, int dtype,
                                  int32_t* count_host, isr_stream_t stream);
/* the same for an f32 call with D <= 16 (the split route's f32-chain recheck list; -1 when that route is not taken) */
int isr_corr_argmax_recheck_count_f32(const void* ws, size_t ws_bytes, int P, int N, int D, int32_t* count_host,
                                      isr_stream_t stream);

/* Parity hook (ISR_DTYPE_BF16_LOG2_SCREENED): the block-scaled FP6 image of R bf16 rows of 64 columns, exactly as isr_corr_argmax
 * forms it.  out (R, 64) bytes: per 32-element half 24 B of e2m3 codes (element j in bits [6j, 6j + 6) of the little-endian
 * stream; bit 5 the sign), byte 24 the E8M0 scale (2^(b - 127), the smallest power of two with max|x| / scale <= 7.5), 7 zero
 * bytes.  nrm (R, 2) f32 = {|x|, |x - x~|} inflated by 1.00001; kmax (2) f32 = max over rows of |x - x~|^2 and |x~|^2.  All on
 * the device; X rows 16-byte aligned. */
int isr_corr_quantize_fp6(const void* X, int R, int ld, void* out, float* nrm, float* kmax, isr_stream_t stream);

/* Diagnostics (ISR_DTYPE_BF16_LOG2_SCREENED): count_host[0] = how many (32-query block, 32-key tile) items the last
 * isr_corr_argmax call on this workspace fetched again and redid on the bf16 matrix cores behind its FP6 screen,
 * count_host[1] = how many 256-query blocks it handed to the dense kernel; zeros on every other route.  HOST pointer
 * (two int64). */
int isr_corr_argmax_screen_redone(const void* ws, size_t ws_bytes, int P, int N, int dtype, long long* count_host,
                                  isr_stream_t stream);

/* Diagnostics: the shader clock (MHz) the bf16 kernel held during the last isr_corr_argmax call on this
 * workspace, from s_memtime / s_memrealtime over the life of one workgroup; 0 for ISR_DTYPE_F32. */
int isr_corr_argmax_clock_mhz(const void* ws, size_t ws_bytes, int P, int N, int dtype, double* mhz_host,
                              isr_stream_t stream);

/* K1 materialising variant for small P: out (P, N) f32 = log_softmax(Q K^T) row-wise.
 * replaces poseEstSurf.py:70 (corr_matrix_log) and getCors with leaves > 1 (caller runs topk).
 * f32, D <= 128: the row log-sum-exps come from isr_corr_argmax's exact-f32 path, then one pass writes
 * logit - lse (HBM-bound on the 4 P N output bytes); ws from isr_corr_logsoftmax_workspace_bytes. */
size_t isr_corr_logsoftmax_workspace_bytes(int P, int N, int D, int dtype);
int isr_corr_logsoftmax(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk,
                        int dtype, float* out, int64_t ldo, void* ws, size_t ws_bytes, isr_stream_t stream);

/* K1 with leaves > 1: getCors(queries, feats, leaves) = topk(log_softmax(queries @ feats.T), leaves) (inference.py:145-149)
 * without the (P x N) matrix.  f32 rows, D <= 128, 1 <= k <= 8; lse (P) = the rows' log-sum-exps from an lse-only
 * isr_corr_argmax call on the same rows.  Per query the k largest k-ordered-fmaf-chain logits (the values
 * isr_corr_logsoftmax writes), descending, equal values by ascending key: idx (P, k) int32 (-1 where N < k), vals (P, k) =
 * logit - lse. */
size_t isr_corr_topk_workspace_bytes(int P, int N);
int isr_corr_topk(const float* Q, const float* K, int P, int N, int D, int ldq, int ldk, int k, const float* lse,
                  int32_t* idx, float* vals, void* ws, size_t ws_bytes, isr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * hand-off from the descriptor network to K1
 * replaces  imfeats[:, ::3, ::3]; inputMask[::3, ::3]; maskIds = torch.where(inputMask);
 *           maskedfeats = imfeats[0][maskIds]; ep2d[:,0] = maskIds[1]; ep2d[:,1] = maskIds[0]
 *           inference.py:248-279
 * feat (H, W, C) f32 channels-last (the network output after movedim), descriptor = channels
 * [c0, c0 + D); mask (H, W) bytes, `mask_pix_stride` bytes between pixels (3 for cropMask[:, :, 0] of
 * an interleaved BGR mask), non-zero = inside; every `step`-th row and column.
 * Q: capacity S = ceil(H/step) * ceil(W/step) rows of ldq elements (bf16 or f32 by dtype; with
 *   ISR_DTYPE_BF16_LOG2 the values are multiplied by log2(e) before their one rounding), masked
 *   pixels compacted in row-major order, all other rows and the columns [D, ldq) zero.
 * pix_xy (S, 2) f32 = (column, row) in the subsampled grid; *n_dev = number of masked pixels. */
size_t isr_prep_queries_workspace_bytes(int H, int W, int step);
int isr_prep_queries(const float* feat, int H, int W, int C, int c0, int D, const uint8_t* mask,
                     int mask_pix_stride, int step, int dtype, int ldq, void* Q, float* pix_xy,
                     int32_t* n_dev, void* ws, size_t ws_bytes, isr_stream_t stream);

/* The same for a GROUP of B crops as one chain of three launches (image = blockIdx.z): the per-image loop of
 * inference.py:163, 248-279 batched.  feat (B, H, W, C), mask (B, H, W) x mask_pix_stride bytes per pixel;
 * Q (B, S, ldq), pix_xy (B, S, 2), n_dev (B).  Rows past n_dev[b] of image b are zero queries: K1 runs ONCE
 * over the B * S capacity rows (a query's result does not depend on the launch it rides in), and
 * isr_select_top_batch / isr_gather_corr_batch / isr_pnp_ransac_batch take the ragged counts from n_dev.
 * Same kernels as isr_prep_queries: image b's rows are bit-identical to the single-image call. */
size_t isr_prep_queries_batch_workspace_bytes(int H, int W, int step, int B);
int isr_prep_queries_batch(const float* feat, int B, int H, int W, int C, int c0, int D, const uint8_t* mask,
                           int mask_pix_stride, int step, int dtype, int ldq, void* Q, float* pix_xy,
                           int32_t* n_dev, void* ws, size_t ws_bytes, isr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * image front end of the per-image loop, image = blockIdx.z: a group costs two launches
 * isr_mask_bbox       x, y, w, h = cv2.boundingRect(mask[:, :, 0])         inference.py:202
 *   mask (B, H, W, C) u8, channel 0 -> bbox_dev (B, 4) i32 {x, y, w, h}; all-zero mask -> {0, 0, 0, 0}.
 * isr_crop_normalize  cropRGB = cv2.warpAffine(rgb, M, (r, r)); cropMask = cv2.warpAffine(mask, M, (r, r));
 *   if useMask: cropRGB[cropMask[:, :, 0] == 0] = 0; normalize(cropRGB).astype(f32)   inference.py:224-232, 135-141
 *   rgb (B, H, W, 3) u8, mask (B, H, W, mask_channels) u8; M_host: HOST (B, 6) f64, the reference's 2x3 M
 *   (source -> crop; inverted here as warpAffine does); mean3 / std3 HOST (3) f64;
 *   out (B, 3, r, r) f32 channels-first (the network input), crop_mask (B, r, r) u8 = cropMask[:, :, 0].
 *   Bilinear warp as defined in csrc/crop_normalize.hip (OpenCV is absent: parity unpinned). */
int isr_mask_bbox(const uint8_t* mask, int B, int H, int W, int C, int32_t* bbox_dev, isr_stream_t stream);
int isr_crop_normalize(const uint8_t* rgb, const uint8_t* mask, int B, int H, int W, int mask_channels,
                       const double* M_host, int out_size, int use_mask, const double* mean3,
                       const double* std3, float* out, uint8_t* crop_mask, isr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a2  top-80 % correspondence filter
 * replaces  torch.sort(in1[:,0])[0][-perc+1] ; torch.where(in1[:,0] > thr)   inference.py:282-290
 * n = P; if n > min_n: perc = (int)(frac*n), rank = n - perc + 1 else rank = 1  (0-based rank into
 * the ascending order); thr = rank-th smallest logp; keep = ascending indices p with logp[p] > thr.
 * keep has capacity P; *M_dev (device int32) receives the number kept; *thr_dev (nullable) thr.
 */
size_t isr_select_top_workspace_bytes(int P);
int isr_select_top(const float* logp, int P, double frac, int min_n, int32_t* keep, int32_t* M_dev,
                   float* thr_dev, void* ws, size_t ws_bytes, isr_stream_t stream);

/* The same cut when the element count lives on the device (after isr_prep_queries): the first
 * min(P_cap, *n_dev) values of logp are the input; a count of 0, or a rank the reference would
 * raise IndexError for, keeps nothing (*M_dev = 0, thr = +inf).  Workspace as for P_cap. */
int isr_select_top_dev(const float* logp, int P_cap, const int32_t* n_dev, double frac, int min_n,
                       int32_t* keep, int32_t* M_dev, float* thr_dev, void* ws, size_t ws_bytes,
                       isr_stream_t stream);

/* The cut for a GROUP of B images in one chain of ten launches (the per-image loop of inference.py:163
 * over the group): logp (B, P), keep (B, P), M_dev (B), thr_dev (B, nullable); n_dev (B) device counts
 * or NULL (every image has P values).  Same kernels as the single-image calls: identical results. */
size_t isr_select_top_batch_workspace_bytes(int P, int B);
int isr_select_top_batch(const float* logp, int P, int B, const int32_t* n_dev, double frac, int min_n,
                         int32_t* keep, int32_t* M_dev, float* thr_dev, void* ws, size_t ws_bytes,
                         isr_stream_t stream);
/* The same cut with its first histogram supplied: digit_hist (B, 2048) as isr_corr_argmax_digits left it for these logp
 * (rows_per_image = P, n_rows = n_dev).  Nine launches instead of ten and one read of logp less; results identical to
 * isr_select_top_batch's (same later passes on the same values).  Workspace: isr_select_top_batch_workspace_bytes. */
int isr_select_top_batch_digits(const float* logp, int P, int B, const int32_t* n_dev, double frac, int min_n,
                                const int32_t* digit_hist, int32_t* keep, int32_t* M_dev, float* thr_dev, void* ws,
                                size_t ws_bytes, isr_stream_t stream);

/* a3  correspondence assembly  (inference.py:274-280, 289-290)
 * p3d[m] = pts[idx[keep[m]]], p2d[m] = pix_xy[keep[m]]  for m < *M_dev.  pts (N,3), pix_xy (P,2)
 * = (col,row) of every query pixel, both f32.  p3d (P,3) / p2d (P,2) have capacity P rows. */
int isr_gather_corr(const int32_t* idx, const int32_t* keep, const int32_t* M_dev, int P,
                    const float* pts, int N, const float* pix_xy, float* p3d, float* p2d,
                    isr_stream_t stream);

/* a3 for a group of B images: idx, keep (B, P), M_dev (B), p3d (B, P, 3), p2d (B, P, 2); pix_xy is
 * (P, 2) shared by all images (shared_pix != 0, e.g. a full pixel grid) or (B, P, 2). */
int isr_gather_corr_batch(const int32_t* idx, const int32_t* keep, const int32_t* M_dev, int P, int B,
                          const float* pts, int N, const float* pix_xy, int shared_pix, float* p3d,
                          float* p2d, isr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K2  PnP + RANSAC:  pnp(h3d, h2d, cam, itr, reperr, P3P)
 * replaces  cv2.solvePnPRansac(..., iterationsCount=itr, reprojectionError=reperr,
 *           flags=SOLVEPNP_P3P) + cv2.Rodrigues      inference.py:123-134, call site :293
 * (OpenCV is not available to this build: the algorithm is owned and documented in DESIGN.md.)
 *
 * isr_p3p_hypotheses: hypothesis h draws 4 correspondence indices from Philox4x32-10(seed, h),
 *   solves P3P (f64) on the first three, keeps the root with the smallest reprojection error on
 *   the fourth.  Rt (H,12) f64 row-major [R|t]; ok (H) u8; sample (H,4) i32 (nullable).
 * isr_ransac_score: n_inl[h] = #{m : z>0 and |proj_h(p3d[m]) - p2d[m]|^2 <= reperr^2} in f32,
 *   evaluated division-free; best_dev = argmax (lowest h on ties, ok hypotheses only),
 *   best_mask = inlier bitmask of the best hypothesis, ceil(M_cap/32) words (nullable);
 *   ws as for isr_pnp_ransac.
 * isr_pnp_refine: `iters` Gauss-Newton steps (f64) on the reprojection error over the masked
 *   correspondences, starting from Rt_io (12 f64), result written back.
 * isr_pnp_ransac: the three above + inlier index compaction, one enqueue.  `confidence` is
 *   cv2.solvePnPRansac's parameter of that name (its default, which the reference uses, is 0.99):
 *   hypotheses are scored in stages [0,32), [32,96), [96,224), ... and a stage runs only while
 *   (1 - (c/M)^4)^b > 1 - confidence for the best count c after the b hypotheses before it;
 *   confidence >= 1 scores all H (isr_ransac_score always does).
 *   pose_dev: 12 f64 [R|t];  inl_idx: capacity M_cap;  n_inl_dev: i32;  status_dev: i32
 *   (1 = pose found, 0 = failed: the Python mirror then returns the reference's (1,1,1));
 *   n_eval_dev (nullable): i32, how many of the H hypotheses the staged loop scored.
 *   The inliers reported are those of the RETURNED pose (after the refit and its local-optimisation round).
 * Kcam: host pointer, 9 doubles row-major.
 */
/* Diagnostics: EVERY root of the device P3P solver for S independent 3-point problems (the production
 * kernels keep one per sample): X (S,3,3), uv (S,3,2) device f64 -> poses (S,4,12) [R|t], n_roots (S). */
int isr_p3p_all_roots(const double* X, const double* uv, const double* Kcam, int S, double* poses,
                      int32_t* n_roots, isr_stream_t stream);
size_t isr_pnp_ransac_workspace_bytes(int M_cap, int H);
int isr_p3p_hypotheses(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap,
                       const double* Kcam, int H, uint64_t seed, double* Rt, uint8_t* ok,
                       int32_t* sample, isr_stream_t stream);
int isr_ransac_score(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap,
                     const double* Kcam, const double* Rt, const uint8_t* ok, int H, float reperr,
                     int32_t* n_inl, int32_t* best_dev, uint32_t* best_mask, void* ws,
                     size_t ws_bytes, isr_stream_t stream);
int isr_pnp_refine(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap,
                   const uint32_t* mask, const double* Kcam, int iters, double* Rt_io, void* ws,
                   size_t ws_bytes, isr_stream_t stream);
int isr_pnp_ransac(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap,
                   const double* Kcam, int H, uint64_t seed, float reperr, double confidence,
                   int refine_iters, double* pose_dev, int32_t* inl_idx, int32_t* n_inl_dev,
                   int32_t* status_dev, int32_t* n_eval_dev, void* ws, size_t ws_bytes, isr_stream_t stream);

/* isr_pnp_ransac for a GROUP of B images as one chain of launches (image = blockIdx.z of every kernel):
 * p3d (B, M_cap, 3), p2d (B, M_cap, 2), M_dev (B); Kcams HOST (B, 9) f64, seeds HOST (B) u64;
 * pose_dev (B, 12), inl_idx (B, M_cap), n_inl_dev (B), status_dev (B).  Same kernels as the
 * single-image call: image b's outputs are bit-identical to isr_pnp_ransac on image b alone. */
size_t isr_pnp_ransac_batch_workspace_bytes(int M_cap, int H, int B);
int isr_pnp_ransac_batch(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap, int B,
                         const double* Kcams, int H, const uint64_t* seeds, float reperr,
                         double confidence, int refine_iters, double* pose_dev, int32_t* inl_idx,
                         int32_t* n_inl_dev, int32_t* status_dev, int32_t* n_eval_dev /* (B), nullable */,
                         void* ws, size_t ws_bytes, isr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K3 / K4  batched brute-force nearest neighbour with fused reductions
 * replaces  sklearn KDTree(...).query(k=1)                    inference.py:118-120 (ADD-S)
 *           open3d compute_point_cloud_distance               verfication.py:97,99; icp.py:113,115
 *           open3d evaluate_registration / registration_icp   icp.py:97-103 (one ICP iteration)
 * For batch item b: queries  q' = Tq[b] * qry,  targets t' = Tt[b] * tgt  (3x4 row-major f64
 * [R|t], NULL = identity; transformed in f64, rounded to f32 for the search).  For each query the
 * nearest target under f32 squared distance (lowest index on ties); its distance is then
 * re-evaluated in f64.  radius < 0: every query counts; else only queries with d <= radius.
 * Outputs (all nullable except n_in or sum_d):
 *   sum_d[b]  = sum of d over counted queries (f64)      n_in[b] = number counted
 *   sum_d2[b] = sum of d^2
 *   nn_idx[b*Nq+q] (-1 if not counted), nn_d[b*Nq+q] (f64 distance of the nearest target)
 *   cov[b*16 + ..] = { sum q'(3), sum t'(3), sum q' t'^T (9, row-major), count } over counted
 *                     pairs — the Kabsch inputs of one point-to-point ICP step.
 * Reductions use fixed-shape trees (no float atomics): results are run-to-run reproducible.
 */
size_t isr_nn_batched_workspace_bytes(int Nq, int Nt, int B);
int isr_nn_batched(const float* qry, int Nq, const float* tgt, int Nt, const double* Tq,
                   const double* Tt, int B, double radius, double* sum_d, double* sum_d2,
                   int32_t* n_in, int32_t* nn_idx, double* nn_d, double* cov, void* ws,
                   size_t ws_bytes, isr_stream_t stream);

/* a11 (the n x n vote, choosePose.py:121-145) needs only the decision ADDS(...) < 0.1 * diameter.  Bounds of
 *   sum_v dist(Tt[b]^-1 Tq[b] v, S)   (= V x ADD-S of item b: isr_nn_batched's sum_d with queries verts, targets S)
 * from a distance field of the surface cloud S: field (nz, ny, nx) f32 = the exact distance from each cell CENTRE of a uniform
 * grid (origin grid_min, edge h; the caller builds it once per cloud, e.g. through isr_nn_batched on the centres) — dist(., S)
 * is 1-Lipschitz, so a vertex x lies within |x - centre(c)| of field[c] for the cell c that holds it (or, outside the grid, the
 * nearest cell of the grid; its distance to surface_bbox (lo xyz, hi xyz) is then a second lower bound): both sums are
 * finite for every pose.  verts (V,3) f32, Tq / Tt (B,12) f64
 * [R|t] (Tt nullable = identity), field on the device.  Tt is inverted as a rigid motion (R^T, -R^T t); where it is only
 * nearly one (f32-born rotations, ||R^T R - I|| = eta ~ 1e-7) every vertex's bracket is widened by eta (max|s| + its upper
 * bound), so the sums stay rigorous bounds of isr_nn_batched's sum_d for the matrix AS GIVEN; grid_min (3 doubles) and surface_bbox (6 floats) on the HOST.
 * lb_sum, ub_sum (B) f64 on the device.  An item whose bounds straddle the threshold is evaluated exactly by the caller. */
int isr_adds_bounds(const float* verts, int V, const double* Tq, const double* Tt, int B, const float* field,
                    const double* grid_min, double h, int nx, int ny, int nz, const float* surface_bbox,
                    double* lb_sum, double* ub_sum, isr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a6 / a7  estimate_pose(): the SurfEmb-style sample-and-score estimator, poseEstSurf.py:11-261
 * (stages; the host driver pose_est_surf.estimate_pose keeps the reference's signature).
 *
 * isr_ep_prepare   :47-69   mask_lgts (r,r), query_img (r,r,e) f32 -> res = r / scale, n = res^2:
 *                  mask_log_prob / neg_mask_log_prob (n) = [3x3-max-pooled] max_pool(logsigmoid(+-lgts)),
 *                  mask_prob (n) = sigmoid(avg_pool(lgts)), queries (n,e) = avg_pool(query_img).
 *                  ws >= 2*n*4 + 512 bytes.
 * isr_ep_pool_corr :97-107  pooled[o][k] = max over the 3x3 pixel neighbourhood of corr_log[.][k].
 * isr_ep_sample    :111-119 corr_idx (n_samples,4) i64 = flat (pixel*m + key) indices drawn with
 *                  probability ~ (exp(corr_log) * mask_prob)^alpha by inversion of Philox uniforms
 *                  ((x+0.5)/2^32, counter (s,1,0,0)); no (n*m) cumulative array is formed.
 * isr_ep_p3p       :133-144 per sample: P3P (f64) on the first three correspondences, roots ordered by
 *                  the 4th point's reprojection error, one picked by Philox (counter (s,2,0,0));
 *                  samples that repeat a correspondence are rejected.  poses (S,12) f64, ok (S) u8.
 *                  Pixel coordinates are (idx % res, idx / res) as img_pts at :56-59.
 * isr_zbuf_score   :182-237 batch_score for B poses (Rt (B,12) f32): z-buffer by atomicMin on packed
 *                  (ordered z, vertex) u64 — lowest vertex on equal z — then
 *                  mask_score = mean_n(hit ? mask_log_prob : neg_mask_log_prob)/ln 2,
 *                  coord_score = mean_hit(corr_log[pixel, vertex])/ln m (-inf without hits).
 * Kcam: host, 9 doubles, ALREADY divided by the down-sample scale (poseEstSurf.py:42-45).
 */
int isr_ep_prepare(const float* mask_lgts, const float* query_img, int r, int e, int scale, int max_pool,
                   float* mask_log_prob, float* neg_mask_log_prob, float* mask_prob, float* queries,
                   void* ws, size_t ws_bytes, isr_stream_t stream);
int isr_ep_pool_corr(const float* corr_log, int res, int m, float* pooled, isr_stream_t stream);
/* isr_ep_corr_matrices :70, :97-107 (avg_queries = True) in ONE pass over the output: corr_raw (n, m) =
 *                  log_softmax(queries @ keys^T) (n = res^2 pooled query pixels, e <= 128) and, when corr_pool is
 *                  not NULL, corr_pool (n, m) = its 3 x 3 spatial max-pool — the logits of the three image rows a
 *                  workgroup needs are recomputed in registers instead of re-reading the matrix nine times.
 *                  ws from isr_corr_logsoftmax_workspace_bytes(n, m, e, ISR_DTYPE_F32). */
int isr_ep_corr_matrices(const float* queries, const float* keys, int res, int m, int e, float* corr_raw,
                         float* corr_pool, void* ws, size_t ws_bytes, isr_stream_t stream);
/* isr_ep_patch_corr :72-96 (avg_queries = False): per-PIXEL log_softmax(query_img[y, x] . obj_keys) pooled per
 *                  scale x scale block without ever forming the (r^2 x m) matrix: corr_centre (n, m) = the value at
 *                  the block's centre pixel (offset scale // 2: the sampling matrix before exp), corr_blockmax (n, m) =
 *                  the block maximum (the scoring matrix before the 3x3 pool).  query_img (r, r, e), obj_keys (m, e) f32. */
/* Round 3: the per-pixel log-sum-exps come from isr_corr_argmax's exact-f32 path (one row per pixel of the crop), then
 * one pass with thread = key and a workgroup per row of output cells writes both matrices (ws from
 * isr_ep_patch_corr_workspace_bytes; without ws, or when `scale` pixel rows of descriptors do not fit 64 KB of LDS, it falls
 * back to isr_ep_patch_corr_cells: one workgroup per cell, three sweeps over the keys, no scratch). */
size_t isr_ep_patch_corr_workspace_bytes(int r, int m, int e);
int isr_ep_patch_corr(const float* query_img, const float* obj_keys, int r, int e, int scale, int m,
                      float* corr_centre, float* corr_blockmax, void* ws, size_t ws_bytes, isr_stream_t stream);
int isr_ep_patch_corr_cells(const float* query_img, const float* obj_keys, int r, int e, int scale, int m,
                            float* corr_centre, float* corr_blockmax, isr_stream_t stream);
size_t isr_ep_sample_workspace_bytes(int n, int m);
int isr_ep_sample(const float* corr_log, const float* mask_prob, int n, int m, double alpha, int n_samples,
                  uint64_t seed, int64_t* corr_idx, void* ws, size_t ws_bytes, isr_stream_t stream);
/* Matrix-free forms of :111-119 and :182-237 (round 3): estimate_pose never needs the (n x m) matrices as arrays, only
 * (a) their weights summed in index order and (b) ~1e6 gathered elements, so both are formed on the spot from the
 * descriptors.  The element for grid pixel g and key k is  <qgrid[g], keys[k]> (k-ordered fmaf chain from 0) - lse_grid[g]
 * — bit for bit what isr_corr_logsoftmax / isr_ep_corr_matrices / isr_ep_patch_corr write — with lse_grid from
 * isr_corr_argmax's `lse` output on the same rows.  The descriptor grid: qgrid (rows of g_pitch pixels, e floats each),
 * lse_grid (same indexing); one output pixel of the res x res grid stands for win x win grid pixels:
 *   avg_queries=True   qgrid = isr_ep_prepare's queries, g_pitch = res, win = 1;
 *   avg_queries=False  qgrid = query_img (r, r, e), g_pitch = r, win = scale (sampling reads the block's centre pixel, offset
 *                      win / 2, as corr_centre; scoring takes the block maximum, as corr_blockmax).
 * isr_ep_sample_direct returns the indices isr_ep_sample returns on the materialised matrix (same weights, same order of
 * f64 additions: per row and 512-key chunk sequentially in key order, chunks in order, rows by a fixed-shape scan);
 * isr_zbuf_score_direct returns isr_zbuf_score's scores on the [3x3 max-pooled when `pool`] matrix.  ws as for the matrix forms.
 * isr_ep_sample_weights writes the n*m f64 weights exp(alpha*corr_log)*mask_prob^alpha the sampler adds (a validation aid). */
int isr_ep_sample_direct(const float* qgrid, const float* lse_grid, int g_pitch, int e, int win, int res,
                         const float* mask_prob, const float* keys, int m, double alpha, int n_samples, uint64_t seed,
                         int64_t* corr_idx, void* ws, size_t ws_bytes, isr_stream_t stream);
int isr_ep_sample_weights(const float* corr_log, const float* mask_prob, int n, int m, double alpha, double* weights,
                          void* ws, size_t ws_bytes, isr_stream_t stream);
int isr_ep_p3p(const int64_t* corr_idx, int res, int m, const float* obj_pts, const double* Kcam, int S,
               uint64_t seed, double* poses, uint8_t* ok, isr_stream_t stream);
/* isr_ep_prune      :147-177 per sample: dist_2d (S) f32 = largest pairwise pixel distance of the first three
 *                  correspondences, size_mask / normals_mask (S) u8, keep (S) u8 = ok and (all three masks, when
 *                  do_prune); then the ordered list keep_idx of the kept samples, their number *n_keep, and the f32
 *                  [R|t] rows Rt32 (max_eval, 12) of the first max_eval of them (the poses batch_score evaluates).
 *                  obj_normals (m, 3) f64 (normals_scaled.npy is float64), K00 = the down-scaled focal length. */
int isr_ep_prune(const int64_t* corr_idx, const double* poses, const uint8_t* ok, const float* obj_pts,
                 const double* obj_normals, int S, int res, int m, double K00, double obj_diameter,
                 double dist_2d_min, int do_prune, int max_eval, float* dist_2d, uint8_t* size_mask,
                 uint8_t* normals_mask, uint8_t* keep, int32_t* keep_idx, int32_t* n_keep, float* Rt32,
                 isr_stream_t stream);
size_t isr_zbuf_score_workspace_bytes(int B, int res);
int isr_zbuf_score(const float* obj_pts, int m, const float* Rt, int B, const double* Kcam, int res,
                   const float* mask_log_prob, const float* neg_mask_log_prob, const float* corr_log,
                   float* pose_score, float* mask_score, float* coord_score, void* ws, size_t ws_bytes,
                   isr_stream_t stream);
int isr_zbuf_score_direct(const float* obj_pts, int m, const float* Rt, int B, const double* Kcam, int res,
                          const float* mask_log_prob, const float* neg_mask_log_prob, const float* qgrid,
                          const float* lse_grid, int g_pitch, int e, int win, int pool, const float* keys,
                          float* pose_score, float* mask_score, float* coord_score, void* ws, size_t ws_bytes,
                          isr_stream_t stream);

/* The whole of estimate_pose(poses=None) as ONE call (matrix-free route): the launches of isr_ep_prepare, isr_corr_argmax (row
 * log-sum-exps), isr_ep_sample_direct, isr_ep_p3p, isr_ep_prune and isr_zbuf_score_direct, intermediates carved from ws.
 * Inputs on the device except Kcam (host, 9 doubles, the UNSCALED crop camera: the (K + .5)/scale - .5 of :42-45 is applied
 * here); obj_normals (m,3) f64.  Outputs on the device: Rt32 (max_pose_evaluations, 12) f32 [R|t] of the scored poses,
 * pose / mask / coord scores (max_pose_evaluations), and per SAMPLE (max_poses each) dist_2d f32, size_mask, normals_mask,
 * solved u8 (the reference's returned arrays are the entries with solved = 1, in order).  *n_poses_host = how many poses were
 * scored (rows of Rt32 / scores that are valid), *n_keep_host (nullable) = how many samples survived the pruning.  The stream
 * is synchronised once INSIDE the call (hipStreamSynchronize: the survivor count sizes the scoring launch), so the entry cannot
 * be used under stream capture; max_pose_evaluations <= 65 535 (one scorer launch, pose = blockIdx.y — larger values: the stage
 * entry points, scoring in slices, as pose_est_surf.estimate_pose does).  Same bits as the stage entry points in sequence. */
size_t isr_estimate_pose_workspace_bytes(int r, int e, int m, int scale, int max_poses, int max_pose_evaluations,
                                         int avg_queries);
int isr_estimate_pose(const float* mask_lgts, const float* query_img, int r, int e, const float* obj_pts,
                      const double* obj_normals, const float* obj_keys, int m, double obj_diameter, const double* Kcam,
                      int max_poses, int max_pose_evaluations, int down_sample_scale, double alpha, double dist_2d_min,
                      int max_pool, int avg_queries, int do_prune, uint64_t seed, float* Rt32, float* pose_scores,
                      float* mask_scores, float* coord_scores, float* dist_2d, uint8_t* size_mask, uint8_t* normals_mask,
                      uint8_t* solved, int32_t* n_poses_host, int32_t* n_keep_host, void* ws, size_t ws_bytes,
                      isr_stream_t stream);

/* a16  refine_pose objective   pose_refine.py:58-91
 * out4 (device, 4 f64) = { score, d score / d t (3) } with
 *   score = -( mean_i <keys_i, bilinear(query_img, p_i)> - mean_i bilinear(denom_img, p_i) ) / 2,
 *   p_i = pixel of X_i under Kcrop [R|t]; bilinear sampling as F.grid_sample(align_corners=False,
 *   padding_mode='border') on (p + 0.5) * 2 / res - 1.  X (N,3) object coordinates, keys (N,e),
 *   query_img (res,res,e), denom_img (res,res) all f32; Kcrop (9) and Rt (12, [R|t]) HOST doubles.
 *   The reference's rotation is constant inside the objective (pose_refine.py:73-76), so only the
 *   translation gradient exists.  ws >= 64*14*8 + 256 bytes. */
#define ISR_INTERP_BILINEAR 0
#define ISR_INTERP_NEAREST 1 /* piecewise constant: the gradient is 0, as torch autograd reports it */
#define ISR_INTERP_BICUBIC 2 /* cubic convolution A = -0.75, taps clipped to the border, source coordinate not clipped */
/* `interpolation`: the `mode=` the reference forwards to F.grid_sample (pose_refine.py:60-68), ISR_INTERP_*. */
int isr_refine_objective(const float* X, const float* keys, int N, int e, const float* query_img,
                         const float* denom_img, int res, int interpolation, const double* Kcrop, const double* Rt,
                         double* out4, void* ws, size_t ws_bytes, isr_stream_t stream);

/* The same objective with the rotation live (the evidently intended variant): out13 (device, 13 f64)
 * = { score, d score / d t (3), d score / d R (9, row-major) }; the caller chains d/dR with the Rodrigues Jacobian.
 * ws >= 64*14*8 + 256 bytes (also enough for isr_refine_objective). */
int isr_refine_objective_full(const float* X, const float* keys, int N, int e, const float* query_img,
                              const float* denom_img, int res, int interpolation, const double* Kcrop, const double* Rt,
                              double* out13, void* ws, size_t ws_bytes, isr_stream_t stream);

/* a8  ADD(verts, gtR, gtT, R, T)   inference.py:116-117
 * mean_out[b] = mean_v || Ta[b] v - Tb[b] v ||  (f64; Ta/Tb (B,12) f64 [R|t], NULL = identity). */
int isr_add_metric(const float* verts, int V, const double* Ta, const double* Tb, int B,
                   double* mean_out, isr_stream_t stream);

/* a14  registration_icp(source, target, threshold, init, PointToPoint)   icp.py:101-103
 * The whole loop on the device, no host round trip: up to max_iter + 1 evaluation passes (K3 with
 * radius + Kabsch sums), after each the stopping rule |d fitness| < rel_fitness and |d rmse| <
 * rel_rmse (Open3D defaults 1e-6, 30 iterations) and the rigid update T <- dT T (Horn's closed form).
 * T_io: device, 16 f64 row-major 4x4 (init in, result out; the bottom row is rewritten as 0 0 0 1).
 * result: device, 4 f64 = { fitness, inlier_rmse, iterations done, correspondences }.
 * The result does not depend on the order of the rows of src / tgt beyond the last bits of the f64 sums.  The searches
 * skip, wave by wave, the 256-row target tiles that lie beyond the wave's bound and beyond the radius: rows that are
 * neighbours in space (e.g. Morton order; a rigid motion keeps it) make that effective — 12.3 -> 8.7 ms per loop at
 * 50 000 points — and cost nothing otherwise. */
size_t isr_icp_workspace_bytes(int Ns, int Nt);
int isr_icp_point_to_point(const float* src, int Ns, const float* tgt, int Nt, double threshold,
                           int max_iter, double rel_fitness, double rel_rmse, double* T_io,
                           double* result, void* ws, size_t ws_bytes, isr_stream_t stream);

/* a10 / a12  relative-pose tables, rows [i0, i1) of the n x n table, written as (i1-i0, n, 12) f64.
 * mode 0: compute_rel_poses       choosePose.py:43-51  ->  [R_i^T R_j | t_j - t_i]
 * mode 1: calculate_relative_pose verfication.py:9-19  ->  [R_j|t_j] * inv([R_i|t_i])
 * R (n,9), t (n,3) f64 device. */
int isr_rel_pose_table(const double* R, const double* t, int n, int i0, int i1, int mode,
                       double* out, isr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ISR_HIP_H */
