// corr_argmax.hip — K1: fused key/query correlation + online log-sum-exp + arg-max for gfx950.
//
// Replaces  getCors(queries, feats, leaves=1):
//             cMat = torch.log_softmax(queries @ feats.T, -1); vals, idx = torch.topk(cMat, 1)
//           inference.py:142-149 (= finalposes.py:38-45 = choosePose.py:35-42)
//           and the logsumexp denominator of pose_refine.py:56 (the `lse` output).
// The (P x N) matrix is never written: each wave keeps 64 queries' descriptors in registers as
// MFMA B operands, the block streams the keys through an XOR-swizzled LDS tile as A operands,
// and each 32(keys) x 32(queries) accumulator tile is consumed in registers.
//
// Orientation: S^T = K Q^T, so the C/D layout puts the QUERY on the lane (col = lane & 31) and 16
// KEYS in the lane's registers (row = (r&3) + 8(r>>2) + 4(lane>>5)): the reduction over keys is
// lane-local — no cross-lane traffic until one 2-lane merge per chunk.
//
// A query's result is a function of (query, keys) ONLY — not of the launch it travels in:
//   * the log-sum-exp is accumulated over CANONICAL CHUNKS of kChunk = 4096 keys (a constant): per chunk a
//     partial (reference R_c, l_c = sum 2^(s log2e - R_c)) whose arithmetic order is fixed, written
//     to the workspace; corr_finalize_kernel adds the chunks in ascending order in f64.  Which
//     workgroup computed a chunk (the plan splits the key range by occupancy) never matters.
//   * bf16 fast path (corr_direct.hpp): R_c = 0 for everyone; queries outside its range are redone
//     per query by corr_bf16_kernel with R_c = ceil(chunk maximum).
//   * arg-max: raw MFMA logits (C = 0) compare equal wherever they are computed; a query whose top-2
//     margin is inside the f32 accumulation error bound is decided in exact arithmetic
//     (corr_recheck_kernel: exact bf16 products summed in f64 in the order of
//     oracle/isr_oracle.c:orc_corr_argmax_bf16), so bf16 indices equal the oracle's, ties to the
//     lowest key.
// dtype f32: v_mfma_f32_32x32x2_f32, bit for bit a k-ordered fmaf chain — the oracle's
// orc_corr_argmax_f32 reproduces its logits exactly (no recheck needed).
// Cost per 32x32 tile at D = 64: 4 MFMA (128 matrix-pipe cycles) against ~320 cycles of VALU issue
// (16 v_exp_f32 at 8.3, 16 adds at 2.8, 8 max at 4.6, ..., the MFMAs' own ~18 each): the loop is bound
// by VALU issue, almost half of it the transcendental unit — DESIGN.md section 4 has the measurements.
#include "isr_common.hpp"

#include <type_traits>

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;     // eight 16-bit operand elements (bf16, or f16 on the F16 routes)
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kQB = 2;                       // 32-query blocks per wave
constexpr int kTK = 128;                     // keys per LDS stage
constexpr int kChunkStages = 32;             // stages per canonical chunk
constexpr int kChunk = kTK * kChunkStages;   // keys per canonical chunk of the log-sum-exp (a CONSTANT)
constexpr float kLog2e = 1.4426950408889634f;
// M2 of a lane that has seen no valid key yet: finite, so a fully masked tile gives
// exp2(fma(-inf, log2e, 1e30)) = 0 instead of NaN, and the first real key rescales l by 2^-huge = 0.
constexpr float kNoM2 = -1.0e30f;
constexpr int kKnBlocks = 64;                // blocks of the key-norm pass
constexpr int kRSplitMax = 4;                // key ranges of the recheck pass (latency, not arithmetic; sizes rval / ridx)

// Workspace of one isr_corr_argmax call.
struct CorrWs {
  float* pm;        // (nsplit, P) maximum logit of the key range (raw units of the kernel)
  float* pm2;       // (nsplit, P) runner-up inside the range (bf16 paths)
  int32_t* pbi;     // (nsplit, P) key of the maximum
  int32_t* pbad;    // (nsplit, P) 1: outside the direct kernel's range, the fallback owns this (range, query)
  float* plc;       // (nchunks, P) chunk sums l_c
  float* pmc;       // (nchunks, P) chunk maxima (fallback / f32: the chunk reference is ceil of it)
  float* qn2;       // (P) |q|^2, then (finalize) the recheck threshold
  int32_t* flags;   // (nsplit, qblocks) workgroup holds a bad query
  int32_t* blist;   // the set flags as a list of (range * qblocks + query block); its length is rcount[1]
  float* kn2;       // (kKnBlocks) per-block max |k|^2
  int32_t* rcount;  // recheck list length
  long long* clk;   // diagnostics: {shader-clock ticks, 100 MHz reference ticks} over the life of workgroup (0, 0)
  int32_t* rlist;   // (P) queries to decide exactly
  double* rval;     // (rsplit, P) exact best value per recheck key range and list entry
  int32_t* ridx;    // (rsplit, P)
  // launch gates (nullable): the f16-plane route of an f32 call runs unless a descriptor was too large for f16 (`skip` set by
  // its split kernel), in which case the f32-MFMA chain kernels behind it run instead (`only`): no host round trip decides.
  const int32_t* skip;   // leave at once when *skip != 0
  const int32_t* only;   // leave at once when *only == 0
  // tile skip (SKIP instantiations of the direct kernel): per query a LOWER bound of its maximum logit (raw units), or null
  const float* lower;
  int lower_stride;      // floats per query in `lower` (the bound first)
  float skip_T;          // log2 units below the bound at which a piece stops counting
  float* lowbuf;         // (P) where the call's own pre-pass leaves those bounds
  // the screened route (corr_sparse.hpp): block-scaled FP6 images of the rows, the queries' {|q|, |dq|}, max |dk|^2 and max |k~|^2
  uint8_t* q6;           // (P, 64 B)
  uint8_t* k6;           // (N, 64 B)
  float* qnrm;           // (P, 2)
  uint32_t* kmax;        // (2) float bit patterns
  unsigned long long* redone;   // diagnostics: [0] tile items redone exactly, [1] query blocks handed to the dense kernel
  int32_t* hand;         // (query blocks of 256) 1: pass 1 left the block to the dense tile-skip kernel
  int nhand;             // its length
  float skip_default;    // the threshold when `lower` is null
  // isr_corr_argmax_digits: per image (rows [b rows_per_image, (b + 1) rows_per_image) of the queries) the histogram of the
  // leading digit of the log-probabilities of its first n_rows[b] rows, counted where logp is written (corr_finish)
  int32_t* digits;       // (images, isr::kDigitBins) or null
  const int32_t* n_rows; // (images) or null: every row counts
  int rows_per_image;
  int ndigits;           // images * isr::kDigitBins
};

__device__ __forceinline__ bool gated_off(const CorrWs& ws) {
  return (ws.skip && *ws.skip != 0) || (ws.only && *ws.only == 0);
}

struct LaneState {
  float m;    // running max logit
  float M2;   // ceil(m * log2e)
  float l;    // sum 2^(s*log2e - M2)
  int bi;     // key index of m
};

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

__device__ __forceinline__ f32x16 splat16(float v) {
  return f32x16{v, v, v, v, v, v, v, v, v, v, v, v, v, v, v, v};
}

// Part A of consuming a tile (rows kb + 4h + (r&3) + 8(r>>2), r = 0..15, of this lane's query):
// the running maximum / arg-max and the integer log2 reference M2.
__device__ __forceinline__ void update_max(const f32x16& acc, int krow0, LaneState& st) {
  const float x0 = max3(acc[0], acc[1], acc[2]), x1 = max3(acc[3], acc[4], acc[5]),
              x2 = max3(acc[6], acc[7], acc[8]), x3 = max3(acc[9], acc[10], acc[11]),
              x4 = max3(acc[12], acc[13], acc[14]);
  const float t = fmaxf(max3(x0, x1, x2), max3(x3, x4, acc[15]));
  if (__any(t > st.m)) {  // wave-uniform; taken for ~64(1 + ln(tiles/64)) of a key range's tiles
    if (t > st.m) {
      int r = 15;
#pragma unroll
      for (int i = 14; i >= 0; --i) r = (acc[i] == t) ? i : r;  // lowest register = lowest key
      st.bi = krow0 + (r & 3) + 8 * (r >> 2);
      const float M2n = ceilf(t * kLog2e);
      st.l *= __builtin_amdgcn_exp2f(st.M2 - M2n);  // exact power of two (0 when M2 = kNoM2)
      st.M2 = M2n;
      st.m = t;
    }
  }
}

// Part B: l += sum_i 2^(acc_i log2e - M2).
__device__ __forceinline__ void accumulate_exp(const f32x16& acc, LaneState& st) {
  const float nM2 = -st.M2;
#pragma unroll
  for (int i = 0; i < 16; ++i) st.l += __builtin_amdgcn_exp2f(__builtin_fmaf(acc[i], kLog2e, nM2));
}

__device__ __forceinline__ void consume_tile(const f32x16& acc, int krow0, LaneState& st) {
  update_max(acc, krow0, st);
  accumulate_exp(acc, st);
}

__device__ __forceinline__ void mask_tail(f32x16& acc, int krow0, int N) {
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (krow0 + (i & 3) + 8 * (i >> 2) >= N) acc[i] = -__builtin_inff();
}

__device__ __forceinline__ bool better(float ma, int ia, float mb, int ib) {
  return (ma > mb) || (ma == mb && ia < ib);
}

// the two lanes (h = 0, 1) of a query -> one (m, M2, l, bi); M2 = ceil(m log2e) stays true
__device__ __forceinline__ void merge_state(LaneState& a, const LaneState& b) {
  const float M2 = fmaxf(a.M2, b.M2);
  const float la = (a.M2 == M2) ? a.l : a.l * __builtin_amdgcn_exp2f(a.M2 - M2);
  const float lb = (b.M2 == M2) ? b.l : b.l * __builtin_amdgcn_exp2f(b.M2 - M2);
  if (better(b.m, b.bi, a.m, a.bi)) { a.m = b.m; a.bi = b.bi; }
  a.M2 = M2;
  a.l = la + lb;
}

__device__ __forceinline__ LaneState merged_with_other_half(LaneState st) {
  LaneState o;
  o.m = __shfl_xor(st.m, 32, 64);
  o.M2 = __shfl_xor(st.M2, 32, 64);
  o.l = __shfl_xor(st.l, 32, 64);
  o.bi = __shfl_xor(st.bi, 32, 64);
  merge_state(st, o);
  return st;
}

// ------------------------------------------------------------------------- operand rows, MFMA chain
// Plain layout (SP = 0): a row is DK 16-wide bf16 blocks, one fragment per block, one MFMA per block and tile.
// Split layout (SP > 0, the exact-f32 route for D <= 16 SP): a row is the three bf16 PLANES x1 | x2 | x3 of an f32 row
// (x = x1 + x2 + x3 exactly, 8 + 8 + 8 mantissa bits), SP blocks each; fragment p * SP + j = block j of plane p + 1.  A tile
// is the six plane pairs down to 2^-16 — q1k3, q2k2, q1k2, q3k1, q2k1, q1k1 (q2k3, q3k2, q3k3 are 2^-24 relative and
// dropped) — issued smallest first and so that every key plane's uses are contiguous (k3 is free after SP MFMAs, k2 after
// 3 SP: their ds_reads for the next key sub-tile start early): 6 SP MFMAs from 3 SP + 3 SP fragments, where the round-3
// form of the route multiplied the 96-wide rows [x1 x1 x2 x2 x1 x3] . [k1 k2 k1 k2 k3 k1] with 6 + 6 fragments at D <= 16.
// F16 (with SP > 0): the planes are f16 — x1 = f16(x), x2s = f16((x - x1) 2^11), x1s = f16(x1 2^-11) — and a tile is the
// THREE pairs k1s q2s, k2s q1s, k1 q1 (x = x1 + x2 to 2^-22: 11 + 11 mantissa bits; the dropped k2 q2 is 2^-22 relative):
// half the matrix instructions of the bf16 planes, which is what bounds that kernel (MFMA busy 0.87 at D = 64).  The 2^11 /
// 2^-11 pair keeps the residual plane out of the subnormals; where x1s does become subnormal (|x1| < 2^-3) its rounding
// error is at most 2^-25 ABSOLUTE — v_mfma_f32_32x32x16_f16 keeps subnormal inputs on gfx950 (tools/mfma_f16_denorm.hip) —
// which the margin test carries as an extra absolute term (split_eabs).  |x| must stay below 65 504: the split kernel raises
// a flag otherwise and the call falls through to the f32-MFMA chain kernels (CorrWs::skip / only).
template <int DK, int SP, bool F16 = false>
struct RowFrags {
  static constexpr int NFR = SP ? 3 * SP : DK;                 // fragments per operand row
  static constexpr int NMF = SP ? (F16 ? 3 : 6) * SP : DK;     // matrix instructions per 32 x 32 tile
  static constexpr int NQN = SP ? SP : DK;                     // fragments that enter the norm of the margin test (plane 1)
  // split rows, phase by phase: key plane, query plane, the key plane whose last reader the phase is (-1: none), and the
  // slice [E0[ph], E0[ph + 1]) of the previous tile's 16 exp + add the phase carries (the maxima ride in phase 0)
  static constexpr int NPH = F16 ? 3 : 6;
  static constexpr int PA[6] = {2, 1, F16 ? 0 : 1, 0, 0, 0};
  static constexpr int PB[6] = {F16 ? 1 : 0, F16 ? 2 : 1, 0, 2, 1, 0};
  static constexpr int RD[6] = {2, F16 ? 1 : -1, F16 ? 0 : 1, -1, -1, 0};
  static constexpr int E0[7] = {0, F16 ? 4 : 0, F16 ? 10 : 4, F16 ? 16 : 8, 12, 16, 16};
};

template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// TIMING-ONLY ablation ISR_ABL_MFMA16 (profiles/r05_k1_mfma_shape_ab.txt): two 16-wide blocks (a0, b0), (a1, b1) of a 32 x 32
// tile through FOUR v_mfma_f32_16x16x32 sub-tile instructions instead of two v_mfma_f32_32x32x16 — the same matrix cycles from
// the same fragments; the numbers are not the tile's logits (the fragments keep the 32x32x16 layout)
template <bool F16>
__device__ __forceinline__ f32x16 mfma16_pair_as_16x16x32(const bf16x8& a0, const bf16x8& a1, const bf16x8& b0, const bf16x8& b1, f32x16 c) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f32x4 t = f32x4{c[4 * u], c[4 * u + 1], c[4 * u + 2], c[4 * u + 3]};
    const bf16x8& a = (u & 1) ? a1 : a0;
    const bf16x8& b = (u >> 1) ? b1 : b0;
    if constexpr (F16) t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), t, 0, 0, 0);
    else t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, t, 0, 0, 0);
    c[4 * u] = t[0]; c[4 * u + 1] = t[1]; c[4 * u + 2] = t[2]; c[4 * u + 3] = t[3];
  }
  return c;
}

template <int DK, int SP, bool F16 = false>
__device__ __forceinline__ f32x16 tile_chain(const bf16x8 (&a)[RowFrags<DK, SP>::NFR], const bf16x8 (&b)[RowFrags<DK, SP>::NFR],
                                             f32x16 c) {
  using RF = RowFrags<DK, SP, F16>;
  if constexpr (SP == 0) {
#if defined(ISR_ABL_MFMA16)
    // TIMING-ONLY ablation (profiles/r05_k1_mfma_shape_ab.txt): the item's operand registers through v_mfma_f32_16x16x32_bf16 —
    // per 32 x 32 item four 16 x 16 sub-tiles x DK / 2 k-steps of 32, i.e. 2 DK instructions of half the cycles from the same
    // 2 DK fragments, 16 results per lane.  The fragments are the 32x32x16 layout's, so the numbers are NOT the tile's logits
    // (random dot products of the same rows): instruction mix, register use, LDS traffic and operand data are the real loop's.
    if constexpr (DK % 2 == 0) {
      using f32x4 = __attribute__((ext_vector_type(4))) float;
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = f32x4{c[4 * u], c[4 * u + 1], c[4 * u + 2], c[4 * u + 3]};
#pragma unroll
      for (int ks = 0; ks < DK / 2; ++ks)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          t[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u & 1) * (DK / 2) + ks], b[(u >> 1) * (DK / 2) + ks], t[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) { c[4 * u] = t[u][0]; c[4 * u + 1] = t[u][1]; c[4 * u + 2] = t[u][2]; c[4 * u + 3] = t[u][3]; }
    } else
#endif
#pragma unroll
    for (int s = 0; s < DK; ++s) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], c, 0, 0, 0);
  } else {
#pragma unroll
    for (int ph = 0; ph < RF::NPH; ++ph)
#pragma unroll
      for (int j = 0; j < SP; ++j) c = mfma16<F16>(a[RF::PA[ph] * SP + j], b[RF::PB[ph] * SP + j], c);
  }
  return c;
}

// one 16-bit operand element as f32
template <bool F16>
__device__ __forceinline__ float elem_f32(uint16_t v) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __uint_as_float((uint32_t)v << 16);
}

// What the margin test's (D + 2) 2^-23 |q||k| must cover on the split route, with |q|, |k| the norms of plane 1 (within
// 2^-8 of the f32 rows' norms): the f32 accumulation of n = 96 SP exact products whose absolute values sum to at most
// 1.004 sum_d |q_d k_d| (Higham, u = 2^-23), the dropped pairs (<= 2^-25 relative), the residuals of the two three-way
// splits and the rounding of q log2 e (<= 5 x 2^-24), and the rounding of the f32 chain the recheck decides by
// (<= 16 SP x 2^-24): D_eff + 2 >= 1.03 (96 SP + 2) + 0.51 (16 SP + 5).
constexpr int split_deff(int SP) { return (103 * (96 * SP + 2) + 51 * (16 * SP + 5)) / 100 + 1; }
// The same for the f16 planes: n = 48 SP exact products, their absolute values summing to <= 1.001 sum_d |q_d k_d|; the dropped
// k2 q2 and the roundings of the two residual planes (3 x 2^-22 = 12 x 2^-24); q log2 e (1); the f32 chain (16 SP):
// D_eff + 2 >= 1.002 (48 SP + 2) + 0.501 (13 + 16 SP).  And absolutely, for subnormal x1s elements, 2^-25 (|k|_1 + |q|_1)
// <= split_eabs(SP) (|k| + |q|) with split_eabs = 1.01 x 2^-25 sqrt(16 SP).
constexpr int split_deff_f16(int SP) { return (1002 * (48 * SP + 2) + 501 * (13 + 16 * SP)) / 1000 + 1; }
constexpr float split_eabs(int SP) { return (SP == 1 ? 4.04f : SP == 2 ? 5.72f : SP == 4 ? 8.08f : 11.43f) * 2.98023223876953125e-08f; }

// ------------------------------------------------------------------------- key staging (LDS)
// Key stages through a swizzled LDS image: coalesced 16-byte global loads, conflict-free ds_read_b128 in MFMA
// A-operand layout.  Shared by the fallback and recheck kernels (the direct kernel stages through a raw buffer
// descriptor, corr_direct.hpp).  Rows of a power-of-two number of 16-byte chunks: XOR swizzle over the row.  Split rows
// (6 SP chunks: a row starts 6 SP row (mod 16) sixteen-byte banks in, a pattern of period PD = 8 / SP rows): XOR inside each
// plane's 2 SP chunks with (row / PD) % (2 SP) — the 16 rows of a ds_read_b128 phase then hit 16 distinct banks.
template <int NCH, int SP>
__device__ __forceinline__ constexpr int key_slot(int row, int c) {
  if constexpr (SP == 0) {
    constexpr int RPB = (NCH >= 16) ? 1 : 16 / NCH;              // key rows per 256-byte LDS bank row
    return c ^ ((row / RPB) & (NCH - 1));
  } else {
    constexpr int W = 2 * SP, PD = (SP >= 8) ? 1 : 8 / SP;
    return (c / W) * W + ((c % W) ^ ((row / PD) & (W - 1)));
  }
}

template <int DK, int SP = 0>
struct KeyStage {
  static constexpr int NFR = RowFrags<DK, SP>::NFR;
  static constexpr int NCH = 2 * NFR;                        // 16-byte chunks per key row
  static constexpr int TK = (NCH <= 16) ? kTK : (NCH <= 32) ? 64 : 32;   // keys per LDS stage: two buffers within 64 KB
  static constexpr int CHUNKS = TK * NCH;                    // chunks per stage
  static constexpr int NLD = (CHUNKS + kThreads - 1) / kThreads;
  static_assert(SP != 0 || (NCH & (NCH - 1)) == 0, "plain rows: a power of two of 16-byte chunks");
  uint4 stg[NLD];
  __device__ __forceinline__ void gload(const uint16_t* __restrict__ K, int ldk, int key0, int k1) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int ci = threadIdx.x + i * kThreads;
      const int row = ci / NCH, c = ci % NCH;
      const int key = key0 + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ci < CHUNKS && key < k1) v = *reinterpret_cast<const uint4*>(K + (size_t)key * ldk + 8 * c);
      stg[i] = v;
    }
  }
  __device__ __forceinline__ void lwrite(uint4* lds) const {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int ci = threadIdx.x + i * kThreads;
      const int row = ci / NCH, c = ci % NCH;
      if (ci < CHUNKS) lds[row * NCH + key_slot<NCH, SP>(row, c)] = stg[i];
    }
  }
  static __device__ __forceinline__ void load_a(const uint4* lds, int sub, int r, int h, bf16x8 (&a)[NFR]) {
    const int row = sub * 32 + r;
#pragma unroll
    for (int s = 0; s < NFR; ++s) {
      const uint4 v = lds[row * NCH + key_slot<NCH, SP>(row, 2 * s + h)];
      a[s] = *reinterpret_cast<const bf16x8*>(&v);
    }
  }
};

// ---------------------------------------------------------------------- bf16, log2-domain fallback
// dtype ISR_DTYPE_BF16_LOG2, per-query reference: the subtraction of the integer reference M2 rides in
// the MFMA's C operand: each query block keeps a 16-register tile holding -M2 (rewritten only when
// M2 moves), the first MFMA of every chain takes it as C, so the accumulator already holds s' - M2
// and the epilogue is exp2 + add per element.  MFMA numerics with the large C term:
// profiles/r01_mfma_numerics.txt (max abs error 1.9e-5 over |s' - M2| <= 190, no bias).
struct L2State {
  float mr;   // max(s') - M2 of this lane's rows (-inf before the first key)
  float M2;   // integer reference, SHARED by the two lanes (h = 0, 1) of a query
  float l;    // sum 2^(s' - M2)
};

// Part A, log2 domain.  acc holds s' - M2; a new maximum above the reference bumps M2 by an integer d
// (both lanes of the query, exchanged with one cross-half shuffle), rescales l by 2^-d exactly,
// shifts this tile's accumulators and rewrites the query block's -M2 tile.
__device__ __forceinline__ void update_max_l2(f32x16& acc, L2State& st, f32x16& cinit) {
  const float x0 = max3(acc[0], acc[1], acc[2]), x1 = max3(acc[3], acc[4], acc[5]),
              x2 = max3(acc[6], acc[7], acc[8]), x3 = max3(acc[9], acc[10], acc[11]),
              x4 = max3(acc[12], acc[13], acc[14]);
  const float t = fmaxf(max3(x0, x1, x2), max3(x3, x4, acc[15]));
  if (__any(t > st.mr)) {
    const float ninf = -__builtin_inff();
    const bool first = !(st.mr > ninf);
    const bool up = t > st.mr;
    float d = up ? (first ? ceilf(t) : fmaxf(0.f, ceilf(t))) : (first ? ninf : 0.f);
    d = fmaxf(d, __shfl_xor(d, 32, 64));
    d = (d > ninf) ? d : 0.f;
    if (up) st.mr = t;
    if (__any(d != 0.f)) {
      if (d != 0.f) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] -= d;
        st.mr -= d;
        st.l = (st.l > 0.f) ? st.l * __builtin_amdgcn_exp2f(-d) : 0.f;
        st.M2 += d;
        cinit = splat16(-st.M2);
      }
    }
  }
}

// The per-query-reference kernel: ONE canonical chunk per work item, so every chunk starts from a
// clean state (C = 0 for the first tile) whatever else the launch holds.  It runs behind
// corr_bf16_direct_kernel and only for the query blocks that kernel LISTED as holding a bad query
// (ws.blist; a fixed grid strides over list x chunks — with nothing listed the launch costs a few
// microseconds, where a (query blocks x chunks) grid of early exits cost 130 us per 32-image launch);
// it leaves (chunk maximum, l_c relative to R_c = ceil(chunk maximum [* log2 e])) — the index of a
// bad query is always decided by the exact recheck, so no arg-max is kept here.
constexpr int kFallbackGrid = 1024;
template <int DK, bool LOG2, int SP = 0, bool F16 = false>
__global__ __launch_bounds__(kThreads, 1) void corr_bf16_kernel(
    const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K, int P, int N, int ldq, int ldk,
    int range_chunks, int qblocks, int nchunks, CorrWs ws) {
  using KS = KeyStage<DK, SP>;
  constexpr int NFR = KS::NFR, TKS = KS::TK;
  __shared__ uint4 lds[2][KS::CHUNKS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  if (gated_off(ws)) return;
  const int nlisted = ws.rcount[1];
  for (int item = blockIdx.x; item < nlisted * range_chunks; item += gridDim.x) {   // block-uniform
  const int ent = ws.blist[item / range_chunks];
  const int range = ent / qblocks, bx = ent - range * qblocks;
  const int chunk = range * range_chunks + item % range_chunks;
  if (chunk >= nchunks) continue;                   // the last key range may hold fewer chunks
  const int q0 = (bx * kWaves + wave) * (kQB * 32);

  bf16x8 bq[kQB][NFR];
#pragma unroll
  for (int qb = 0; qb < kQB; ++qb) {
    int row = q0 + qb * 32 + r;
    row = row < P ? row : P - 1;
    const uint16_t* src = Q + (size_t)row * ldq + 8 * h;
#pragma unroll
    for (int s = 0; s < NFR; ++s) bq[qb][s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  }
  LaneState st[kQB];
  L2State s2[kQB];
  f32x16 cinit[kQB];                    // LOG2: -M2 of the lane's query, the C operand of each chain
#pragma unroll
  for (int qb = 0; qb < kQB; ++qb) {
    st[qb].m = -__builtin_inff(); st[qb].M2 = kNoM2; st[qb].l = 0.f; st[qb].bi = 0;
    s2[qb].mr = -__builtin_inff(); s2[qb].M2 = 0.f; s2[qb].l = 0.f;
    cinit[qb] = splat16(0.f);
  }
  const int k0 = chunk * kChunk;
  const int k1 = min(N, k0 + kChunk);
  const int nstage = (k1 - k0 + TKS - 1) / TKS;
  KS ks;
  ks.gload(K, ldk, k0, k1);
  ks.lwrite(lds[0]);
  __syncthreads();
  for (int stage = 0; stage < nstage; ++stage) {
    const int buf = stage & 1;
    if (stage + 1 < nstage) ks.gload(K, ldk, k0 + (stage + 1) * TKS, k1);
#pragma unroll
    for (int sub = 0; sub < TKS / 32; ++sub) {
      const int kb = k0 + stage * TKS + sub * 32;
      if (kb < k1) {  // block-uniform
        bf16x8 a[NFR];
        KS::load_a(lds[buf], sub, r, h, a);
#pragma unroll
        for (int qb = 0; qb < kQB; ++qb) {
          f32x16 acc = tile_chain<DK, SP, F16>(a, bq[qb], LOG2 ? cinit[qb] : splat16(0.f));
          if (kb + 32 > k1) mask_tail(acc, kb + 4 * h, k1);
          if (LOG2) {
            update_max_l2(acc, s2[qb], cinit[qb]);
            float l = s2[qb].l;
#pragma unroll
            for (int i = 0; i < 16; ++i) l += __builtin_amdgcn_exp2f(acc[i]);
            s2[qb].l = l;
          } else {
            consume_tile(acc, kb + 4 * h, st[qb]);
          }
        }
      }
    }
    if (stage + 1 < nstage) ks.lwrite(lds[buf ^ 1]);
    __syncthreads();
  }
#pragma unroll
  for (int qb = 0; qb < kQB; ++qb) {
    const int q = q0 + qb * 32 + r;
    float mc, lc;
    if (LOG2) {
      // the halves share M2: chunk maximum mc = M2 + max(mr); stored reference Rs = ceil(mc) (equal to
      // M2 except when the f32 sum M2 + mr rounds down to M2 - 1: l is then rescaled by an exact 2)
      const float mr = fmaxf(s2[qb].mr, __shfl_xor(s2[qb].mr, 32, 64));
      const float l = s2[qb].l + __shfl_xor(s2[qb].l, 32, 64);
      mc = s2[qb].M2 + mr;
      lc = l * __builtin_amdgcn_exp2f(s2[qb].M2 - ceilf(mc));
    } else {
      const LaneState mgd = merged_with_other_half(st[qb]);      // M2 = ceil(m * log2 e)
      mc = mgd.m;
      lc = mgd.l;
    }
    if (h == 0 && q < P && ws.pbad[(size_t)range * P + q]) {
      ws.pmc[(size_t)chunk * P + q] = mc;
      ws.plc[(size_t)chunk * P + q] = lc;
    }
  }
  }  // work items
}

// ------------------------------------------------------------------------------------- f32
// Exact path: v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fmaf chain.  Per canonical chunk the
// lane state restarts; the range keeps the exact (maximum, lowest key).
template <int DP>  // padded D (multiple of 2): k-steps = DP / 2
__global__ __launch_bounds__(kThreads) void corr_f32_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, int P, int N, int D, int ldq, int ldk,
    int range_chunks, CorrWs ws) {
  constexpr int KS = DP / 2;
  constexpr int LD = DP + 1;  // odd dword stride: conflict-free ds_read_b32 down a column
  if (gated_off(ws)) return;
  constexpr int TKF = (DP <= 64) ? kTK : 32;     // keys per LDS stage (D = 128: 2 x 32 x 129 x 4 B = 33 KB)
  __shared__ float lds[2][TKF * LD];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.y;
  const int q0 = (blockIdx.x * kWaves + wave) * (kQB * 32);

  float bq[kQB][KS];
#pragma unroll
  for (int qb = 0; qb < kQB; ++qb) {
    int row = q0 + qb * 32 + r;
    row = row < P ? row : P - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int col = 2 * s + h;
      bq[qb][s] = col < D ? Q[(size_t)row * ldq + col] : 0.f;
    }
  }
  LaneState st[kQB];
  float gm[kQB];
  int gbi[kQB];
#pragma unroll
  for (int qb = 0; qb < kQB; ++qb) { gm[qb] = -__builtin_inff(); gbi[qb] = 0; }

  const int c0 = split * range_chunks;
  const int cend = min((N + kChunk - 1) / kChunk, c0 + range_chunks);
  // A workgroup whose 256 queries are all the zero vector (the padding rows behind a crop's masked pixels in a
  // capacity-sized batch, isr_prep_queries_batch) writes what the loop below would compute for them — every logit
  // exactly 0: chunk maximum 0 at the chunk's first key, chunk sum = the number of keys (an exact f32) — and leaves.
  bool nonzero = false;
#pragma unroll
  for (int qb = 0; qb < kQB; ++qb)
#pragma unroll
    for (int s = 0; s < KS; ++s) nonzero |= bq[qb][s] != 0.f;
  if (!__syncthreads_or(nonzero ? 1 : 0)) {
#pragma unroll
    for (int qb = 0; qb < kQB; ++qb) {
      const int q = q0 + qb * 32 + r;
      if (h != 0 || q >= P) continue;
      for (int c = c0; c < cend; ++c) {
        ws.pmc[(size_t)c * P + q] = 0.f;
        ws.plc[(size_t)c * P + q] = (float)(min(N, (c + 1) * kChunk) - c * kChunk);
      }
      ws.pm[(size_t)split * P + q] = 0.f;
      ws.pbi[(size_t)split * P + q] = c0 * kChunk;
    }
    return;
  }
  constexpr int NEL = (TKF * DP + kThreads - 1) / kThreads;
  float stg[NEL];
  for (int c = c0; c < cend; ++c) {
    const int k0 = c * kChunk;
    const int k1 = min(N, k0 + kChunk);
    const int nstage = (k1 - k0 + TKF - 1) / TKF;
#pragma unroll
    for (int qb = 0; qb < kQB; ++qb) { st[qb].m = -__builtin_inff(); st[qb].M2 = kNoM2; st[qb].l = 0.f; st[qb].bi = 0; }
    auto gload = [&](int stage) {
#pragma unroll
      for (int i = 0; i < NEL; ++i) {
        const int e = tid + i * kThreads;
        const int row = e / DP, col = e % DP;
        const int key = k0 + stage * TKF + row;
        stg[i] = (e < TKF * DP && key < k1 && col < D) ? K[(size_t)key * ldk + col] : 0.f;
      }
    };
    auto lwrite = [&](int buf) {
#pragma unroll
      for (int i = 0; i < NEL; ++i) {
        const int e = tid + i * kThreads;
        if (e < TKF * DP) lds[buf][(e / DP) * LD + (e % DP)] = stg[i];
      }
    };
    __syncthreads();                    // the previous chunk's last reads are done
    gload(0);
    lwrite(0);
    __syncthreads();
    for (int stage = 0; stage < nstage; ++stage) {
      const int buf = stage & 1;
      if (stage + 1 < nstage) gload(stage + 1);
#pragma unroll
      for (int sub = 0; sub < TKF / 32; ++sub) {
        const int kb = k0 + stage * TKF + sub * 32;
        if (kb < k1) {
          const float* arow = &lds[buf][(sub * 32 + r) * LD + h];
          f32x16 acc[kQB];
#pragma unroll
          for (int qb = 0; qb < kQB; ++qb) acc[qb] = splat16(0.f);
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const float a = arow[2 * s];
#pragma unroll
            for (int qb = 0; qb < kQB; ++qb)
              acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[qb][s], acc[qb], 0, 0, 0);
          }
          const int krow0 = kb + 4 * h;
          if (kb + 32 > k1) {
#pragma unroll
            for (int qb = 0; qb < kQB; ++qb) mask_tail(acc[qb], krow0, k1);
          }
#pragma unroll
          for (int qb = 0; qb < kQB; ++qb) consume_tile(acc[qb], krow0, st[qb]);
        }
      }
      if (stage + 1 < nstage) lwrite(buf ^ 1);
      __syncthreads();
    }
#pragma unroll
    for (int qb = 0; qb < kQB; ++qb) {
      const LaneState mgd = merged_with_other_half(st[qb]);
      const int q = q0 + qb * 32 + r;
      if (h == 0 && q < P) {
        ws.pmc[(size_t)c * P + q] = mgd.m;
        ws.plc[(size_t)c * P + q] = mgd.l;
      }
      if (mgd.m > gm[qb]) { gm[qb] = mgd.m; gbi[qb] = mgd.bi; }   // ascending chunks: ties keep the lower key
    }
  }
#pragma unroll
  for (int qb = 0; qb < kQB; ++qb) {
    const int q = q0 + qb * 32 + r;
    if (h == 0 && q < P) {
      const size_t off = (size_t)split * P + q;
      ws.pm[off] = gm[qb];
      ws.pbi[off] = gbi[qb];
    }
  }
}

// ------------------------------------------------------- exact-f32 K1 on the bf16 matrix cores (split route)
// The f32 MFMA runs on the SIMD's FP32 lanes: it cannot overlap with the softmax epilogue, and corr_f32_kernel pays MFMA time
// plus VALU time per tile (profiles/r03_k1_f32_pmc.txt).  The bf16 MFMA runs beside the VALU.  An f32 number is the exact sum
// of three bf16 numbers (x = x1 + x2 + x3, 8 + 8 + 8 mantissa bits), so <q, k> = sum over the nine plane pairs; the six
// pairs down to 2^-16 — q1k1, q1k2, q2k1, q2k2, q1k3, q3k1 — are ONE bf16 dot product of the 96-wide rows
//     Q' = [q1 | q1 | q2 | q2 | q1 | q3 | 0 | 0]     K' = [k1 | k2 | k1 | k2 | k3 | k1 | 0 | 0]      (blocks of 16, D <= 16)
// with products exact in f32 and f32 accumulation: logits to f32 accuracy (the dropped pairs are 2^-24 relative) from
// corr_bf16_direct_kernel<8> at its D = 128 rate — with the queries pre-multiplied by log2 e, the log2-domain kernel.
// Indices stay EXACT: the margin test's error bound also covers, in units of 2^-24 |q log2 e| |k|, the dropped pairs q2k3, q3k2,
// q3k3 and the last-bit residuals of the two three-way splits (<= 4), the rounding of q log2 e (<= 1) and the rounding of the
// f32 chain itself (<= D <= 16): 21 x 2^-24 |q~||k| <= 3.6 x 2^-23 sqrt(|q'|^2 |k'|^2) since |q'||k'| >= 2.97 |q~||k| — the
// bound (D' + 2) 2^-23 sqrt(|q'|^2 |k'|^2) with D' = 128 grows by 2.8 %, the key-norm term is inflated by 1.08 (3.9 %) — and
// what it cannot certify goes to corr_recheck_kernel, which decides by the f32 fmaf chain of the ORIGINAL rows — the logit
// corr_f32_kernel and the oracle compute — lowest key on ties.  logp / lse come from the split logits (a few 1e-7 of the
// chain's).  A query's result depends on (query, keys) only, as on every path.
__device__ __forceinline__ uint16_t bf16_rne(float x, float* rem) {
  uint32_t u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  u &= 0xFFFF0000u;
  *rem = x - __uint_as_float(u);          // exact: the discarded low bits
  return (uint16_t)(u >> 16);
}

// rows (R, ld) f32, D <= 16 -> (R, 128) bf16; QUERY: [x1 x1 x2 x2 x1 x3 0 0] of x * prescale, else [x1 x2 x1 x2 x3 x1 0 0]
template <bool QUERY>
__global__ __launch_bounds__(256) void corr_split_f32_kernel(const float* __restrict__ X, int R, int D, int ld, float prescale,
                                                             uint16_t* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)R * 16) return;
  const long row = i >> 4;
  const int d = (int)(i & 15);
  float x = d < D ? X[row * ld + d] : 0.f;
  if (QUERY) x = x * prescale;
  float r1, r2, r3;
  const uint16_t x1 = bf16_rne(x, &r1), x2 = bf16_rne(r1, &r2), x3 = bf16_rne(r2, &r3);
  uint16_t* o = out + row * 128 + d;
  if (QUERY) { o[0] = x1; o[16] = x1; o[32] = x2; o[48] = x2; o[64] = x1; o[80] = x3; }
  else       { o[0] = x1; o[16] = x2; o[32] = x1; o[48] = x2; o[64] = x3; o[80] = x1; }
  o[96] = 0; o[112] = 0;
}

// rows (R, ld) f32, D <= 16 SP -> (R, 48 SP) bf16: the planes [x1 | x2 | x3] of x * prescale, 16 SP columns each, zero beyond D
template <int SP>
__global__ __launch_bounds__(256) void corr_split3_f32_kernel(const float* __restrict__ X, int R, int D, int ld, float prescale,
                                                              uint16_t* __restrict__ out) {
  constexpr int W = 16 * SP;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)R * W) return;
  const long row = i / W;
  const int d = (int)(i % W);
  const float x = d < D ? X[row * ld + d] * prescale : 0.f;     // prescale = 1 (keys): exact
  float r1, r2, r3;
  const uint16_t x1 = bf16_rne(x, &r1), x2 = bf16_rne(r1, &r2), x3 = bf16_rne(r2, &r3);
  uint16_t* o = out + row * (3 * W) + d;
  o[0] = x1; o[W] = x2; o[2 * W] = x3;
}

// rows (R, ld) f32, D <= 16 SP -> (R, 48 SP) f16 planes [x1 | x2s | x1s] of x * prescale (RowFrags' header): x1 = f16(x),
// x2s = f16((x - x1) 2^11), x1s = f16(x1 2^-11); *ovf is raised when an |x| does not fit f16
template <int SP>
__global__ __launch_bounds__(256) void corr_split2h_f32_kernel(const float* __restrict__ X, int R, int D, int ld, float prescale,
                                                               uint16_t* __restrict__ out, int32_t* __restrict__ ovf) {
  constexpr int W = 16 * SP;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)R * W) return;
  const long row = i / W;
  const int d = (int)(i % W);
  const float x = d < D ? X[row * ld + d] * prescale : 0.f;
  if (fabsf(x) > 65000.f) *ovf = 1;                            // (finite descriptors are a precondition of every route)
  const _Float16 x1 = (_Float16)x;                              // v_cvt_f16_f32: round to nearest even
  const float r = x - (float)x1;                                // exact
  const _Float16 x2s = (_Float16)(r * 2048.f);
  const _Float16 x1s = (_Float16)((float)x1 * 4.8828125e-4f);   // exact unless subnormal (|x1| < 2^-3): then <= 2^-25 off
  uint16_t* o = out + row * (3 * W) + d;
  o[0] = __builtin_bit_cast(uint16_t, x1); o[W] = __builtin_bit_cast(uint16_t, x2s); o[2 * W] = __builtin_bit_cast(uint16_t, x1s);
}

// ------------------------------------------------------------------------------ key norms
// max_n |k_n|^2 for the error bound of the margin test, one partial per block; block 0 also zeroes
// the recheck counter of this call.
template <bool F16 = false>
__global__ __launch_bounds__(256) void corr_keynorm_kernel(const uint16_t* __restrict__ K, int N, int D,
                                                           int ldk, float inflate, CorrWs ws) {
  __shared__ float red[4];
  // (before the gate: when the f16-plane kernels are gated off, the chain kernels behind them count into the same histogram)
  if (ws.digits)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ws.ndigits; i += kKnBlocks * 256) ws.digits[i] = 0;
  if (gated_off(ws)) return;
  float mx = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ws.nhand; i += kKnBlocks * 256) ws.hand[i] = 0;     // the screened route's hand-over flags
  for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += kKnBlocks * 256) {
    const uint16_t* row = K + (size_t)n * ldk;
    float s = 0.f;
    for (int d = 0; d < D; d += 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(row + d);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = elem_f32<F16>((uint16_t)(w[j] & 0xFFFFu)), hi = elem_f32<F16>((uint16_t)(w[j] >> 16));
        s = __builtin_fmaf(lo, lo, s);
        s = __builtin_fmaf(hi, hi, s);
      }
    }
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    ws.kn2[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * inflate;   // inflate > 1: the split-f32 route's extra error terms
    if (blockIdx.x == 0) {
      ws.rcount[0] = 0; ws.rcount[1] = 0; ws.rcount[2] = 0; ws.rcount[3] = 0;
      if (ws.kmax) { ws.kmax[0] = 0u; ws.kmax[1] = 0u; ws.redone[0] = 0ull; ws.redone[1] = 0ull; }
    }
  }
}

// ------------------------------------------------------------------------------- finalize
// Per query: the winner over the key ranges (ascending, strict >: ties keep the lower key), the
// canonical f64 sum over the chunks (added in ascending order starting from the first chunk — exactly what
// the direct kernel does in registers when ONE workgroup owns the whole key range, in which case it also
// finishes the query itself and nothing is written for this kernel to read)
//     L = sum_c l_c 2^(R_c - Rf),   Rf = max_c R_c
// and   logp = -(ln L + Rf ln2 - m)   [natural logits; log2-domain logits: -(ln L + (Rf - m') ln2)],
// formed without the m - lse cancellation.  MODE 0: f32 path (every chunk carries its maximum).
// MODE 1: bf16 log2 domain, MODE 2: bf16 natural units — chunks of good queries have R_c = 0, chunks of
// bad (range, query) pairs R_c = ceil(chunk maximum [* log2 e]); then the margin test:
//     |MFMA logit - exact logit| <= eps = (D + 2) 2^-23 |q| max|k|
// holds for ANY order of at-least-faithfully-rounded f32 additions of the D exact products (Higham's
// gamma_n bound with u = 2^-23; Cauchy-Schwarz on sum |q_d k_d|), so a winner more than 2 eps above the
// runner-up is the exact arg-max; everything else goes on the recheck list with the threshold below
// which no key can be the exact winner.
// The last step for one query, shared by corr_finalize_kernel and the direct kernel's own epilogue (same
// code, same f64 operations: the two routes give bit-identical outputs): log-probability and lse from the
// canonical sum, then the margin test.
// One count per finished query in its image's digit histogram: lanes of the wave that hold the same (image, digit) send ONE
// integer atomic between them (a launch of the bench adds 9.8 M counts to ~20 bins per image) — integer atomics: the
// histogram does not depend on the order in which the waves arrive.
__device__ __forceinline__ void digit_count(const CorrWs& ws, int q, float lp) {
  const int img = q / ws.rows_per_image, r = q - img * ws.rows_per_image;
  const bool valid = !ws.n_rows || r < ws.n_rows[img];
  const int key = valid ? img * isr::kDigitBins + (int)(isr::ordered_bits(lp) >> isr::kDigitShift) : -1;
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
  while (todo != 0ull) {                                   // wave-uniform
    const int leader = (int)__builtin_ctzll(todo);
    const int k = __builtin_amdgcn_readlane(key, leader);
    const unsigned long long same = __builtin_amdgcn_ballot_w64(key == k);
    if (lane == leader) atomicAdd(&ws.digits[k], (int)__builtin_popcountll(same));
    todo &= ~same;
  }
}

template <int MODE>
__device__ __forceinline__ void corr_finish(int q, float G1, float G2, int bi, bool anybad, double L, double Rf, int D, float eabs,
                                            float qn2, float kn2, const CorrWs& ws, int32_t* __restrict__ idx,
                                            float* __restrict__ logp, float* __restrict__ lse) {
  const double ln2 = 0.6931471805599453094;
  // <= 0 like log_softmax (the sum contains the maximum's own term); rounding can leave it a few 1e-6 above
  // lse = ln L + Rf ln 2 from the canonical sum alone (round 4: it used to be formed as maximum - logp, which tied its last bit
  // to the maximum; an lse-only call — idx == nullptr — has no maximum, and must return the same bits)
  const double lnL = log(L);
  const double lp = (MODE == 1) ? fmin(0.0, -(lnL + (Rf - (double)G1) * ln2)) : fmin(0.0, -(lnL + (Rf * ln2 - (double)G1)));
  if (lse) lse[q] = (float)(lnL + Rf * ln2);
  if (!idx) return;                       // lse-only call: no index to certify
  idx[q] = bi;
  if (logp) logp[q] = (float)lp;
  if (ws.digits) digit_count(ws, q, (float)lp);
  // The zero vector (a padding row of a capacity-sized crop batch, isr_prep_queries_batch): every product is an exact
  // zero, every logit is exactly 0, the arg-max is the lowest key — which is what `bi` already holds — and there is
  // nothing an exact recheck could decide differently.  (With eps = 0 the margin test below would list every such row:
  // 4 600 of the 5 625 rows of a typical crop, each re-scanned against all keys.)
  if (MODE != 0 && qn2 == 0.f && !anybad) return;
  if (MODE != 0) {
    // eabs (f16 planes only): what subnormal plane elements add absolutely, eabs (|q| + |k|)
    const double eps = (double)(D + 2) * 1.1920928955078125e-7 * sqrt((double)qn2 * (double)kn2) * 1.0001 +
                       (double)eabs * (sqrt((double)qn2) + sqrt((double)kn2));
    const double margin = (double)G1 - (double)G2;
    if (anybad || !(margin > 2.0 * eps)) {
      // exact winner x* >= exact(winner) >= G1 - eps, so its MFMA logit is >= G1 - 2 eps; bad ranges carry
      // logits shifted by their reference (error up to ~3 eps) and the f32 rounding of M2 + mr
      const double thr = anybad ? (double)G1 - 8.0 * eps - 1e-5 * fabs((double)G1) : (double)G1 - 2.0 * eps;
      ws.qn2[q] = (float)(thr - 1e-6 * fabs(thr) - 1e-30);      // rounded down
      ws.rlist[atomicAdd(ws.rcount, 1)] = q;
    }
  }
}

// max_n |k_n|^2 as a SCALAR (every lane holds the same value after the butterfly; readfirstlane moves it to an SGPR, where it
// costs the loop nothing to keep: called twice from the direct kernel, the vector form left its six shuffle addresses alive
// across the loop for the second call, and in the 168-register plain-row kernel that was three spilled registers per lane —
// the 12 B per query of scratch traffic behind profiles/k1_hbm_traffic.json's 2.5 x write figure)
__device__ __forceinline__ float kn2_max(const CorrWs& ws) {
  float v = ws.kn2[threadIdx.x & 63];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}

#include "corr_direct.hpp"   // corr_bf16_direct_kernel: the VALU-minimal bf16 loop (log2 and natural units)
#include "corr_sparse.hpp"   // the screened route for D = 64: FP6 screen, exact bf16 pieces

template <int MODE>
__global__ __launch_bounds__(256) void corr_finalize_kernel(int P, int D, float eabs, int nsplit, int range_chunks,
                                                            int nchunks, CorrWs ws, int32_t* __restrict__ idx,
                                                            float* __restrict__ logp, float* __restrict__ lse) {
  if (gated_off(ws)) return;
  // bf16, ONE key range: the direct kernel finished its good queries itself; only workgroups that hold a bad
  // query (the same 256-query blocks there and here) have anything left to do
  // (there the grid is a fixed one striding over the direct kernel's list of such blocks: nothing listed, nothing done)
  const bool finished_in_kernel = MODE != 0 && nsplit == 1;
  const int nblocks = finished_in_kernel ? ws.rcount[1] : (int)gridDim.x;
  const float kn2 = MODE != 0 ? kn2_max(ws) : 0.f;
  for (int e = blockIdx.x; e < nblocks; e += gridDim.x) {
  const int qblock = finished_in_kernel ? ws.blist[e] : e;
  const int q = qblock * blockDim.x + threadIdx.x;
  if (q >= P) continue;
  if (finished_in_kernel && !ws.pbad[q]) continue;
  float G1 = -__builtin_inff(), G2 = -__builtin_inff();
  int bi = 0;
  bool anybad = false;
  for (int r = 0; r < nsplit; ++r) {
    const size_t o = (size_t)r * P + q;
    if (MODE != 0 && ws.pbad[o]) { anybad = true; continue; }
    const float m = ws.pm[o];
    if (m > G1) { G2 = fmaxf(G2, G1); G1 = m; bi = ws.pbi[o]; }
    else G2 = fmaxf(G2, m);
    if (MODE != 0) G2 = fmaxf(G2, ws.pm2[o]);
  }
  double L = 0.0, Rf = -__builtin_inf();
  for (int c = 0; c < nchunks; ++c) {
    const size_t o = (size_t)c * P + q;
    float R = 0.f;
    if (MODE == 0 || ws.pbad[(size_t)(c / range_chunks) * P + q]) {
      const float mc = ws.pmc[o];
      R = (MODE == 1) ? ceilf(mc) : ceilf(mc * kLog2e);
      if (MODE != 0) {                                    // approximate maximum of a bad range
        if (mc > G1) { G2 = fmaxf(G2, G1); G1 = mc; } else G2 = fmaxf(G2, mc);
      }
    }
    const double l = ws.plc[o];
    if ((double)R > Rf) { L = L * exp2(Rf - (double)R) + l; Rf = R; }
    else L += l * exp2((double)R - Rf);
  }
  corr_finish<MODE>(q, G1, G2, bi, anybad, L, Rf, D, eabs, MODE != 0 ? ws.qn2[q] : 0.f, kn2, ws, idx, logp, lse);
  }  // query blocks
}

// ------------------------------------------------------------------------------ exact recheck
// exact <q, k> of two bf16 rows: products of bf16 values are exact in f64 and are added in ascending d
// exactly as oracle/isr_oracle.c:orc_corr_argmax_bf16 does, then scaled like the oracle's logit.
__device__ __attribute__((noinline)) double exact_logit(const uint16_t* __restrict__ qrow, const uint16_t* __restrict__ krow,
                                              int D, double scale) {
  double acc = 0.0;
  for (int d = 0; d < D; d += 8) {
    const uint4 qa = *reinterpret_cast<const uint4*>(qrow + d);
    const uint4 ka = *reinterpret_cast<const uint4*>(krow + d);
    const uint32_t qw[4] = {qa.x, qa.y, qa.z, qa.w}, kw[4] = {ka.x, ka.y, ka.z, ka.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc += (double)__uint_as_float(qw[j] << 16) * (double)__uint_as_float(kw[j] << 16);
      acc += (double)__uint_as_float(qw[j] & 0xFFFF0000u) * (double)__uint_as_float(kw[j] & 0xFFFF0000u);
    }
  }
  return acc * scale;
}

// The listed queries (gathered through rlist) against one key range per blockIdx.y, on the same MFMA
// chain as the main kernels (C = 0: the same f32 logits); every element at or above the query's
// threshold is evaluated exactly; per (range, list entry) the best exact value and its lowest key.
// Key ranges of the recheck pass: its cost is the latency of ONE workgroup streaming its key range for a group of listed
// queries, so a short list (the usual case: ~0.1 % of the queries) is spread over up to kRSplitGrid ranges — as many as
// the rval / ridx scratch (rsplit * P entries, sized for a list of all P queries over rsplit ranges) holds for the actual
// list length.  (With 4 ranges fixed the pass took 0.43 ms of a 1.6 ms estimate_pose call at P = 5 476, N = 80 000.)
// The outcome does not depend on the partition: ranges merge in ascending order, ties keep the lower key.
constexpr int kRSplitGrid = 64;
// A LONG list needs no key ranges at all: every (group of 256 listed queries, range) unit gathers the group's query rows again and
// pays its own prologue, so ranges are halved while half as many still leave kRUnits units for the launch's slots (configs[3]'s
// keys send 6.8 % of 9.8 M queries here: 2 608 groups x 64 ranges of 7 stages each took 8.9 ms per launch; see the header of
// profiles/r04_k1_recheck_ranges.txt).
#ifndef ISR_K1_RECHECK_UNITS
#define ISR_K1_RECHECK_UNITS 2048
#endif
constexpr long kRUnits = ISR_K1_RECHECK_UNITS;
__device__ __forceinline__ int recheck_ranges(int cnt, int P, int rsplit, int nstage_all) {
  const long cap = (long)rsplit * P;
  const long groups = ((long)cnt + kWaves * kQB * 32 - 1) / (kWaves * kQB * 32);
  int rs = kRSplitGrid;
  while (rs > 1 && ((long)rs * cnt > cap || rs > nstage_all || groups * (rs >> 1) >= kRUnits)) rs >>= 1;
  return rs;
}

// F32 originals (split-f32 route): the candidates are decided by the k-ordered f32 fmaf chain of the ORIGINAL rows —
// the exact-f32 kernel's (and the oracle's) logit — instead of the exact products of the bf16 rows.
struct F32Rows { const float* q; const float* k; int ldq, ldk, D; };

__device__ __attribute__((noinline)) double chain_logit_f32(const float* __restrict__ qrow, const float* __restrict__ krow, int D) {
  float acc = 0.f;
  for (int d = 0; d < D; ++d) acc = __builtin_fmaf(qrow[d], krow[d], acc);
  return (double)acc;
}

#ifndef ISR_K1_RECHECK_WAVES
// waves per SIMD corr_recheck_kernel is compiled for on plain rows of D <= 64.  3 = 168 registers (96 B of spills) instead of 200:
// 7.7 against 8.4 ms on configs[3]'s long lists (profiles/r04_k1_recheck_ranges.txt), and — what made it the default in round 5 —
// a wave that fits the slot a K1 wave leaves: since the closing kernels of a call run BESIDE the next call's chip-filling kernel
// (isr_corr_argmax_phase), the 200-register build waited for two K1 waves to leave one SIMD together (up to 26 ms in a trace)
#define ISR_K1_RECHECK_WAVES 3
#endif
template <int DK, int SP = 0, bool F16 = false>
__global__ __launch_bounds__(kThreads, (DK <= 4 && SP == 0) ? ISR_K1_RECHECK_WAVES : 1) void corr_recheck_kernel(
    const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K, int P, int N, int ldq, int ldk,
    int rsplit, double scale, F32Rows f32, CorrWs ws) {
  using KS = KeyStage<DK, SP>;
  constexpr int NFR = KS::NFR, TKS = KS::TK;     // the key ranges below are cut in units of kTK keys, staged TKS at a time
  __shared__ uint4 lds[2][KS::CHUNKS];
  if (gated_off(ws)) return;
  const int cnt = *ws.rcount;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int nstage_all = (N + kTK - 1) / kTK;
  const int rs = recheck_ranges(cnt, P, rsplit, nstage_all);
  // the launch is a flat list of workgroups: rs key ranges x (gridDim.x / rs) slots striding over the groups of listed queries
  const int range = blockIdx.x % rs, slot = blockIdx.x / rs, nslots = gridDim.x / rs;
  const int per = (nstage_all + rs - 1) / rs;
  const int k0 = range * per * kTK;
  const int k1 = min(N, k0 + per * kTK);
  for (int g = slot; g * (kWaves * kQB * 32) < cnt; g += nslots) {   // block-uniform
    const int e0 = (g * kWaves + wave) * (kQB * 32);
    bf16x8 bq[kQB][NFR];
    float thr[kQB];
    int qrow[kQB];
    double best[kQB];
    int bidx[kQB];
#pragma unroll
    for (int qb = 0; qb < kQB; ++qb) {
      const int e = e0 + qb * 32 + r;
      qrow[qb] = ws.rlist[e < cnt ? e : cnt - 1];
      thr[qb] = e < cnt ? ws.qn2[qrow[qb]] : __builtin_inff();
      const uint16_t* src = Q + (size_t)qrow[qb] * ldq + 8 * h;
#pragma unroll
      for (int s = 0; s < NFR; ++s) bq[qb][s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
      best[qb] = -__builtin_inf();
      bidx[qb] = -1;
    }
    const int nstage = k1 > k0 ? (k1 - k0 + TKS - 1) / TKS : 0;
    KS ks;
    __syncthreads();                    // the previous group's last LDS reads are done
    if (nstage > 0) {
      ks.gload(K, ldk, k0, k1);
      ks.lwrite(lds[0]);
    }
    __syncthreads();
    for (int stage = 0; stage < nstage; ++stage) {
      const int buf = stage & 1;
      if (stage + 1 < nstage) ks.gload(K, ldk, k0 + (stage + 1) * TKS, k1);
#pragma unroll
      for (int sub = 0; sub < TKS / 32; ++sub) {
        const int kb = k0 + stage * TKS + sub * 32;
        if (kb < k1) {  // block-uniform
          bf16x8 a[NFR];
          KS::load_a(lds[buf], sub, r, h, a);
#pragma unroll
          for (int qb = 0; qb < kQB; ++qb) {
            f32x16 acc = tile_chain<DK, SP, F16>(a, bq[qb], splat16(0.f));
            if (kb + 32 > k1) mask_tail(acc, kb + 4 * h, k1);
            const float t = tile_max(acc);
            if (__any(t >= thr[qb])) {                    // rare
              if (t >= thr[qb]) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {            // ascending key inside the lane: strict > keeps the lowest
                  if (acc[i] >= thr[qb]) {
                    const int n = kb + 4 * h + (i & 3) + 8 * (i >> 2);
#ifdef ISR_ABL_RECHECK_NOEXACT   // timing-only ablation: what the in-line exact evaluations cost the pass (results are the f32 arg-max)
                    const double v = (double)acc[i];
#else
                    const double v = f32.q ? chain_logit_f32(f32.q + (size_t)qrow[qb] * f32.ldq, f32.k + (size_t)n * f32.ldk, f32.D)
                                           : exact_logit(Q + (size_t)qrow[qb] * ldq, K + (size_t)n * ldk, 16 * DK, scale);
#endif
                    if (v > best[qb] || (v == best[qb] && n < bidx[qb])) { best[qb] = v; bidx[qb] = n; }
                  }
                }
              }
            }
          }
        }
      }
      if (stage + 1 < nstage) ks.lwrite(lds[buf ^ 1]);
      __syncthreads();
    }
#pragma unroll
    for (int qb = 0; qb < kQB; ++qb) {
      const double ov = __shfl_xor(best[qb], 32, 64);
      const int oi = __shfl_xor(bidx[qb], 32, 64);
      if (oi >= 0 && (bidx[qb] < 0 || ov > best[qb] || (ov == best[qb] && oi < bidx[qb]))) { best[qb] = ov; bidx[qb] = oi; }
      const int e = e0 + qb * 32 + r;
      if (h == 0 && e < cnt) {
        ws.rval[(size_t)range * cnt + e] = best[qb];
        ws.ridx[(size_t)range * cnt + e] = bidx[qb];
      }
    }
  }
}

// list entry e: best over the recheck key ranges (ascending: ties keep the lower key) -> idx[rlist[e]]
__global__ __launch_bounds__(256) void corr_recheck_merge_kernel(int P, int N, int rsplit, CorrWs ws,
                                                                 int32_t* __restrict__ idx) {
  if (gated_off(ws)) return;
  const int cnt = *ws.rcount;
  const int rs = recheck_ranges(cnt, P, rsplit, (N + kTK - 1) / kTK);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += gridDim.x * blockDim.x) {
    double best = -__builtin_inf();
    int bi = -1;
#pragma unroll 8
    for (int r = 0; r < rs; ++r) {
      const int i = ws.ridx[(size_t)r * cnt + e];
      const double v = ws.rval[(size_t)r * cnt + e];
      if (i >= 0 && (bi < 0 || v > best)) { best = v; bi = i; }
    }
    if (bi >= 0) idx[ws.rlist[e]] = bi;
  }
}

// ------------------------------------------------------------------------------------ plan
constexpr int kMaxSplit = 64;     // upper bound on key ranges per launch
constexpr int kSlotCap = 2048;    // device-independent cap on the resident-workgroup estimate

struct CorrPlan {
  int qblocks, nchunks, nsplit, range_chunks, rsplit;
};

constexpr int kSmallQ = 8192;     // query blocks below which a few key ranges are always allowed (tail balancing)

inline int max_split_for(int qblocks, int nchunks) {
  int ns = (kSlotCap + qblocks - 1) / qblocks;
  if (qblocks < kSmallQ && ns < 4) ns = 4;
  if (ns > nchunks) ns = nchunks;
  if (ns > kMaxSplit) ns = kMaxSplit;
  return ns < 1 ? 1 : ns;
}

// Work units = (query block, key range of whole chunks); every resident slot runs one unit at a time and the hardware
// hands out the next unit as slots free up, so a launch lasts  ceil(units / slots) x (one unit).  The number of key ranges
// is the one that minimises that, with a unit priced at its chunks plus 0.3 chunk for what every range pays on its own
// (prologue, row recovery, partials for the finalize pass).  Round 3: the old rule — split only when the query blocks
// alone cannot fill the machine, and then just enough to give every slot one unit — put 588 units on 512 slots at
// P = 50 176, N = 80 000: two rounds for 1.15 rounds of work (1.20 ms; 0.74 ms with 5 ranges).
// ISR_TUNE_K1_SPLIT > 0 forces the number of ranges (capped as above): a caller that knows most of its rows are zero
// padding (crop batches: ~40 % dense) asks for finer units than the nominal row count suggests.
// The split changes which workgroup computes a chunk, never a chunk's arithmetic.
CorrPlan make_plan(int P, int N, int slots, int q_per_block) {
  CorrPlan p;
  p.qblocks = (P + q_per_block - 1) / q_per_block;
  p.nchunks = (N + kChunk - 1) / kChunk;
  if (slots > kSlotCap) slots = kSlotCap;
  if (slots < 1) slots = 1;
  const int cap = max_split_for(p.qblocks, p.nchunks);
  int ns = 1;
  const int forced = isr::tuning(ISR_TUNE_K1_SPLIT);
  if (forced > 0) {
    ns = forced < cap ? forced : cap;
  } else {
    double best = 1e300;
    for (int c = 1; c <= cap; ++c) {
      const int rc = (p.nchunks + c - 1) / c;
      const int nsp = (p.nchunks + rc - 1) / rc;
      if (nsp != c) continue;                                  // the same plan as a smaller c
      const long units = (long)p.qblocks * nsp;
      const double cost = (double)((units + slots - 1) / slots) * ((double)rc + 0.3);
      if (cost < best - 1e-9) { best = cost; ns = c; }
    }
  }
  p.range_chunks = (p.nchunks + ns - 1) / ns;
  p.nsplit = (p.nchunks + p.range_chunks - 1) / p.range_chunks;
  const int nstage = (N + kTK - 1) / kTK;
  p.rsplit = nstage < kRSplitMax ? nstage : kRSplitMax;
  return p;
}

template <typename Kern>
int resident_slots(Kern kern) {
  int dev = 0, cus = 256, per_cu = 4;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kThreads, 0) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      per_cu = 4;
    }
  } else {
    (void)hipGetLastError();
  }
  return cus * per_cu;
}

int slots_for(int dtype, int D) {
  if (dtype == ISR_DTYPE_BF16_LOG2_SCREENED) dtype = ISR_DTYPE_BF16_LOG2;     // (a screened call plans one key range whatever this says)
  // cached per (kernel family, padded D): the occupancy query costs tens of microseconds
  static int cache[3][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
  int v = 0;
  if (dtype != ISR_DTYPE_F32) v = (D <= 16) ? 0 : (D <= 32) ? 1 : (D <= 64) ? 2 : 3;
  else v = (D <= 8) ? 0 : (D <= 16) ? 1 : (D <= 32) ? 2 : (D <= 64) ? 3 : 4;
  int& c = cache[dtype == ISR_DTYPE_F32 ? 1 : dtype == ISR_DTYPE_BF16_LOG2 ? 2 : 0][v];
  if (c == 0) {
    if (dtype == ISR_DTYPE_BF16_LOG2) {
      c = v == 0 ? resident_slots(corr_bf16_direct_kernel<1, kQB, false>)
        : v == 1 ? resident_slots(corr_bf16_direct_kernel<2, kQB, false>)
        : v == 2 ? resident_slots(corr_bf16_direct_kernel<4, kQB, false>)
                 : resident_slots(corr_bf16_direct_kernel<8, kQB, false>);
    } else if (dtype != ISR_DTYPE_F32) {
      c = v == 0 ? resident_slots(corr_bf16_direct_kernel<1, kQB, true>)
        : v == 1 ? resident_slots(corr_bf16_direct_kernel<2, kQB, true>)
        : v == 2 ? resident_slots(corr_bf16_direct_kernel<4, kQB, true>)
                 : resident_slots(corr_bf16_direct_kernel<8, kQB, true>);
    } else {
      c = v == 0 ? resident_slots(corr_f32_kernel<8>) : v == 1 ? resident_slots(corr_f32_kernel<16>)
        : v == 2 ? resident_slots(corr_f32_kernel<32>) : v == 3 ? resident_slots(corr_f32_kernel<64>)
                 : resident_slots(corr_f32_kernel<128>);
    }
  }
  return c;
}

int slots_planes(int sp, bool f16) {
  static int cache[2][9] = {{0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0}};
  int& c = cache[f16 ? 1 : 0][sp];
  if (c == 0) {
    if (f16)
      c = sp == 1 ? resident_slots(corr_bf16_direct_kernel<3, kQB, false, 3, 1, true>)
        : sp == 2 ? resident_slots(corr_bf16_direct_kernel<6, kQB, false, 6, 2, true>)
        : sp == 4 ? resident_slots(corr_bf16_direct_kernel<12, kQB, false, 12, 4, true>)
                  : 256;      // SP = 8: one workgroup per CU (96 KB of LDS, 512 registers per wave)
    else
      c = sp == 1 ? resident_slots(corr_bf16_direct_kernel<3, kQB, false, 3, 1>)
        : sp == 2 ? resident_slots(corr_bf16_direct_kernel<6, kQB, false, 6, 2>)
                  : resident_slots(corr_bf16_direct_kernel<12, kQB, false, 12, 4>);
  }
  return c;
}

// carve the workspace for the worst plan the call can take (the split bound does not depend on the device)
size_t carve(isr::Workspace& w, int P, int N, int dtype, CorrWs* o) {
  const int qblocks = (P + kWaves * kQB * 32 - 1) / (kWaves * kQB * 32);
  const int nchunks = (N + kChunk - 1) / kChunk;
  const int ns = max_split_for(qblocks, nchunks);
  const int nstage = (N + kTK - 1) / kTK;
  const int rs = nstage < kRSplitMax ? nstage : kRSplitMax;
  const bool bf16 = dtype != ISR_DTYPE_F32;
  o->pm = w.take<float>((size_t)ns * P);
  o->pbi = w.take<int32_t>((size_t)ns * P);
  o->plc = w.take<float>((size_t)nchunks * P);
  o->pmc = w.take<float>((size_t)nchunks * P);
  o->pm2 = bf16 ? w.take<float>((size_t)ns * P) : nullptr;
  o->pbad = bf16 ? w.take<int32_t>((size_t)ns * P) : nullptr;
  o->qn2 = bf16 ? w.take<float>(P) : nullptr;
  o->flags = bf16 ? w.take<int32_t>((size_t)ns * qblocks) : nullptr;
  o->blist = bf16 ? w.take<int32_t>((size_t)ns * qblocks) : nullptr;
  o->kn2 = bf16 ? w.take<float>(kKnBlocks) : nullptr;
  o->rcount = bf16 ? w.take<int32_t>(4) : nullptr;
  o->clk = bf16 ? w.take<long long>(2) : nullptr;
  o->rlist = bf16 ? w.take<int32_t>(P) : nullptr;
  o->rval = bf16 ? w.take<double>((size_t)rs * P) : nullptr;
  o->ridx = bf16 ? w.take<int32_t>((size_t)rs * P) : nullptr;
  const bool scr = dtype == ISR_DTYPE_BF16_LOG2_SCREENED;
  o->lowbuf = scr ? w.take<float>((size_t)P * 10) : nullptr;      // kLowStride floats per query
  o->q6 = scr ? w.take<uint8_t>((size_t)P * 64) : nullptr;
  o->k6 = scr ? w.take<uint8_t>((size_t)N * 64) : nullptr;
  o->qnrm = scr ? w.take<float>((size_t)P * 2) : nullptr;
  o->kmax = scr ? w.take<uint32_t>(4) : nullptr;
  o->redone = scr ? w.take<unsigned long long>(2) : nullptr;
  o->hand = scr ? w.take<int32_t>(qblocks) : nullptr;
  o->nhand = scr ? qblocks : 0;
  o->skip = nullptr;
  o->only = nullptr;
  o->lower = nullptr;
  o->lower_stride = 1;
  o->skip_T = 42.f;
  o->skip_default = -__builtin_inff();
  o->digits = nullptr;
  o->n_rows = nullptr;
  o->rows_per_image = 1;
  o->ndigits = 0;
  return w.off;
}

// the split route of the f32 path (D <= 16): the two 128-wide bf16 images, then a bf16 workspace
size_t carve_split(isr::Workspace& w, int P, int N, uint16_t** q2, uint16_t** k2, CorrWs* o) {
  *q2 = w.take<uint16_t>((size_t)P * 128);
  *k2 = w.take<uint16_t>((size_t)N * 128);
  return carve(w, P, N, ISR_DTYPE_BF16_LOG2, o);
}

constexpr int kSplitMaxD = 16;    // round-3 form of the split route (96-wide rows on the generic direct kernel)
constexpr int kSplit3MaxD = 64;   // plane forms with two waves per SIMD: SP = 1, 2, 4 blocks per plane
constexpr int kSplitF16MaxD = 128;  // f16 planes also at SP = 8 (one wave per SIMD, dynamic LDS)

// How an f32 call runs.  kind: 0 the f32-MFMA chain kernel; 1 round 3's 96-wide split (D <= 16); 2 three bf16 planes;
// 3 f16 planes (gated: falls through to the chain kernel when a descriptor does not fit f16).  ISR_TUNE_K1_F32_CHAIN:
// 0 default (f16 planes for D <= 64), 1 chain kernel everywhere, 2 bf16 planes for every D <= 64, 3 f16 planes (= default),
// 4 round 3's 96-wide split for D <= 16 (bf16 planes above).
struct F32Route { int kind, sp; };
F32Route f32_route(int D) {
  const int t = isr::tuning(ISR_TUNE_K1_F32_CHAIN);
  if (t == 1 || D > kSplitF16MaxD) return {0, 0};
  const int sp = D <= 16 ? 1 : D <= 32 ? 2 : D <= 64 ? 4 : 8;
  if (t == 4 && D <= kSplitMaxD) return {1, 0};
  if (t == 2 || t == 4) return D <= kSplit3MaxD ? F32Route{2, sp} : F32Route{0, 0};      // bf16 planes stop at D = 64
  return {3, sp};
}

// the plane routes: the two (rows, 48 SP) 16-bit images and a gate word, then a bf16 workspace; gated (f16 planes): the
// f32 workspace of the chain kernels behind it as well
size_t carve_planes(isr::Workspace& w, int P, int N, int SP, bool gated, uint16_t** q3, uint16_t** k3, int32_t** gate, CorrWs* o,
                    CorrWs* chain) {
  *q3 = w.take<uint16_t>((size_t)P * 48 * SP);
  *k3 = w.take<uint16_t>((size_t)N * 48 * SP);
  *gate = w.take<int32_t>(4);
  carve(w, P, N, ISR_DTYPE_BF16_LOG2, o);
  if (gated) {
    CorrWs tmp;
    carve(w, P, N, ISR_DTYPE_F32, chain ? chain : &tmp);
  }
  return w.off;
}

}  // namespace

extern "C" size_t isr_corr_argmax_workspace_bytes(int P, int N, int D, int dtype) {
  if (P <= 0 || N <= 0) return 0;
  isr::Workspace w(nullptr, 0);
  CorrWs o;
  size_t bytes = carve(w, P, N, dtype, &o) + 256;
  if (dtype == ISR_DTYPE_F32 && (D <= 0 || D <= kSplitMaxD)) {     // either f32 route fits (D = 0: unknown, assume the larger)
    isr::Workspace w2(nullptr, 0);
    uint16_t *q2, *k2;
    const size_t split = carve_split(w2, P, N, &q2, &k2, &o) + 256;
    if (split > bytes) bytes = split;
  }
  if (dtype == ISR_DTYPE_F32 && D <= kSplitF16MaxD) {              // the plane routes (any knob setting: the size is a function of the shape)
    isr::Workspace w3(nullptr, 0);
    uint16_t *q3, *k3;
    int32_t* gate;
    const size_t split = carve_planes(w3, P, N, D <= 0 ? 8 : D <= 16 ? 1 : D <= 32 ? 2 : D <= 64 ? 4 : 8, true, &q3, &k3, &gate, &o, nullptr) + 256;
    if (split > bytes) bytes = split;
  }
  return bytes;
}


// Diagnostics: the shader clock the direct kernel actually ran at in the LAST call on this workspace:
// workgroup (0, 0) reads s_memtime (counts at the shader clock) and s_memrealtime (constant 100 MHz) when it
// starts and when it ends (~0.8 ms at the bench's shape, under full load).  Synchronises the stream.
extern "C" int isr_corr_argmax_clock_mhz(const void* ws_, size_t ws_bytes, int P, int N, int dtype,
                                         double* mhz_host, isr_stream_t stream_) {
  ISR_REQUIRE(ws_ && mhz_host && P > 0 && N > 0, "isr_corr_argmax_clock_mhz: bad argument");
  ISR_REQUIRE(ws_bytes >= isr_corr_argmax_workspace_bytes(P, N, 128, dtype), "isr_corr_argmax_clock_mhz: workspace too small");
  *mhz_host = 0.0;
  if (dtype == ISR_DTYPE_F32) return ISR_OK;
  isr::Workspace w(const_cast<void*>(ws_), ws_bytes);
  CorrWs ws;
  carve(w, P, N, dtype, &ws);
  hipStream_t stream = isr::as_stream(stream_);
  long long t[2] = {0, 0};
  ISR_CHECK_HIP(hipMemcpyAsync(t, ws.clk, sizeof t, hipMemcpyDeviceToHost, stream));
  ISR_CHECK_HIP(hipStreamSynchronize(stream));
  if (t[1] > 0) *mhz_host = 100.0 * (double)t[0] / (double)t[1];
  return ISR_OK;
}

// Diagnostics: how many queries of the LAST call that used this workspace (same P, N, dtype) went to
// the exact recheck.  Synchronises the stream.  *count_host = -1 for the f32 path (no recheck).
extern "C" int isr_corr_argmax_recheck_count(const void* ws_, size_t ws_bytes, int P, int N, int dtype,
                                             int32_t* count_host, isr_stream_t stream_) {
  ISR_REQUIRE(ws_ && count_host && P > 0 && N > 0, "isr_corr_argmax_recheck_count: bad argument");
  ISR_REQUIRE(ws_bytes >= isr_corr_argmax_workspace_bytes(P, N, 128, dtype), "isr_corr_argmax_recheck_count: workspace too small");
  *count_host = -1;
  if (dtype == ISR_DTYPE_F32) return ISR_OK;
  isr::Workspace w(const_cast<void*>(ws_), ws_bytes);
  CorrWs ws;
  carve(w, P, N, dtype, &ws);
  hipStream_t stream = isr::as_stream(stream_);
  ISR_CHECK_HIP(hipMemcpyAsync(count_host, ws.rcount, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  ISR_CHECK_HIP(hipStreamSynchronize(stream));
  return ISR_OK;
}

// Parity hook of the screened route: the block-scaled FP6 image of bf16 rows of 64 columns exactly as isr_corr_argmax forms it
// (tests compare it with oracle/fp6_screen_oracle.py bit for bit; a caller could also keep a key image across calls one day).
extern "C" int isr_corr_quantize_fp6(const void* X, int R, int ld, void* out, float* nrm, float* kmax, isr_stream_t stream_) {
  ISR_REQUIRE(X && out && nrm && kmax && R > 0 && ld >= 64 && ld % 8 == 0 && ((uintptr_t)X % 16 == 0) && ((uintptr_t)out % 16 == 0),
              "isr_corr_quantize_fp6: null pointer, R=%d, or rows not 16-byte aligned (ld=%d)", R, ld);
  hipStream_t stream = isr::as_stream(stream_);
  CorrWs ws{};
  const unsigned g = (unsigned)((2l * R + 255) / 256);
  ISR_CHECK_HIP(hipMemsetAsync(kmax, 0, 2 * sizeof(float), stream));
  corr_quant_fp6_kernel<true><<<g, 256, 0, stream>>>(static_cast<const uint16_t*>(X), R, ld, static_cast<uint8_t*>(out), nrm, nullptr, ws);
  corr_quant_fp6_kernel<false><<<g, 256, 0, stream>>>(static_cast<const uint16_t*>(X), R, ld, static_cast<uint8_t*>(out), nullptr,
                                                      reinterpret_cast<uint32_t*>(kmax), ws);
  ISR_CHECK_LAUNCH("fp6 quantisation kernels");
  return ISR_OK;
}

// Diagnostics of the screened route: tile items redone exactly in the last call on this workspace.  Synchronises the stream.
extern "C" int isr_corr_argmax_screen_redone(const void* ws_, size_t ws_bytes, int P, int N, int dtype, long long* count_host,
                                             isr_stream_t stream_) {
  ISR_REQUIRE(ws_ && count_host && P > 0 && N > 0, "isr_corr_argmax_screen_redone: bad argument");
  ISR_REQUIRE(ws_bytes >= isr_corr_argmax_workspace_bytes(P, N, 128, dtype), "isr_corr_argmax_screen_redone: workspace too small");
  count_host[0] = 0;
  count_host[1] = 0;
  if (dtype != ISR_DTYPE_BF16_LOG2_SCREENED) return ISR_OK;
  isr::Workspace w(const_cast<void*>(ws_), ws_bytes);
  CorrWs ws;
  carve(w, P, N, dtype, &ws);
  hipStream_t stream = isr::as_stream(stream_);
  ISR_CHECK_HIP(hipMemcpyAsync(count_host, ws.redone, 2 * sizeof(long long), hipMemcpyDeviceToHost, stream));
  ISR_CHECK_HIP(hipStreamSynchronize(stream));
  return ISR_OK;
}

// the same for an f32 call (D <= 16 with the split route active: the length of its f32-chain recheck list; -1 otherwise)
extern "C" int isr_corr_argmax_recheck_count_f32(const void* ws_, size_t ws_bytes, int P, int N, int D,
                                                 int32_t* count_host, isr_stream_t stream_) {
  ISR_REQUIRE(ws_ && count_host && P > 0 && N > 0 && D > 0, "isr_corr_argmax_recheck_count_f32: bad argument");
  ISR_REQUIRE(ws_bytes >= isr_corr_argmax_workspace_bytes(P, N, D, ISR_DTYPE_F32), "isr_corr_argmax_recheck_count_f32: workspace too small");
  *count_host = -1;
  const F32Route route = f32_route(D);
  if (route.kind == 0) return ISR_OK;
  isr::Workspace w(const_cast<void*>(ws_), ws_bytes);
  uint16_t *q2, *k2;
  int32_t* gate;
  CorrWs ws;
  if (route.kind == 1) carve_split(w, P, N, &q2, &k2, &ws);
  else carve_planes(w, P, N, route.sp, route.kind == 3, &q2, &k2, &gate, &ws, nullptr);
  hipStream_t stream = isr::as_stream(stream_);
  int32_t gated = 0;
  if (route.kind == 3) ISR_CHECK_HIP(hipMemcpyAsync(&gated, gate, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  ISR_CHECK_HIP(hipMemcpyAsync(count_host, ws.rcount, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  ISR_CHECK_HIP(hipStreamSynchronize(stream));
  if (gated) *count_host = -1;           // the call fell through to the f32-MFMA chain kernels (a descriptor beyond f16's range)
  return ISR_OK;
}

namespace {

// the bf16 kernels of one call (direct kernel, per-query fallback, finalize, exact recheck, merge).  f32.q != nullptr: the
// split-f32 route — Q / K are the 128-wide split images, the recheck decides by the f32 chain of the original rows.
// sp > 0: the three-plane split — Q / K are (rows, 48 sp) plane images, D = split_deff(sp) (the margin test's bound).
// f16: the planes are f16 (RowFrags), eabs the margin test's absolute term.
int launch_bf16(const uint16_t* q, const uint16_t* k, int P, int N, int D, int ldq, int ldk, bool log2, const CorrPlan& p,
                const CorrWs& ws, int32_t* idx, float* logp, float* lse, F32Rows f32, float kn_inflate, hipStream_t stream,
                int sp = 0, bool f16 = false, bool screened = false, int phase = 3) {
  // phase bit 0 ("open"): the key-norm kernel and the chip-filling kernel(s); bit 1 ("close"): fallback, finalize, recheck, merge —
  // small launches at low occupancy that a caller may put on another stream beside the NEXT call's chip-filling kernel
  // (isr_corr_argmax_phase)
  const bool ph1 = (phase & 1) != 0, ph2 = (phase & 2) != 0;
  const float eabs = (sp && f16) ? split_eabs(sp) : 0.f;
  const bool lse_only = idx == nullptr;      // no maxima, no recovery, no recheck (LSE instantiations where they exist)
  const dim3 grid(p.qblocks, p.nsplit);
  const int fin_blocks = (P + 255) / 256;
  const int fin_bf16 = (p.nsplit == 1 && fin_blocks > kFallbackGrid) ? kFallbackGrid : fin_blocks;   // one key range: list-driven
  ISR_REQUIRE(sp != 0 || D == 16 || D == 32 || D == 64 || D == 128,
              "isr_corr_argmax(bf16): D=%d must be 16, 32, 64 or 128 (zero-pad the columns)", D);
  ISR_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0),
              "isr_corr_argmax(bf16): rows must be 16-byte aligned (ldq=%d ldk=%d)", ldq, ldk);
  ISR_REQUIRE((long long)p.range_chunks * kChunk * ldk * 2 < (1ll << 31),
              "isr_corr_argmax(bf16): a key range of %d rows x ldk=%d exceeds the 2 GiB buffer window",
              p.range_chunks * kChunk, ldk);
  const int cgrid = (int)(((long)p.qblocks * p.nchunks < kFallbackGrid) ? (long)p.qblocks * p.nchunks : kFallbackGrid);   // fallback: strides over listed blocks x chunks
  const dim3 rgrid(16 * kRSplitGrid);   // workgroups = key ranges (by list length, <= 64) x slots striding over the groups of 256 listed queries
  const double scale = log2 ? 0.6931471805599453094 : 1.0;   // the oracle's logit_scale
  if (ph1) {
    if (f16) corr_keynorm_kernel<true><<<kKnBlocks, 256, 0, stream>>>(k, N, 16 * sp, ldk, kn_inflate, ws);
    else corr_keynorm_kernel<false><<<kKnBlocks, 256, 0, stream>>>(k, N, sp ? 16 * sp : D, ldk, kn_inflate, ws);   // planes: |k1|^2
  }
#define ISR_LAUNCH_BF16(DKv)                                                                                  \
  do {                                                                                                        \
    if (log2) {                                                                                               \
      if (ph1) {                                                                                              \
      if (lse_only && DKv <= 2)   /* LSE kernels where both copies of their loop fit the registers: D <= 32 */ \
        corr_bf16_direct_kernel<DKv, kQB, false, DKv, 0, false, (DKv <= 2)><<<grid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, \
                                                                              p.range_chunks, ws, idx, logp, lse); \
      else                                                                                                    \
      corr_bf16_direct_kernel<DKv, kQB, false><<<grid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk,          \
                                                                              p.range_chunks, ws, idx, logp, lse); \
      }                                                                                                       \
      if (ph2) {                                                                                              \
      corr_bf16_kernel<DKv, true><<<cgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.range_chunks, p.qblocks,      \
                                                                  p.nchunks, ws);  \
      corr_finalize_kernel<1><<<fin_bf16, 256, 0, stream>>>(P, D, eabs, p.nsplit, p.range_chunks, p.nchunks, ws, \
                                                              idx, logp, lse);                               \
      }                                                                                                       \
    } else {                                                                                                  \
      if (ph1)                                                                                                \
      corr_bf16_direct_kernel<DKv, kQB, true><<<grid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk,           \
                                                                             p.range_chunks, ws, idx, logp, lse); \
      if (ph2) {                                                                                              \
      corr_bf16_kernel<DKv, false><<<cgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.range_chunks, p.qblocks,     \
                                                                   p.nchunks, ws); \
      corr_finalize_kernel<2><<<fin_bf16, 256, 0, stream>>>(P, D, eabs, p.nsplit, p.range_chunks, p.nchunks, ws, \
                                                              idx, logp, lse);                               \
      }                                                                                                       \
    }                                                                                                         \
    if (ph2 && !lse_only) corr_recheck_kernel<DKv><<<rgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.rsplit, scale, f32, ws);  \
  } while (0)
#define ISR_LAUNCH_PLANES(SPv, F16v)                                                                                              \
  do {                                                                                                                            \
    if (ph1) {                                                                                                                    \
    if (lse_only && F16v && SPv <= 2)                                                                                             \
      corr_bf16_direct_kernel<3 * SPv, kQB, false, 3 * SPv, SPv, F16v, (F16v && SPv <= 2)><<<grid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk,  \
                                                                                                     p.range_chunks, ws, idx, logp, lse); \
    else                                                                                                                          \
    corr_bf16_direct_kernel<3 * SPv, kQB, false, 3 * SPv, SPv, F16v><<<grid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk,          \
                                                                                                     p.range_chunks, ws, idx, logp, lse); \
    }                                                                                                                             \
    if (ph2) {                                                                                                                    \
    corr_bf16_kernel<3 * SPv, true, SPv, F16v><<<cgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.range_chunks, p.qblocks,   \
                                                                                p.nchunks, ws);                                    \
    corr_finalize_kernel<1><<<fin_bf16, 256, 0, stream>>>(P, D, eabs, p.nsplit, p.range_chunks, p.nchunks, ws, idx, logp, lse);   \
    if (!lse_only)                                                                                                                \
      corr_recheck_kernel<3 * SPv, SPv, F16v><<<rgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.rsplit, scale, f32, ws);    \
    }                                                                                                                             \
  } while (0)
  if (sp) {          // plane routes (log2 domain, f32 originals decide the recheck)
    switch (sp * 2 + (f16 ? 1 : 0)) {
      case 2: ISR_LAUNCH_PLANES(1, false); break;
      case 3: ISR_LAUNCH_PLANES(1, true); break;
      case 4: ISR_LAUNCH_PLANES(2, false); break;
      case 5: ISR_LAUNCH_PLANES(2, true); break;
      case 8: ISR_LAUNCH_PLANES(4, false); break;
      case 9: ISR_LAUNCH_PLANES(4, true); break;
      default: {     // SP = 8, f16 planes: two 48 KB stage buffers as dynamic LDS
        constexpr int kDynLds = 2 * 64 * 48 * 16;
        static bool attr_set = false;
        if (!attr_set) {
          ISR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_bf16_direct_kernel<24, kQB, false, 24, 8, true, false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kDynLds));
          attr_set = true;
        }
        if (ph1)
          corr_bf16_direct_kernel<24, kQB, false, 24, 8, true, false><<<grid, kThreads, kDynLds, stream>>>(q, k, P, N, ldq, ldk, p.range_chunks,
                                                                                                         ws, idx, logp, lse);
        if (ph2) {
          corr_bf16_kernel<24, true, 8, true><<<cgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.range_chunks, p.qblocks, p.nchunks, ws);
          corr_finalize_kernel<1><<<fin_bf16, 256, 0, stream>>>(P, D, eabs, p.nsplit, p.range_chunks, p.nchunks, ws, idx, logp, lse);
          if (!lse_only) corr_recheck_kernel<24, 8, true><<<rgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.rsplit, scale, f32, ws);
        }
        break;
      }
    }
  } else
#undef ISR_LAUNCH_PLANES
  if (f32.q) {       // split-f32 route: 128-wide rows whose last two blocks are zero, log2 domain
    if (ph1) corr_bf16_direct_kernel<8, kQB, false, 6><<<grid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.range_chunks, ws, idx, logp, lse);
    if (ph2) {
      corr_bf16_kernel<8, true><<<cgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.range_chunks, p.qblocks, p.nchunks, ws);
      corr_finalize_kernel<1><<<fin_bf16, 256, 0, stream>>>(P, D, eabs, p.nsplit, p.range_chunks, p.nchunks, ws, idx, logp, lse);
      if (!lse_only) corr_recheck_kernel<8><<<rgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.rsplit, scale, f32, ws);
    }
  } else
  if (screened) {     // the screened route (corr_sparse.hpp): one key range, D = 64, log2 domain
    ISR_REQUIRE(p.nsplit == 1 && D == 64 && log2, "isr_corr_argmax: the screened route runs D = 64 log2-domain rows over one key range");
    CorrWs w2 = ws;
    w2.lower = ws.lowbuf;
    w2.lower_stride = kLowStride;
    w2.skip_T = (float)screen_T(N);
    const unsigned gq = (unsigned)((2l * P + 255) / 256), gk = (unsigned)((2l * N + 255) / 256);
    if (ph1) {
      corr_quant_fp6_kernel<false><<<gk, 256, 0, stream>>>(k, N, ldk, ws.k6, nullptr, ws.kmax, ws);
      corr_quant_fp6_kernel<true><<<gq, 256, 0, stream>>>(q, P, ldq, ws.q6, ws.qnrm, nullptr, ws);
      corr_fp6_lower_kernel<<<(P + kQPB0 - 1) / kQPB0, kThreads, 0, stream>>>(ws.q6, ws.k6, q, k, P, N, ldq, ldk, ws.lowbuf, ws);
      corr_fp6_sparse_kernel<<<(P + kQPB1 - 1) / kQPB1, kThreads, 0, stream>>>(ws.q6, ws.k6, q, k, P, N, ldq, ldk, ws.qnrm, ws.kmax, w2, idx, logp, lse);
      // query blocks pass 1 handed over (most of their first tiles had to be redone: flat logits): the dense kernel with the same
      // rule — a piece counts when its exact maximum reaches L_q - T — on the same bf16 logits; every other block leaves at once
      corr_bf16_direct_kernel<4, kQB, false, 4, 0, false, false, 1><<<grid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.range_chunks, w2, idx, logp, lse);
    }
    if (ph2) {
      corr_bf16_kernel<4, true><<<cgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.range_chunks, p.qblocks, p.nchunks, w2);
      corr_finalize_kernel<1><<<fin_bf16, 256, 0, stream>>>(P, D, eabs, p.nsplit, p.range_chunks, p.nchunks, w2, idx, logp, lse);
      if (!lse_only) corr_recheck_kernel<4><<<rgrid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.rsplit, scale, f32, w2);
    }
  } else
  switch (D) {
    case 16: ISR_LAUNCH_BF16(1); break;
    case 32: ISR_LAUNCH_BF16(2); break;
    case 64: ISR_LAUNCH_BF16(4); break;
    default: ISR_LAUNCH_BF16(8); break;
  }
#undef ISR_LAUNCH_BF16
  if (ph2 && !lse_only) corr_recheck_merge_kernel<<<64, 256, 0, stream>>>(P, N, p.rsplit, ws, idx);
  ISR_CHECK_LAUNCH("corr bf16 kernels");
  return ISR_OK;
}

// the f32-MFMA chain kernels of one call
int launch_f32_chain(const float* q, const float* k, int P, int N, int D, int ldq, int ldk, const CorrPlan& p, const CorrWs& ws,
                     int32_t* idx, float* logp, float* lse, hipStream_t stream, int phase = 3) {
  ISR_REQUIRE(D <= 128, "isr_corr_argmax(f32): D=%d > 128", D);
  const bool ph1 = (phase & 1) != 0, ph2 = (phase & 2) != 0;
  const dim3 grid(p.qblocks, p.nsplit);
  const int fin_blocks = (P + 255) / 256;
  const float eabs = 0.f;
  if (ph1) {
    if (D <= 8) corr_f32_kernel<8><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.range_chunks, ws);
    else if (D <= 12) corr_f32_kernel<12><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.range_chunks, ws);   // the reference's 12-D descriptors: 6 k-steps, not 8
    else if (D <= 16) corr_f32_kernel<16><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.range_chunks, ws);
    else if (D <= 32) corr_f32_kernel<32><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.range_chunks, ws);
    else if (D <= 64) corr_f32_kernel<64><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.range_chunks, ws);
    else corr_f32_kernel<128><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.range_chunks, ws);
  }
  if (ph2) corr_finalize_kernel<0><<<fin_blocks, 256, 0, stream>>>(P, D, eabs, p.nsplit, p.range_chunks, p.nchunks, ws, idx, logp, lse);
  ISR_CHECK_LAUNCH("corr f32 kernels");
  return ISR_OK;
}

struct DigitArgs {
  int32_t* hist = nullptr;
  const int32_t* n_rows = nullptr;
  int rows_per_image = 1;
  int images = 0;
  int phase = 3;          // bit 0: the opening kernels (pre-processing, key norms, the chip-filling kernels); bit 1: the closing ones
  void apply(CorrWs* w) const {
    w->digits = hist;
    w->n_rows = n_rows;
    w->rows_per_image = rows_per_image;
    w->ndigits = images * isr::kDigitBins;
  }
};

int corr_argmax_impl(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk,
                     int dtype, int32_t* idx, float* logp, float* lse, void* ws_,
                     size_t ws_bytes, isr_stream_t stream_, const DigitArgs& dg) {
  ISR_REQUIRE(Q && K && (idx || (lse && !logp)),
              "isr_corr_argmax: null pointer (idx may be null only for an lse-only call: logp null, lse given)");
  ISR_REQUIRE(P > 0 && N > 0 && D > 0, "isr_corr_argmax: P=%d N=%d D=%d must be positive", P, N, D);
  ISR_REQUIRE(ldq >= D && ldk >= D, "isr_corr_argmax: ldq=%d ldk=%d < D=%d", ldq, ldk, D);
  ISR_REQUIRE(dtype == ISR_DTYPE_BF16 || dtype == ISR_DTYPE_BF16_LOG2 || dtype == ISR_DTYPE_F32 || dtype == ISR_DTYPE_BF16_LOG2_SCREENED,
              "isr_corr_argmax: dtype %d", dtype);
  if (!ws_ || ws_bytes < isr_corr_argmax_workspace_bytes(P, N, D, dtype)) {
    isr::set_error("isr_corr_argmax: workspace %zu < %zu", ws_bytes,
                   isr_corr_argmax_workspace_bytes(P, N, D, dtype));
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  const F32Route route = dtype == ISR_DTYPE_F32 ? f32_route(D) : F32Route{0, 0};
  if (route.kind >= 2) {
    // plane routes: exact indices and f32-accurate sums from the 16-bit matrix cores at D <= 64 (RowFrags' header)
    const int sp = route.sp;
    const bool f16 = route.kind == 3;
    isr::Workspace w(ws_, ws_bytes);
    uint16_t *q3, *k3;
    int32_t* gate;
    CorrWs ws, cws;
    carve_planes(w, P, N, sp, f16, &q3, &k3, &gate, &ws, &cws);
    dg.apply(&ws);
    dg.apply(&cws);
    const CorrPlan p = make_plan(P, N, slots_planes(sp, f16), kWaves * kQB * 32);
    const float* qf = static_cast<const float*>(Q);
    const float* kf = static_cast<const float*>(K);
    const unsigned gq = (unsigned)(((long)P * 16 * sp + 255) / 256), gk = (unsigned)(((long)N * 16 * sp + 255) / 256);
#define ISR_SPLIT_PLANES(SPv)                                                                              \
  do {                                                                                                     \
    if (f16) {                                                                                             \
      corr_split2h_f32_kernel<SPv><<<gq, 256, 0, stream>>>(qf, P, D, ldq, kLog2e, q3, gate);               \
      corr_split2h_f32_kernel<SPv><<<gk, 256, 0, stream>>>(kf, N, D, ldk, 1.f, k3, gate);                  \
    } else {                                                                                               \
      corr_split3_f32_kernel<SPv><<<gq, 256, 0, stream>>>(qf, P, D, ldq, kLog2e, q3);                      \
      corr_split3_f32_kernel<SPv><<<gk, 256, 0, stream>>>(kf, N, D, ldk, 1.f, k3);                         \
    }                                                                                                      \
  } while (0)
    if (dg.phase & 1) {
      if (f16) ISR_CHECK_HIP(hipMemsetAsync(gate, 0, sizeof(int32_t), stream));
      switch (sp) {
        case 1: ISR_SPLIT_PLANES(1); break;
        case 2: ISR_SPLIT_PLANES(2); break;
        case 4: ISR_SPLIT_PLANES(4); break;
        default:      // SP = 8 exists with f16 planes only (f32_route)
          corr_split2h_f32_kernel<8><<<gq, 256, 0, stream>>>(qf, P, D, ldq, kLog2e, q3, gate);
          corr_split2h_f32_kernel<8><<<gk, 256, 0, stream>>>(kf, N, D, ldk, 1.f, k3, gate);
          break;
      }
    }
#undef ISR_SPLIT_PLANES
    if (f16) ws.skip = gate;
    const int rc = launch_bf16(q3, k3, P, N, f16 ? split_deff_f16(sp) : split_deff(sp), 48 * sp, 48 * sp, true, p, ws, idx, logp, lse,
                               F32Rows{qf, kf, ldq, ldk, D}, 1.f, stream, sp, f16, false, dg.phase);
    if (rc != ISR_OK || !f16) return rc;
    // behind the gate: the f32-MFMA chain kernels, which leave at once unless a descriptor did not fit f16
    cws.only = gate;
    return launch_f32_chain(qf, kf, P, N, D, ldq, ldk, make_plan(P, N, slots_for(ISR_DTYPE_F32, D), kWaves * kQB * 32), cws, idx,
                            logp, lse, stream, dg.phase);
  }
  if (route.kind == 1) {
    // round 3's split route: 96-wide rows on the generic direct kernel (corr_split_f32_kernel's header)
    isr::Workspace w(ws_, ws_bytes);
    uint16_t *q2, *k2;
    CorrWs ws;
    carve_split(w, P, N, &q2, &k2, &ws);
    dg.apply(&ws);
    const CorrPlan p = make_plan(P, N, slots_for(ISR_DTYPE_BF16_LOG2, 128), kWaves * kQB * 32);
    const float* qf = static_cast<const float*>(Q);
    const float* kf = static_cast<const float*>(K);
    if (dg.phase & 1) {
      corr_split_f32_kernel<true><<<(unsigned)(((long)P * 16 + 255) / 256), 256, 0, stream>>>(qf, P, D, ldq, kLog2e, q2);
      corr_split_f32_kernel<false><<<(unsigned)(((long)N * 16 + 255) / 256), 256, 0, stream>>>(kf, N, D, ldk, 1.f, k2);
    }
    return launch_bf16(q2, k2, P, N, 128, 128, 128, true, p, ws, idx, logp, lse, F32Rows{qf, kf, ldq, ldk, D}, 1.08f, stream, 0, false,
                       false, dg.phase);
  }
  CorrPlan p = make_plan(P, N, slots_for(dtype, D), kWaves * kQB * 32);
  const bool screened = screened_route(dtype, N, D);
  if (screened) {     // one key range: L_q and the pieces' canonical order are defined over the whole range
    p.nsplit = 1;
    p.range_chunks = p.nchunks;
  }
  isr::Workspace w(ws_, ws_bytes);
  CorrWs ws;
  carve(w, P, N, dtype, &ws);
  dg.apply(&ws);
  if (dtype != ISR_DTYPE_F32) {
    return launch_bf16(static_cast<const uint16_t*>(Q), static_cast<const uint16_t*>(K), P, N, D, ldq, ldk,
                       dtype != ISR_DTYPE_BF16, p, ws, idx, logp, lse, F32Rows{nullptr, nullptr, 0, 0, 0}, 1.f, stream, 0, false, screened,
                       dg.phase);
  }
  // (the chain kernels alone: no key-norm kernel runs ahead of them to zero the histogram)
  if (dg.hist && (dg.phase & 1))
    ISR_CHECK_HIP(hipMemsetAsync(dg.hist, 0, sizeof(int32_t) * (size_t)dg.images * isr::kDigitBins, stream));
  return launch_f32_chain(static_cast<const float*>(Q), static_cast<const float*>(K), P, N, D, ldq, ldk, p, ws, idx, logp, lse, stream,
                          dg.phase);
}

}  // namespace

extern "C" int isr_corr_argmax(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk,
                               int dtype, int32_t* idx, float* logp, float* lse, void* ws_,
                               size_t ws_bytes, isr_stream_t stream_) {
  return corr_argmax_impl(Q, K, P, N, D, ldq, ldk, dtype, idx, logp, lse, ws_, ws_bytes, stream_, DigitArgs{});
}

extern "C" int isr_corr_argmax_digits(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk,
                                      int dtype, int32_t* idx, float* logp, float* lse, int rows_per_image,
                                      const int32_t* n_rows, int32_t* digit_hist, void* ws_, size_t ws_bytes,
                                      isr_stream_t stream_) {
  ISR_REQUIRE(idx && logp && digit_hist, "isr_corr_argmax_digits: idx, logp and digit_hist are required (the digits are those of logp)");
  ISR_REQUIRE(rows_per_image > 0 && P > 0 && P % rows_per_image == 0,
              "isr_corr_argmax_digits: P=%d is not a whole number of images of %d rows", P, rows_per_image);
  DigitArgs dg;
  dg.hist = digit_hist;
  dg.n_rows = n_rows;
  dg.rows_per_image = rows_per_image;
  dg.images = P / rows_per_image;
  return corr_argmax_impl(Q, K, P, N, D, ldq, ldk, dtype, idx, logp, lse, ws_, ws_bytes, stream_, dg);
}

extern "C" int isr_corr_argmax_phase(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk,
                                     int dtype, int32_t* idx, float* logp, float* lse, int rows_per_image,
                                     const int32_t* n_rows, int32_t* digit_hist, int phase, void* ws_, size_t ws_bytes,
                                     isr_stream_t stream_) {
  ISR_REQUIRE(phase >= 1 && phase <= 3, "isr_corr_argmax_phase: phase=%d (1 open, 2 close, 3 both)", phase);
  DigitArgs dg;
  dg.phase = phase;
  if (digit_hist) {
    ISR_REQUIRE(idx && logp, "isr_corr_argmax_phase: digit_hist needs idx and logp (the digits are those of logp)");
    ISR_REQUIRE(rows_per_image > 0 && P > 0 && P % rows_per_image == 0,
                "isr_corr_argmax_phase: P=%d is not a whole number of images of %d rows", P, rows_per_image);
    dg.hist = digit_hist;
    dg.n_rows = n_rows;
    dg.rows_per_image = rows_per_image;
    dg.images = P / rows_per_image;
  }
  return corr_argmax_impl(Q, K, P, N, D, ldq, ldk, dtype, idx, logp, lse, ws_, ws_bytes, stream_, dg);
}
