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
// lane-local — no cross-lane traffic until one 2-lane merge at the end.
//
// Per-lane online state for its query:  m  (max logit, exact, for the arg-max; lowest key on ties)
//   M2 = ceil(m * log2 e)  (an integer, so exp2(fma(s, log2e, -M2)) has one rounding and every
//                           rescale  l *= 2^(M2old - M2new)  is exact)
//   l  = sum_n 2^(s_n log2e - M2)
// Cost per 32x32 tile at D = 64: 4 MFMA (128 matrix-pipe cycles) against ~310 cycles of VALU issue
// (16 v_exp_f32 at 8.3, 16 adds at 2.8, 8 v_max3 at 4.6, the MFMAs' own ~18 each): the loop is bound
// by VALU issue, almost half of it the transcendental unit — DESIGN.md section 4 has the measurements.
//
// dtype bf16: v_mfma_f32_32x32x16_bf16.  dtype f32: v_mfma_f32_32x32x2_f32, which is bit for bit
// a k-ordered fmaf chain — oracle/isr_oracle.c:orc_corr_argmax_f32 reproduces its logits exactly.
// dtype bf16-log2 (queries prescaled by log2 e): corr_bf16_direct_kernel, the VALU-minimal loop
// (wave-uniform reference in the MFMA C operand, record-only arg-max, post-loop row recovery),
// with corr_bf16_kernel<.., LOG2> as its flagged per-workgroup fallback.
#include "isr_common.hpp"

#include <type_traits>

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kQB = 2;                       // 32-query blocks per wave (f32 kernel)
#ifndef ISR_BF16_QB
#define ISR_BF16_QB 2
#endif
constexpr int kQBbf16 = ISR_BF16_QB;         // 32-query blocks per wave, bf16 kernel (D <= 64)
constexpr int kTK = 128;                     // keys per LDS stage
constexpr float kLog2e = 1.4426950408889634f;
// M2 of a lane that has seen no valid key yet: finite, so a fully masked tile gives
// exp2(fma(-inf, log2e, 1e30)) = 0 instead of NaN, and the first real key rescales l by 2^-huge = 0.
constexpr float kNoM2 = -1.0e30f;

struct LaneState {
  float m;    // running max logit
  float M2;   // ceil(m * log2e)
  float l;    // sum 2^(s*log2e - M2)
  int bi;     // key index of m
};

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// Part A of consuming a tile (rows kb + 4h + (r&3) + 8(r>>2), r = 0..15, of this lane's query):
// the running maximum / arg-max and the integer log2 reference M2.
__device__ __forceinline__ void update_max(const f32x16& acc, int krow0, LaneState& st) {
#ifdef ISR_ABL_NOMAX  // timing-only ablation (tools/ablate_corr.hip): keep acc live, skip part A
  asm volatile("" :: "v"(acc[0]), "v"(acc[15]));
  (void)krow0; (void)st;
  return;
#endif
  const float x0 = max3(acc[0], acc[1], acc[2]), x1 = max3(acc[3], acc[4], acc[5]),
              x2 = max3(acc[6], acc[7], acc[8]), x3 = max3(acc[9], acc[10], acc[11]),
              x4 = max3(acc[12], acc[13], acc[14]);
  const float t = fmaxf(max3(x0, x1, x2), max3(x3, x4, acc[15]));
#ifdef ISR_ABL_NOUPD  // timing-only ablation: maximum without the arg-max/rescale branch
  st.m = fmaxf(st.m, t);
  return;
#endif
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

// Part B of tile `cur` interleaved with the DK MFMAs of the next tile, so one wave keeps both the
// matrix pipe and the VALU/transcendental pipe fed (co-resident waves run this same program in
// lockstep, so overlap cannot be left to chance between waves — measured: 14 % co-execution).
template <int DK>
__device__ __forceinline__ f32x16 exp_and_next_mfma(const f32x16& cur, LaneState& st,
                                                    const bf16x8 (&a)[DK], const bf16x8 (&b)[DK]) {
  constexpr int E = 16 / DK > 0 ? 16 / DK : 1;  // exp elements per MFMA
  const float nM2 = -st.M2;
  float l = st.l;
  f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < DK; ++s) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], c, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = s * E + e;
#ifdef ISR_ABL_NOEXP  // timing-only ablation: keep the accumulator live, no fma/exp/add
      if (i < 16) asm volatile("" :: "v"(cur[i]));
#else
      if (i < 16) l += __builtin_amdgcn_exp2f(__builtin_fmaf(cur[i], kLog2e, nM2));
#endif
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, 3 * E, 0);  // then this step's fma/exp/add
  }
  if (DK * E < 16) {
#pragma unroll
    for (int i = DK * E; i < 16; ++i) l += __builtin_amdgcn_exp2f(__builtin_fmaf(cur[i], kLog2e, nM2));
  }
  st.l = l;
  return c;
}

__device__ __forceinline__ void mask_tail(f32x16& acc, int krow0, int N) {
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (krow0 + (i & 3) + 8 * (i >> 2) >= N) acc[i] = -__builtin_inff();
}

__device__ __forceinline__ bool better(float ma, int ia, float mb, int ib) {
  return (ma > mb) || (ma == mb && ia < ib);
}

__device__ __forceinline__ void merge_state(LaneState& a, const LaneState& b) {
  const float M2 = fmaxf(a.M2, b.M2);
  const float la = (a.M2 == M2) ? a.l : a.l * __builtin_amdgcn_exp2f(a.M2 - M2);
  const float lb = (b.M2 == M2) ? b.l : b.l * __builtin_amdgcn_exp2f(b.M2 - M2);
  if (better(b.m, b.bi, a.m, a.bi)) { a.m = b.m; a.bi = b.bi; }
  a.M2 = M2;
  a.l = la + lb;
}

__device__ __forceinline__ void store_partial(const LaneState& st_, int q, int P, int split,
                                              float* pm, float* pM2, float* pl, int32_t* pbi) {
  LaneState st = st_;
  LaneState o;
  o.m = __shfl_xor(st.m, 32, 64);
  o.M2 = __shfl_xor(st.M2, 32, 64);
  o.l = __shfl_xor(st.l, 32, 64);
  o.bi = __shfl_xor(st.bi, 32, 64);
  merge_state(st, o);
  if ((threadIdx.x & 63) < 32 && q < P) {
    const size_t off = (size_t)split * P + q;
    pm[off] = st.m; pM2[off] = st.M2; pl[off] = st.l; pbi[off] = st.bi;
  }
}


// ---------------------------------------------------------------------- bf16, log2-domain variant
// dtype ISR_DTYPE_BF16_LOG2: the caller multiplied the queries by log2(e) BEFORE rounding them to
// bf16, so the MFMA produces logits in log2 units and exp2 needs no multiply.  The subtraction of
// the integer reference M2 rides in the MFMA's C operand: each query block keeps a 16-register
// tile holding -M2 (rewritten only when M2 moves), the first MFMA of every chain takes it as C and
// writes a different destination, so the accumulator already holds s' - M2 and the epilogue is
// exp2 + add per element: 16 VALU instructions per tile fewer than the natural-log path, no extra
// MFMA.  (Carrying -M2 as an extra k-step instead was measured: the fifth MFMA cost what the 16
// fmas saved.)  MFMA numerics with the large C term: profiles/r01_mfma_numerics.txt (max abs error
// 1.9e-5 over |s' - M2| <= 190, no bias).
struct L2State {
  float mr;   // max(s') - M2 of this lane's rows (-inf before the first key)
  float M2;   // integer reference, SHARED by the two lanes (h = 0, 1) of a query
  float l;    // sum 2^(s' - M2)
  int bi;
};

__device__ __forceinline__ f32x16 splat16(float v) {
  return f32x16{v, v, v, v, v, v, v, v, v, v, v, v, v, v, v, v};
}

// Part A, log2 domain.  acc holds s' - M2; a new maximum above the reference bumps M2 by an integer d
// (both lanes of the query, exchanged with one cross-half shuffle), rescales l by 2^-d exactly,
// shifts this tile's accumulators and rewrites the query block's -M2 tile.
__device__ __forceinline__ void update_max_l2(f32x16& acc, int krow0, L2State& st, f32x16& cinit) {
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
    if (up) {
      int r = 15;
#pragma unroll
      for (int i = 14; i >= 0; --i) r = (acc[i] == t) ? i : r;
      st.bi = krow0 + (r & 3) + 8 * (r >> 2);
      st.mr = t;
    }
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

// Part B, log2 domain, interleaved with the DK MFMAs of the next tile (first one takes C = -M2).
// Compile-time recursion over the steps: sched_group_barrier needs literal group sizes.
template <int DK, int S>
__device__ __forceinline__ void l2_step(const f32x16& cur, float& l, f32x16& c, const bf16x8 (&a)[DK],
                                        const bf16x8 (&b)[DK], const f32x16& cinit) {
  constexpr int E = 16 / DK > 0 ? 16 / DK : 1;
  if constexpr (S == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], cinit, 0, 0, 0);
  else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S], b[S], c, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < E; ++e) l += __builtin_amdgcn_exp2f(cur[S * E + e]);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 2 * E, 0);
  if constexpr (S + 1 < DK) l2_step<DK, S + 1>(cur, l, c, a, b, cinit);
}

template <int DK>
__device__ __forceinline__ f32x16 exp_and_next_mfma_l2(const f32x16& cur, L2State& st, const bf16x8 (&a)[DK],
                                                       const bf16x8 (&b)[DK], const f32x16& cinit) {
  static_assert(DK <= 16, "one exp group per MFMA");
  float l = st.l;
  f32x16 c;
  l2_step<DK, 0>(cur, l, c, a, b, cinit);
  st.l = l;
  return c;
}

// Halves share M2, so the winner is decided on mr directly (exact).  Partial of a key range, common
// to this kernel and corr_bf16_direct_kernel: (max logit in log2 units, reference R, l = sum 2^(s' - R), idx).
__device__ __forceinline__ void store_partial_l2(const L2State& st, int q, int P, int split, float* pm,
                                                 float* pM2, float* pl, int32_t* pbi) {
  const float omr = __shfl_xor(st.mr, 32, 64);
  const float ol = __shfl_xor(st.l, 32, 64);
  const int obi = __shfl_xor(st.bi, 32, 64);
  const bool other = better(omr, obi, st.mr, st.bi);
  if ((threadIdx.x & 63) < 32 && q < P) {
    const size_t off = (size_t)split * P + q;
    pm[off] = st.M2 + (other ? omr : st.mr);
    pM2[off] = st.M2;
    pl[off] = st.l + ol;
    pbi[off] = other ? obi : st.bi;
  }
}

// Merge key ranges in f64: winner by the maximum (ascending ranges, strict >: ties keep the lower
// key), l rescaled to the largest reference;
//   logp = -(ln l + (Rf - mw) ln2),  lse = mw ln2 - logp   (natural-log units).
__global__ void corr_finalize_l2_kernel(int P, int nsplit, const float* __restrict__ pm,
                                        const float* __restrict__ pM2, const float* __restrict__ pl,
                                        const int32_t* __restrict__ pbi, int32_t* __restrict__ idx,
                                        float* __restrict__ logp, float* __restrict__ lse) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P) return;
  double mw = pm[q], Rf = pM2[q], l = pl[q];
  int bi = pbi[q];
  for (int s = 1; s < nsplit; ++s) {
    const size_t o = (size_t)s * P + q;
    const double m = pm[o], R = pM2[o], ls = pl[o];
    if (m > mw) { mw = m; bi = pbi[o]; }
    if (R > Rf) { l = l * exp2(Rf - R) + ls; Rf = R; }
    else l += ls * exp2(R - Rf);
  }
  const double ln2 = 0.6931471805599453094;
  // <= 0 like log_softmax (the sum contains the maximum's own term); rounding of the shifted
  // logits can leave it a few 1e-6 above
  const double lp = fmin(0.0, -(log(l) + (Rf - mw) * ln2));
  idx[q] = bi;
  if (logp) logp[q] = (float)lp;
  if (lse) lse[q] = (float)(mw * ln2 - lp);
}

// ------------------------------------------------------------------------------------ bf16
// Keys are staged through LDS once per workgroup (coalesced 16-byte global loads, XOR-swizzled
// image, ds_read_b128 in MFMA operand layout) and shared by the workgroup's four waves.
// (Tried and rejected, measured on MI355X: every wave streaming its own A fragments straight from
// global memory — no LDS, no barrier — is bound by the CU's vector L1: 245 ns per tile against
// 202 ns here, with the epilogue entirely hidden behind the loads.)
#ifndef ISR_BF16_WAVES
#define ISR_BF16_WAVES 1
#endif
template <int DK, int QB, bool LOG2 = false>  // D = 16 * DK, DK in {1, 2, 4, 8}; QB 32-query blocks per wave
__global__ __launch_bounds__(kThreads, (DK <= 4 && !LOG2) ? ISR_BF16_WAVES : 1) void corr_bf16_kernel(
    const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K, int P, int N, int ldq, int ldk,
    int split_len, float* __restrict__ pm, float* __restrict__ pM2, float* __restrict__ pl,
    int32_t* __restrict__ pbi, const int32_t* __restrict__ flags) {
  // flags != nullptr: this kernel runs as the fallback of corr_bf16_direct_kernel — only the
  // workgroups that kernel flagged do any work.
  if (flags && flags[blockIdx.y * gridDim.x + blockIdx.x] == 0) return;
  constexpr int NCH = 2 * DK;                         // 16-byte chunks per key row
  constexpr int RPB = (NCH >= 16) ? 1 : 16 / NCH;     // key rows per 256-byte LDS bank row
  constexpr int CHUNKS = kTK * NCH;                   // chunks per stage
  constexpr int NLD = (CHUNKS + kThreads - 1) / kThreads;
  __shared__ uint4 lds[2][CHUNKS];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.y;
  const int q0 = (blockIdx.x * kWaves + wave) * (QB * 32);

  bf16x8 bq[QB][DK];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int row = q0 + qb * 32 + r;
    row = row < P ? row : P - 1;
    const uint16_t* src = Q + (size_t)row * ldq + 8 * h;
#pragma unroll
    for (int s = 0; s < DK; ++s) bq[qb][s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  }

  LaneState st[QB];
  L2State s2[QB];
  f32x16 cinit[QB];                     // LOG2: -M2 of the lane's query, the C operand of each chain
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    st[qb].m = -__builtin_inff(); st[qb].M2 = kNoM2; st[qb].l = 0.f; st[qb].bi = 0;
    s2[qb].mr = -__builtin_inff(); s2[qb].M2 = 0.f; s2[qb].l = 0.f; s2[qb].bi = 0;
    cinit[qb] = splat16(0.f);
  }

  const int k0 = split * split_len;
  const int k1 = min(N, k0 + split_len);
  const int nstage = (k1 - k0 + kTK - 1) / kTK;

  uint4 stg[NLD];
  auto gload = [&](int stage) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int ci = tid + i * kThreads;
      const int row = ci / NCH, c = ci % NCH;
      const int key = k0 + stage * kTK + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ci < CHUNKS && key < k1)
        v = *reinterpret_cast<const uint4*>(K + (size_t)key * ldk + 8 * c);
      stg[i] = v;
    }
  };
  auto lwrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int ci = tid + i * kThreads;
      const int row = ci / NCH, c = ci % NCH;
      if (ci < CHUNKS) lds[buf][row * NCH + (c ^ ((row / RPB) & (NCH - 1)))] = stg[i];
    }
  };
  // A fragments of key sub-tile `sub` of LDS buffer `buf`
  bf16x8 a[DK];
  auto load_a = [&](int buf, int sub) {
    const int row = sub * 32 + r;
    const int sw = (row / RPB) & (NCH - 1);
#pragma unroll
    for (int s = 0; s < DK; ++s) {
      const uint4 v = lds[buf][row * NCH + ((2 * s + h) ^ sw)];
      a[s] = *reinterpret_cast<const bf16x8*>(&v);
    }
  };

  // Software pipeline over the stage's work items w = (sub, qb): while item w's logits go through
  // the softmax epilogue, the MFMAs of item w + 1 run.  The stage barrier sits before the LAST
  // item's epilogue (all LDS reads of the stage are done by then), so the first MFMAs of the next
  // stage overlap that epilogue too.  Two named accumulators alternate by the (static) item
  // parity: no register copies between items.
  constexpr int NSUB = kTK / 32, NW = NSUB * QB;
  static_assert(NW % 2 == 0, "items per stage must be even for the accumulator ping-pong");
  gload(0);
  lwrite(0);
  __syncthreads();
  load_a(0, 0);
  f32x16 acc[2];
  acc[0] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < DK; ++s) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bq[0][s], acc[0], 0, 0, 0);

  // FULL stages (every key of the stage exists) run without any per-item condition, so the two
  // accumulators never meet in a phi and the compiler keeps them in place (a v_mov between the
  // MFMA chain and the epilogue would stall the in-order wave until the chain retires).
  // The last, partial stage takes the guarded body.
  auto stage_body = [&](int stage, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int buf = stage & 1;
    const bool has_next = stage + 1 < nstage;
    if (has_next) gload(stage + 1);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int sub = w / QB, qb = w % QB;
      const int qbn = (w + 1) % QB;
      if (w + 1 == NW) {
        if (has_next) lwrite(buf ^ 1);
        __syncthreads();
        if (has_next) load_a(buf ^ 1, 0);
      } else if (qbn == 0) {
        load_a(buf, sub + 1);
      }
      const int kb = k0 + stage * kTK + sub * 32;
      const int krow0 = kb + 4 * h;
      if (FULL || kb < k1) {  // block-uniform
        if (!FULL && kb + 32 > k1) mask_tail(acc[w & 1], krow0, k1);
        if (LOG2) {
          update_max_l2(acc[w & 1], krow0, s2[qb], cinit[qb]);
          acc[(w + 1) & 1] = exp_and_next_mfma_l2<DK>(acc[w & 1], s2[qb], a, bq[qbn], cinit[qbn]);
        } else {
          update_max(acc[w & 1], krow0, st[qb]);
          acc[(w + 1) & 1] = exp_and_next_mfma<DK>(acc[w & 1], st[qb], a, bq[qbn]);
        }
      }
    }
  };
  const int nfull = (k1 - k0) / kTK;  // stages whose kTK keys all exist
  for (int stage = 0; stage < nfull; ++stage) stage_body(stage, std::true_type{});
  if (nfull < nstage) stage_body(nfull, std::false_type{});

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    if (LOG2) store_partial_l2(s2[qb], q0 + qb * 32 + r, P, split, pm, pM2, pl, pbi);
    else store_partial(st[qb], q0 + qb * 32 + r, P, split, pm, pM2, pl, pbi);
  }
}

#include "corr_direct.hpp"   // corr_bf16_direct_kernel: the VALU-minimal bf16 loop (log2 and natural units)

// ------------------------------------------------------------------------------------- f32
template <int DP>  // padded D (multiple of 2): k-steps = DP / 2
__global__ __launch_bounds__(kThreads) void corr_f32_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, int P, int N, int D, int ldq, int ldk,
    int split_len, float* __restrict__ pm, float* __restrict__ pM2, float* __restrict__ pl,
    int32_t* __restrict__ pbi) {
  constexpr int KS = DP / 2;
  constexpr int LD = DP + 1;  // odd dword stride: conflict-free ds_read_b32 down a column
  __shared__ float lds[2][kTK * LD];

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
#pragma unroll
  for (int qb = 0; qb < kQB; ++qb) {
    st[qb].m = -__builtin_inff(); st[qb].M2 = kNoM2; st[qb].l = 0.f; st[qb].bi = 0;
  }

  const int k0 = split * split_len;
  const int k1 = min(N, k0 + split_len);
  const int nstage = (k1 - k0 + kTK - 1) / kTK;
  constexpr int NEL = (kTK * DP + kThreads - 1) / kThreads;
  float stg[NEL];
  auto gload = [&](int stage) {
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
      const int e = tid + i * kThreads;
      const int row = e / DP, col = e % DP;
      const int key = k0 + stage * kTK + row;
      stg[i] = (e < kTK * DP && key < k1 && col < D) ? K[(size_t)key * ldk + col] : 0.f;
    }
  };
  auto lwrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
      const int e = tid + i * kThreads;
      if (e < kTK * DP) lds[buf][(e / DP) * LD + (e % DP)] = stg[i];
    }
  };

  if (nstage > 0) { gload(0); lwrite(0); }
  __syncthreads();
  for (int stage = 0; stage < nstage; ++stage) {
    const int buf = stage & 1;
    if (stage + 1 < nstage) gload(stage + 1);
#pragma unroll
    for (int sub = 0; sub < kTK / 32; ++sub) {
      const int kb = k0 + stage * kTK + sub * 32;
      if (kb < k1) {
        const float* arow = &lds[buf][(sub * 32 + r) * LD + h];
        f32x16 acc[kQB];
#pragma unroll
        for (int qb = 0; qb < kQB; ++qb)
          acc[qb] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
  for (int qb = 0; qb < kQB; ++qb) store_partial(st[qb], q0 + qb * 32 + r, P, split, pm, pM2, pl, pbi);
}

// Merge the key-range splits (ascending, so equal maxima keep the lowest key) and write outputs.
// logp = -ln sum_n e^(s_n - m) is formed without the m - lse cancellation, in f64:
//   sum_n e^(s_n - m) = l * 2^M2 / e^m   ->   logp = -(ln l + M2 ln2 - m)
__global__ void corr_finalize_kernel(int P, int nsplit, const float* __restrict__ pm,
                                     const float* __restrict__ pM2, const float* __restrict__ pl,
                                     const int32_t* __restrict__ pbi, int32_t* __restrict__ idx,
                                     float* __restrict__ logp, float* __restrict__ lse) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P) return;
  LaneState a{pm[q], pM2[q], pl[q], pbi[q]};
  for (int s = 1; s < nsplit; ++s) {
    const size_t o = (size_t)s * P + q;
    const LaneState b{pm[o], pM2[o], pl[o], pbi[o]};
    merge_state(a, b);
  }
  const double ln2 = 0.6931471805599453094;
  const double lp = fmin(0.0, -(log((double)a.l) + ((double)a.M2 * ln2 - (double)a.m)));   // <= 0 like log_softmax
  idx[q] = a.bi;
  if (logp) logp[q] = (float)lp;
  if (lse) lse[q] = (float)((double)a.m - lp);
}

// upper bound on nsplit (also sizes the workspace, which must not depend on the device)
constexpr int kMaxSplit = 64;

struct CorrPlan {
  int qblocks, nsplit, split_len;
};

// Work units = (query block, key range).  Splitting the key range costs VALU work: every split
// restarts its running maximum, and the arg-max update path (taken whenever ANY of a wave's 64
// lanes improves, i.e. for ~64(1 + ln(tiles/64)) of a split's tiles) is the expensive part of the
// epilogue — measured: 23 splits raised VALU instructions per tile from ~75 to 113.  So the range
// is split only when the query blocks alone cannot fill the machine (small P, e.g. the
// reference's 75x75 crops), and then just enough to give every resident slot one unit.
// Launch tails of large-P calls are hidden by the caller pipelining images over streams.
CorrPlan make_plan(int P, int N, int slots, int q_per_block) {
  CorrPlan p;
  p.qblocks = (P + q_per_block - 1) / q_per_block;
  const int max_split = (N + kTK - 1) / kTK;
  int ns = 1;
  if (p.qblocks < slots) ns = (slots + p.qblocks - 1) / p.qblocks;
  if (ns > max_split) ns = max_split;
  if (ns > kMaxSplit) ns = kMaxSplit;
  const int stages = (max_split + ns - 1) / ns;
  p.split_len = stages * kTK;
  p.nsplit = (N + p.split_len - 1) / p.split_len;
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
  // cached per (kernel family, padded D): the occupancy query costs tens of microseconds
  static int cache[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  int v = 0;
  if (dtype != ISR_DTYPE_F32) v = (D <= 16) ? 0 : (D <= 32) ? 1 : (D <= 64) ? 2 : 3;
  else v = (D <= 8) ? 0 : (D <= 16) ? 1 : (D <= 32) ? 2 : 3;
  int& c = cache[dtype == ISR_DTYPE_F32 ? 1 : dtype == ISR_DTYPE_BF16_LOG2 ? 2 : 0][v];
  if (c == 0) {
    if (dtype == ISR_DTYPE_BF16_LOG2) {
      c = v == 0 ? resident_slots(corr_bf16_direct_kernel<1, kQBbf16, false>)
        : v == 1 ? resident_slots(corr_bf16_direct_kernel<2, kQBbf16, false>)
        : v == 2 ? resident_slots(corr_bf16_direct_kernel<4, kQBbf16, false>)
                 : resident_slots(corr_bf16_direct_kernel<8, 2, false>);
    } else if (dtype != ISR_DTYPE_F32) {
      c = v == 0 ? resident_slots(corr_bf16_direct_kernel<1, kQBbf16, true>)
        : v == 1 ? resident_slots(corr_bf16_direct_kernel<2, kQBbf16, true>)
        : v == 2 ? resident_slots(corr_bf16_direct_kernel<4, kQBbf16, true>)
                 : resident_slots(corr_bf16_direct_kernel<8, 2, true>);
    } else {
      c = v == 0 ? resident_slots(corr_f32_kernel<8>) : v == 1 ? resident_slots(corr_f32_kernel<16>)
        : v == 2 ? resident_slots(corr_f32_kernel<32>) : resident_slots(corr_f32_kernel<64>);
    }
  }
  return c;
}

// upper bound on nsplit for the workspace query (which must not depend on the device)
}  // namespace

extern "C" size_t isr_corr_argmax_workspace_bytes(int P, int N, int D, int dtype) {
  (void)D; (void)dtype;
  if (P <= 0 || N <= 0) return 0;
  const int max_split = (N + kTK - 1) / kTK;
  const int ns = max_split < kMaxSplit ? max_split : kMaxSplit;
  const size_t nflags = (size_t)ns * ((size_t)P / (kWaves * 32) + 2);   // one per (query block, key range)
  return 4 * isr::align_up((size_t)ns * P * 4, 256) + isr::align_up(nflags * 4, 256) + 256;
}

extern "C" int isr_corr_argmax(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk,
                               int dtype, int32_t* idx, float* logp, float* lse, void* ws,
                               size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(Q && K && idx, "isr_corr_argmax: null pointer");
  ISR_REQUIRE(P > 0 && N > 0 && D > 0, "isr_corr_argmax: P=%d N=%d D=%d must be positive", P, N, D);
  ISR_REQUIRE(ldq >= D && ldk >= D, "isr_corr_argmax: ldq=%d ldk=%d < D=%d", ldq, ldk, D);
  if (!ws || ws_bytes < isr_corr_argmax_workspace_bytes(P, N, D, dtype)) {
    isr::set_error("isr_corr_argmax: workspace %zu < %zu", ws_bytes,
                   isr_corr_argmax_workspace_bytes(P, N, D, dtype));
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  const int qb_wave = (dtype != ISR_DTYPE_F32 && D <= 64) ? kQBbf16 : 2;
  const CorrPlan p = make_plan(P, N, slots_for(dtype, D), kWaves * qb_wave * 32);
  isr::Workspace w(ws, ws_bytes);
  float* pm = w.take<float>((size_t)p.nsplit * P);
  float* pM2 = w.take<float>((size_t)p.nsplit * P);
  float* pl = w.take<float>((size_t)p.nsplit * P);
  int32_t* pbi = w.take<int32_t>((size_t)p.nsplit * P);
  int32_t* flags = w.take<int32_t>((size_t)p.nsplit * p.qblocks);
  const dim3 grid(p.qblocks, p.nsplit);

  if (dtype == ISR_DTYPE_BF16 || dtype == ISR_DTYPE_BF16_LOG2) {
    ISR_REQUIRE(D == 16 || D == 32 || D == 64 || D == 128,
                "isr_corr_argmax(bf16): D=%d must be 16, 32, 64 or 128 (zero-pad the columns)", D);
    ISR_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)K % 16 == 0),
                "isr_corr_argmax(bf16): rows must be 16-byte aligned (ldq=%d ldk=%d)", ldq, ldk);
    ISR_REQUIRE((long long)p.split_len * ldk * 2 < (1ll << 31),
                "isr_corr_argmax(bf16): a key range of %d rows x ldk=%d exceeds the 2 GiB buffer window",
                p.split_len, ldk);
    const uint16_t* q = static_cast<const uint16_t*>(Q);
    const uint16_t* k = static_cast<const uint16_t*>(K);
#define ISR_LAUNCH_BF16(DKv, QBv)                                                                         \
  do {                                                                                                    \
    if (dtype == ISR_DTYPE_BF16_LOG2) {                                                                   \
      corr_bf16_direct_kernel<DKv, QBv, false><<<grid, kThreads, 0, stream>>>(                            \
          q, k, P, N, ldq, ldk, p.split_len, pm, pM2, pl, pbi, flags);                                    \
      corr_bf16_kernel<DKv, QBv, true><<<grid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.split_len, \
                                                                      pm, pM2, pl, pbi, flags);            \
    } else {                                                                                              \
      corr_bf16_direct_kernel<DKv, QBv, true><<<grid, kThreads, 0, stream>>>(                             \
          q, k, P, N, ldq, ldk, p.split_len, pm, pM2, pl, pbi, flags);                                    \
      corr_bf16_kernel<DKv, QBv, false><<<grid, kThreads, 0, stream>>>(q, k, P, N, ldq, ldk, p.split_len, \
                                                                       pm, pM2, pl, pbi, flags);           \
    }                                                                                                     \
  } while (0)
    switch (D) {
      case 16: ISR_LAUNCH_BF16(1, kQBbf16); break;
      case 32: ISR_LAUNCH_BF16(2, kQBbf16); break;
      case 64: ISR_LAUNCH_BF16(4, kQBbf16); break;
      default: ISR_LAUNCH_BF16(8, 2); break;
    }
#undef ISR_LAUNCH_BF16
  } else if (dtype == ISR_DTYPE_F32) {
    ISR_REQUIRE(D <= 64, "isr_corr_argmax(f32): D=%d > 64", D);
    const float* q = static_cast<const float*>(Q);
    const float* k = static_cast<const float*>(K);
    if (D <= 8) corr_f32_kernel<8><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.split_len, pm, pM2, pl, pbi);
    else if (D <= 16) corr_f32_kernel<16><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.split_len, pm, pM2, pl, pbi);
    else if (D <= 32) corr_f32_kernel<32><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.split_len, pm, pM2, pl, pbi);
    else corr_f32_kernel<64><<<grid, kThreads, 0, stream>>>(q, k, P, N, D, ldq, ldk, p.split_len, pm, pM2, pl, pbi);
  } else {
    isr::set_error("isr_corr_argmax: dtype %d", dtype);
    return ISR_ERR_ARG;
  }
  ISR_CHECK_LAUNCH("corr kernel");
  if (dtype == ISR_DTYPE_BF16_LOG2)
    corr_finalize_l2_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, p.nsplit, pm, pM2, pl, pbi, idx, logp, lse);
  else
    corr_finalize_kernel<<<(P + 255) / 256, 256, 0, stream>>>(P, p.nsplit, pm, pM2, pl, pbi, idx, logp, lse);
  ISR_CHECK_LAUNCH("corr_finalize_kernel");
  return ISR_OK;
}
