// corr_direct.hpp — corr_bf16_direct_kernel, the fast path of K1 for bf16 descriptors.  Included by
// corr_argmax.hip inside its anonymous namespace after the shared pieces (vector types, kThreads /
// kWaves / kTK / kLog2e, max3, splat16, mask_tail, better); corr_bf16_kernel in that file is its
// flagged per-workgroup fallback and both write the same partials.
#pragma once

// -------------------------------------------------------------- bf16, log2 domain, direct sums
// The fast path of ISR_DTYPE_BF16_LOG2.  The loop above is bound by VALU issue (81 % busy, PMC), so
// this one carries the fewest per-element instructions the math allows: with logits already in log2
// units, l = sum 2^(s' - S) needs exp2 + add per element once the reference S is WAVE-uniform: it
// rides in the C operand of the first MFMA of every chain (one 16-register tile shared by the wave's
// two query blocks), so the accumulator already holds s' - S.  S is set from the wave's first tile
// so that the largest logit sits at +kAnchor, and is bumped
// (checked once per 128-key stage, taken a handful of times per launch at most) when a lane's
// running maximum passes +kBump: l is rescaled by an exact power of two.  Every term that matters
// relative to 2^m is a normal f32 as long as the lane's maximum ends above kLow (relative to the
// wave's reference), and an overflow (a jump of ~100 log2 units within one stage) leaves l = inf.
// A workgroup with such a lane — maxima of neighbouring pixels more than ~120 log2 units apart, or
// that jump — raises its flag, and the per-query-reference kernel above (launched right behind,
// returning at once for unflagged workgroups) redoes it: results never depend on the range
// assumption.
// Arg-max: the loop records only (m, first tile that reached m) — two instructions, no branch; the
// row inside the winning tile is found after the loop by running that one tile through the MFMA
// chain again with C = 0: the decision inside the tile and the reported maximum are unshifted
// logits, so equal keys compare equal whatever wave or key range saw them (lowest key on ties as
// everywhere; only the choice BETWEEN tiles is made on shifted values).
constexpr float kAnchor = 24.f;    // log2 units: where the wave's largest logit is put
constexpr float kBump = 64.f;      // a running maximum above this moves the reference
constexpr float kLow = -100.f;     // below: terms of the sum may have been flushed -> fallback

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

struct DirectState {
  float m;   // running max of s' - S (log2 units)
  float l;   // sum 2^(s' - S)
  int tb;    // first key of the first 32-key tile whose maximum reached m
};

// max(m, the 16 logits of the tile) in eight v_max3_f32
__device__ __forceinline__ float tile_max_with(const f32x16& acc, float m) {
  const float x0 = max3(acc[0], acc[1], acc[2]), x1 = max3(acc[3], acc[4], acc[5]),
              x2 = max3(acc[6], acc[7], acc[8]), x3 = max3(acc[9], acc[10], acc[11]),
              x4 = max3(acc[12], acc[13], acc[14]);
  return max3(max3(x0, x1, x2), max3(x3, x4, acc[15]), m);
}

// NAT = false: log2-unit logits (ISR_DTYPE_BF16_LOG2), reference in the MFMA C operand.
// NAT = true:  natural-unit logits (ISR_DTYPE_BF16): the accumulator stays raw (C = 0) and the
//   wave-uniform integer reference S (log2 units) enters through the one-rounding
//   exp2(fma(s, log2 e, -S)) — one more VALU instruction per element than the log2 path, still
//   without the per-query reference, its rescale branch and the in-loop arg-max update of
//   corr_bf16_kernel.  The running maximum is then the raw logit itself.
template <int DK, int QB, bool NAT>
__global__ __launch_bounds__(kThreads, DK <= 4 ? 3 : 2) void corr_bf16_direct_kernel(
    const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K, int P, int N, int ldq, int ldk,
    int split_len, float* __restrict__ pm, float* __restrict__ pM2, float* __restrict__ pl,
    int32_t* __restrict__ pbi, int32_t* __restrict__ flags) {
  constexpr int NCH = 2 * DK;
  constexpr int RPB = (NCH >= 16) ? 1 : 16 / NCH;
  constexpr int CHUNKS = kTK * NCH;
  constexpr int NLD = CHUNKS / kThreads;
  static_assert(CHUNKS % kThreads == 0, "every thread stages the same number of chunks");
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
  DirectState st[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { st[qb].m = -__builtin_inff(); st[qb].l = 0.f; st[qb].tb = 0; }

  const int k0 = split * split_len;
  const int k1 = min(N, k0 + split_len);
  const int nstage = (k1 - k0 + kTK - 1) / kTK;
  const int nfull = (k1 - k0) / kTK;                  // stages whose kTK keys all exist

  // Key rows come through a raw buffer descriptor over this key range: one 32-bit offset per load,
  // and rows beyond the range read as zero in hardware (no predicates in the loop).
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(K + (size_t)k0 * ldk), 0, (k1 - k0) * ldk * 2, 0x00020000);
  int koff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int ci = tid + i * kThreads;
    koff[i] = ((ci / NCH) * ldk + 8 * (ci % NCH)) * 2;
  }
  uint4 stg[NLD];
  auto gload = [&](int stage) {
    const int so = stage * kTK * ldk * 2;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(krs, koff[i] + so, 0, 0);
      stg[i] = *reinterpret_cast<const uint4*>(&v);
    }
  };
  auto lwrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int ci = tid + i * kThreads;
      const int row = ci / NCH, c = ci % NCH;
      lds[buf][row * NCH + (c ^ ((row / RPB) & (NCH - 1)))] = stg[i];
    }
  };
  // A fragments of key sub-tile `sub` of LDS buffer `buf`.  One register set: a fragment's ds_read
  // for the NEXT sub-tile is issued right behind the last MFMA that reads the current one, i.e. one
  // whole item (~300 cycles) before the MFMA that needs it.
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

  // Same software pipeline as corr_bf16_kernel: item w's epilogue runs under item w + 1's MFMAs.
  // The stage barrier sits two items before the stage's end (every LDS read of the stage has been
  // issued by then), so the next stage's first fragments are a full item ahead too.
  constexpr int NSUB = kTK / 32, NW = NSUB * QB;
  static_assert(QB == 2 && NSUB % 2 == 0, "item schedule below is written for two query blocks per wave");
  gload(0);
  lwrite(0);
  __syncthreads();
  load_a(0, 0);
  f32x16 acc[2];
  acc[0] = splat16(0.f);
#pragma unroll
  for (int s = 0; s < DK; ++s) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bq[0][s], acc[0], 0, 0, 0);

  // reference from the first tile (rows >= k1 of a short range are zero rows: logit 0, harmless here)
  constexpr float kUnit = NAT ? kLog2e : 1.0f;         // log2 units per logit unit
  float S = ceilf(wave_max(tile_max_with(acc[0], -__builtin_inff())) * kUnit) - kAnchor;
  S = fminf(fmaxf(S, -3.0e38f), 3.0e38f);
  f32x16 cS = splat16(NAT ? 0.f : -S);
  if (!NAT) {
    asm volatile("" : "+v"(cS));        // one resident tile, not sixteen moves per chain
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] -= S;
  }
  float nS = -S;                        // NAT: the addend of the exp2 argument
  float bump_at = NAT ? (S + kBump) * 0.6931471805599453f : kBump;   // running maximum that moves S

  auto stage_body = [&](int stage, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int buf = stage & 1;
    const bool has_next = stage + 1 < nstage;
    if (has_next) gload(stage + 1);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int sub = w / QB, qb = w % QB;
      const int qbn = (w + 1) % QB;
      if (w == NW - 2) {                                  // block-uniform
        if (has_next) lwrite(buf ^ 1);
#ifndef ISR_ABL_NOBARRIER  // timing-only ablation: what the stage barrier costs (results are wrong without it)
        __syncthreads();
#endif
      }
      const int kb = k0 + stage * kTK + sub * 32;
      if (FULL || kb < k1) {  // block-uniform
        if (!FULL && kb + 32 > k1) mask_tail(acc[w & 1], kb + 4 * h, k1);
        const f32x16& cur = acc[w & 1];
        f32x16& nxt = acc[(w + 1) & 1];
        if (NAT) nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[qbn][0], splat16(0.f), 0, 0, 0);
        else nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[qbn][0], cS, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < DK; ++s) nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bq[qbn][s], nxt, 0, 0, 0);
        if (qb == 0) {                                    // the chain above was the fragments' last reader
          if (w == NW - 2) { if (has_next) load_a(buf ^ 1, 0); }
          else load_a(buf, sub + 1);
        }
#ifdef ISR_ABL_DNOMAX   // timing-only ablations (tools/ablate_direct.sh): keep the accumulator live
        asm volatile("" :: "v"(cur[0]), "v"(cur[15]));
        st[qb].m = 0.f;      // inside the range check: no fallback launch work
#else
        const float mn = tile_max_with(cur, st[qb].m);
        st[qb].tb = (mn > st[qb].m) ? kb : st[qb].tb;  // strict: the first tile to reach m keeps it
        st[qb].m = mn;
#endif
        float l = st[qb].l;
#ifdef ISR_ABL_DNOEXP
        asm volatile("" :: "v"(cur[1]), "v"(cur[14]));
        l = 1.f;
#else
#pragma unroll
        for (int i = 0; i < 16; ++i)
          l += __builtin_amdgcn_exp2f(NAT ? __builtin_fmaf(cur[i], kLog2e, nS) : cur[i]);
#endif
        st[qb].l = l;
        // issue order: the item's 42 VALU instructions (8 max3, 2 record, 16 exp2, 16 add; NAT: 16
        // fma more) spread evenly behind the DK MFMAs of the next item.  The empty asm ties the
        // item's results to a fixed point of the instruction stream: a stage is one basic block, and
        // without it instruction selection sinks all eight epilogues below all eight MFMA chains
        // (eight tiles live, 243 VGPRs, nothing overlapped).
        constexpr int G = ((NAT ? 58 : 42) + DK - 1) / DK;
#pragma unroll
        for (int s = 0; s < DK; ++s) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (qb == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // this fragment's next ds_read
          __builtin_amdgcn_sched_group_barrier(0x002, G, 0);
        }
        asm volatile("" : "+v"(nxt), "+v"(st[qb].l), "+v"(st[qb].m), "+v"(st[qb].tb));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // reference upkeep, once per stage (wave-uniform, rare): acc[0] is the tile in flight
    const float mm = fmaxf(st[0].m, st[1].m);
    if (__any(mm > bump_at)) {
      const float top = ceilf(wave_max(mm) * kUnit);      // log2 units; NAT: absolute, else relative to S
      const float d = fminf(NAT ? top - kAnchor - S : top - kAnchor, 3.0e38f);
      S += d;
      const float sc = __builtin_amdgcn_exp2f(-d);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) st[qb].l *= sc;     // an overflowed l stays inf
      if (NAT) {
        nS = -S;
        bump_at = (S + kBump) * 0.6931471805599453f;
      } else {
        cS = splat16(-S);
        asm volatile("" : "+v"(cS));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) st[qb].m -= d;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][i] -= d;
      }
    }
  };
  for (int stage = 0; stage < nfull; ++stage) stage_body(stage, std::true_type{});
  if (nfull < nstage) stage_body(nfull, std::false_type{});

  // ---- range check: any lane outside the direct-sum range sends the workgroup to the fallback
  bool bad = false;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
    bad |= !((NAT ? __builtin_fmaf(st[qb].m, kLog2e, -S) : st[qb].m) >= kLow && st[qb].l <= 3.0e38f);
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (tid == 0) flags[blockIdx.y * gridDim.x + blockIdx.x] = any_bad;
  if (any_bad) return;   // block-uniform; the fallback kernel writes this workgroup's partials

  // ---- row recovery: one MFMA chain per distinct winning tile of the wave's queries
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    // the two lanes (h = 0, 1) of a query agree on (M, T): the maximum and the lowest tile reaching it
    const float mo = __shfl_xor(st[qb].m, 32, 64);
    const int tbo = __shfl_xor(st[qb].tb, 32, 64);
    const float M = fmaxf(st[qb].m, mo);
    const int T = (st[qb].m == M) ? ((mo == M) ? min(st[qb].tb, tbo) : st[qb].tb) : tbo;
    int cand = T;
    float cmax = -__builtin_inff();
    unsigned long long todo = __ballot(true);
    auto fetch = [&](int kb, bf16x8 (&dst)[DK]) {
      int row = kb + r;
      row = row < N ? row : N - 1;
      const uint16_t* src = K + (size_t)row * ldk + 8 * h;
#pragma unroll
      for (int s = 0; s < DK; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
    };
    bf16x8 a0[DK], a1[DK];
    int kb_cur = __shfl(T, __ffsll(todo) - 1, 64);
    fetch(kb_cur, a0);
    while (true) {                                        // wave-uniform trip count (<= 32)
      todo &= ~__ballot(T == kb_cur);
      const bool more = todo != 0ull;
      int kb_nxt = kb_cur;
      if (more) {
        kb_nxt = __shfl(T, __ffsll(todo) - 1, 64);
        fetch(kb_nxt, a1);                                // in flight under this tile's MFMAs
      }
      f32x16 c = splat16(0.f);
#pragma unroll
      for (int s = 0; s < DK; ++s) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[s], bq[qb][s], c, 0, 0, 0);
      if (kb_cur + 32 > k1) mask_tail(c, kb_cur + 4 * h, k1);
      if (T == kb_cur) {
        cmax = tile_max_with(c, -__builtin_inff());
        int rr = 15;
#pragma unroll
        for (int i = 14; i >= 0; --i) rr = (c[i] == cmax) ? i : rr;   // lowest register = lowest key
        cand = kb_cur + 4 * h + (rr & 3) + 8 * (rr >> 2);
      }
      if (!more) break;
#pragma unroll
      for (int s = 0; s < DK; ++s) a0[s] = a1[s];
      kb_cur = kb_nxt;
    }
    const int co = __shfl_xor(cand, 32, 64);
    const float cmo = __shfl_xor(cmax, 32, 64);
    if (better(cmo, co, cmax, cand)) { cand = co; cmax = cmo; }
    const float lo = __shfl_xor(st[qb].l, 32, 64);
    const int q = q0 + qb * 32 + r;
    if (h == 0 && q < P) {
      const size_t off = (size_t)split * P + q;
      pm[off] = cmax;             // the unshifted logit: equal keys compare equal across waves and ranges
      pM2[off] = S;
      pl[off] = st[qb].l + lo;
      pbi[off] = cand;
    }
  }
}

