// corr_direct.hpp — corr_bf16_direct_kernel, the fast path of K1 for bf16 descriptors.  Included by
// corr_argmax.hip inside its anonymous namespace after the shared pieces (vector types, kThreads /
// kWaves / kTK / kChunk / kLog2e, CorrWs, max3, splat16, mask_tail, better); corr_bf16_kernel in that
// file is its per-query fallback.
#pragma once

// -------------------------------------------------------------- bf16, direct sums, fixed reference
// The loop is bound by VALU issue (PMC: ~80 % busy), so it carries the fewest per-element
// instructions the math allows: with logits in log2 units, l = sum 2^s' is exp2 + add per element —
// no reference subtraction at all.  The reference of the log-sum-exp is the CONSTANT 0, so what a
// query's sums are made of depends on the query and the keys only, never on its wave-mates or on
// the launch (round 1 anchored the reference on the wave's first tile: results then moved by ~1e-5
// with the launch shape, enough to flip the strict `>` of the top-80 % cut, inference.py:282-290).
// Canonical order of the sum: keys are cut into CHUNKS of kChunk = 4096 (a constant); inside a
// chunk each of the query's two lanes adds its 16 rows per 32-key tile in register order, tile
// after tile, in f32; the two lane sums are added; the chunk sums go to memory and
// corr_finalize_kernel adds them in ascending order in f64.  A workgroup may own any number of
// whole chunks (the plan splits the key range only when the query blocks alone cannot fill the
// chip): the grouping of chunks into workgroups never enters the arithmetic.
// Range: 2^s' is a normal f32 for -126 <= s' < 128.  A query whose maximum lies below kLow (terms
// that matter relative to the maximum could flush) or whose chunk sum overflows is marked bad and
// redone by the per-query-reference kernel (corr_bf16_kernel, same canonical chunks) — per QUERY,
// so a good query's result does not depend on its neighbours being good.
// Arg-max: the loop records, per lane, the two largest tile maxima (v_med3 keeps the runner-up) and
// the first tile that reached the maximum; the row inside the winning tile is found after the loop
// by running that one tile through the MFMA chain again.  All comparisons are on raw MFMA outputs
// (C = 0): a given (query, key) pair yields the same f32 logit wherever it is computed.
// The runner-up bounds the top-2 margin; corr_finalize_kernel sends queries whose margin is inside
// the f32 accumulation error bound to corr_recheck_kernel, which decides them in exact arithmetic.
constexpr float kLow = -100.f;     // log2 units: a maximum below this sends the query to the fallback
// The range tests below are written `!(x >= kLow)` / `!(lc <= 3.0e38f)`.  This file is compiled with -fno-honor-nans
// (build.py), under which the compiler may read them as `x < kLow` / `lc > 3.0e38f`: that is all they have to catch — a
// maximum of -inf (no key), an overflowed sum (+inf).  A NaN cannot arise from FINITE descriptors (no inf - inf, no 0 x inf:
// products of finite bf16 / f16 values are finite in f32 and the sums' only non-finite value is +inf); finite descriptors
// are a stated precondition of K1, and nothing here relies on a NaN taking either branch.

struct DirectState {
  float m;   // largest tile maximum so far (raw logit)
  float m2;  // second largest tile maximum (with multiplicity)
  float l;   // sum 2^(s') over the current chunk
  int tb;    // first key of the first 32-key tile whose maximum reached m
  float thr; // screen (ISR_K1_SCREEN): 2^(query's maximum so far - dlt) — a tile whose sum of exponentials stays below it holds nothing near the maximum
  double L;  // the canonical f64 sum of the chunk sums so far (what corr_finalize_kernel forms from memory)
};

// max of the 16 logits of a tile: seven v_max3_f32 + one v_max_f32
__device__ __forceinline__ float tile_max(const f32x16& acc) {
  const float x0 = max3(acc[0], acc[1], acc[2]), x1 = max3(acc[3], acc[4], acc[5]),
              x2 = max3(acc[6], acc[7], acc[8]), x3 = max3(acc[9], acc[10], acc[11]),
              x4 = max3(acc[12], acc[13], acc[14]);
  return fmaxf(max3(x0, x1, x2), max3(x3, x4, acc[15]));
}

// NAT = false: log2-unit logits (ISR_DTYPE_BF16_LOG2): l += exp2(s').
// NAT = true:  natural-unit logits (ISR_DTYPE_BF16):   l += exp2(s * log2 e) — one multiply more per
//   element.  Maxima are raw logits in either case.
// DKU <= DK: the 16-wide blocks that can be non-zero (the split-f32 route stores 8 blocks of which the last two are zero
// padding: their MFMAs would add exact zeros, so they are not issued — same bits, 6 instead of 8 matrix instructions per tile).
// SP > 0: split rows (RowFrags / tile_chain in corr_argmax.hip): three planes of SP blocks per operand row, 6 SP matrix
// instructions per tile (3 SP with f16 planes, F16); DK = 3 SP, DKU unused.
// LSE: the caller wants the log-sum-exp only (pose_refine.py:56's denominator image, estimate_pose's row sums): no maxima
// are tracked (12 of the 44 VALU instructions of a tile), no row is recovered, nothing is rechecked.  What the maximum did for
// the RANGE of the sum — a query is redone with a per-query reference when its maximum is below kLow — is decided from the
// Cauchy-Schwarz bound instead: |s'| <= |q||k|_max < kLseBound puts every term inside [2^-99, 2^99] and the maximum above kLow;
// a query that fails the bound goes to the fallback as a bad one does.  The sums, and with them lse, are the full kernel's bits.
constexpr float kLseBound2 = 99.f * 99.f;
// ISR_K1_SCREEN (plain rows): the tile's 16 exponentials are summed on their own (ts) before they join the chunk sum, and
// ts >= max_i 2^(s_i) (1 - 16 u) SCREENS the tile for the maxima: with thr = 2^(M - dlt), M the largest logit the query's two
// lanes have seen and dlt > 2 eps + the roundings of exp2 / the sum / M - dlt (eps: corr_finish's margin bound), ts < thr
// proves that every logit of the tile lies more than 2 eps below M, hence below the final maximum: it is neither the
// arg-max nor a runner-up the margin test could care about (a logit within 2 eps of the winner always sits in a tile that
// took the slow path, where m / m2 / tb are kept exactly as before).  The 12 max-type instructions of a tile (of 44) then
// run only when SOME lane of the wave fails the screen (wave-uniform branch); the slow path also refreshes thr from the
// maximum of the query's two lanes (v_permlane32_swap).  The first tiles always take it (thr starts at 0).
#ifndef ISR_K1_SCREEN
#define ISR_K1_SCREEN 1
#endif
#ifndef ISR_K1_SCREEN_PLANES          // bit SP: plane rows of SP blocks per plane use the screen
#define ISR_K1_SCREEN_PLANES 0x106    // SP = 1, 2, 8
#endif
// SP = 8 (f16 planes of 128-wide rows): 24 + 24 + 24 fragments of a wave's two query blocks and of a key sub-tile need more
// than 256 registers — one wave per SIMD with the whole 512-register file, and two 48 KB stage buffers as dynamic LDS.
extern __shared__ uint4 corr_direct_dyn_lds[];
#ifndef ISR_K1_PLAIN_WAVES
#define ISR_K1_PLAIN_WAVES 3      // waves per SIMD the plain-row kernels (D <= 64) are compiled for
#endif
template <int DK, int QB, bool NAT, int DKU = DK, int SP = 0, bool F16 = false, bool LSE = false, int SKIP = 0>
__global__ __launch_bounds__(kThreads, (DK <= 4 && SP == 0) ? ISR_K1_PLAIN_WAVES : (DK <= 4 ? 3 : (SP >= 8 ? 1 : 2))) void corr_bf16_direct_kernel(
    const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K, int P, int N, int ldq, int ldk,
    int range_chunks, CorrWs ws, int32_t* __restrict__ idx_out, float* __restrict__ logp_out,
    float* __restrict__ lse_out) {
  // 16-byte chunks staged per key row: all 2 DK of them, or — DKU < DK, the split-f32 route's rows with trailing zero blocks —
  // only the 2 DKU that can be non-zero (12 for DKU = 6: a quarter less key traffic, which is what bounds that route: its
  // 256-byte key rows do not fit the L2, profiles/r03_estimate_pose_hbm_traffic.txt).
  using RF = RowFrags<DKU, SP, F16>;
  constexpr int NFR = RF::NFR, NMF = RF::NMF;
  static_assert(!F16 || SP != 0, "f16 operands exist as split planes only");
  // the margin test's bound: plain rows (D + 2) 2^-23 |q||k|; split rows split_deff / split_deff_f16 (+ the absolute term)
  constexpr int DEFF = SP ? (F16 ? split_deff_f16(SP) : split_deff(SP)) : 16 * DK;
  constexpr float EABS = (SP && F16) ? split_eabs(SP ? SP : 1) : 0.f;
  // the sum screen of the maxima: plain rows always; plane rows where it pays — SP = 1, 2 and 8 (D = 12: 3.30 -> 3.05 ms on the
  // crop batch, D = 32: 1.56 -> 1.50, D = 128: 6.7 -> 5.8 ms per image); at SP = 4 the 12 (24) matrix instructions of a tile
  // already hide the maxima and the branch behind the item costs more than they did (2.56 -> 2.77 ms, profiles/r04_k1_screen.txt)
  constexpr bool SCR = ISR_K1_SCREEN != 0 && (SP == 0 || ((ISR_K1_SCREEN_PLANES >> SP) & 1) != 0);
  if (gated_off(ws)) return;
  if constexpr (SKIP != 0) {                    // behind the screened route: only the query blocks its pass 1 handed over
    if (ws.hand && !ws.hand[blockIdx.x]) return;
  }
  static_assert(SP == 0 || DK == 3 * SP, "split rows: DK counts the 3 SP blocks of a row");
  constexpr int NCH = 2 * NFR;
  constexpr bool POW2 = (NCH & (NCH - 1)) == 0;
  constexpr int RPB = (NCH >= 16) ? 1 : 16 / NCH;
  constexpr int TKS = (NCH <= 16) ? kTK : 64;        // keys per LDS stage (24-chunk rows: two 24 KB buffers, two workgroups per CU)
  constexpr int CSTAGES = kChunk / TKS;              // stages per canonical chunk
  constexpr int CHUNKS = TKS * NCH;
  constexpr int NLD = CHUNKS / kThreads;
  static_assert(CHUNKS % kThreads == 0, "every thread stages the same number of chunks");
  static_assert(SP != 0 || POW2 || NCH == 12, "LDS chunk placement is written for a power of two or 12 chunks per row");
  static_assert(NCH <= 48, "wider rows need a shorter stage");
  // two stage buffers: static up to 64 KB, dynamic beyond (the launch passes 2 * CHUNKS * 16 bytes)
  constexpr bool DYN = 2 * CHUNKS * sizeof(uint4) > 64 * 1024;
  __shared__ uint4 lds_static[DYN ? 1 : 2 * CHUNKS];
  uint4* const lds = DYN ? corr_direct_dyn_lds : lds_static;
  // where chunk c of key row `row` lives inside the row's NCH slots: an XOR swizzle for power-of-two rows; for 12-chunk rows a
  // rotation by (row / 4) % 4 — a row starts 12 row (mod 16) sixteen-byte banks in, which only depends on row % 4, and the
  // rotation separates the four rows of every such class: the 16 lanes of a ds_read_b128 phase hit 16 distinct banks.
  // split rows: key_slot<NCH, SP> (XOR inside each plane, corr_argmax.hip).
  auto slot = [](int row, int c) {
    if constexpr (SP != 0) return key_slot<NCH, SP>(row, c);
    else return POW2 ? (c ^ ((row / RPB) & (NCH - 1))) : (c + ((row >> 2) & 3)) % NCH;
  };

  // the wave index as a scalar: the LDS-DMA destination of a wave (M0) is then formed without vector instructions
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.y;
  const int q0 = (blockIdx.x * kWaves + wave) * (QB * 32);
  // diagnostics (workgroup (0, 0) only): shader-clock and 100 MHz reference counters at start and end
  const bool probe = blockIdx.x == 0 && blockIdx.y == 0;
  const long long t_sclk0 = probe ? (long long)__builtin_amdgcn_s_memtime() : 0;
  const long long t_ref0 = probe ? (long long)__builtin_amdgcn_s_memrealtime() : 0;

  bf16x8 bq[QB][NFR];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int row = q0 + qb * 32 + r;
    row = row < P ? row : P - 1;
    const uint16_t* src = Q + (size_t)row * ldq + 8 * h;
#pragma unroll
    for (int s = 0; s < NFR; ++s) bq[qb][s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  }
  // ONE key range (gridDim.y == 1: every launch whose query blocks alone fill the chip): this workgroup sees
  // every chunk of its queries, adds the chunk sums in f64 in ascending order itself — the very operations
  // corr_finalize_kernel performs on the stored chunk sums — and finishes its good queries in its epilogue.
  const bool whole = gridDim.y == 1;
  const float kn2_all = kn2_max(ws);            // max |k|^2, a scalar: read once, used by the screen, the range bound and the epilogue
  // |q|^2 for the error bound of the margin test (f32, upper bound to rounding; corr_finish inflates it).  The epilogue's copy
  // waits in LDS (2 KB per workgroup), not in registers: the plain-row kernel sits at its 168-register budget and used to
  // spill three registers per lane to scratch — 12 B written and read back per query, the whole of the 2.5 x write traffic
  // profiles/k1_hbm_traffic.json showed against 8 B of outputs per query.
  __shared__ float qn2_keep[QB][kThreads];
  float qn2[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float n2 = 0.f;
#pragma unroll
    for (int s = 0; s < RF::NQN; ++s)          // split rows: plane 1 (split_deff covers what the other planes add)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = elem_f32<F16>((uint16_t)bq[qb][s][e]);
        n2 = __builtin_fmaf(v, v, n2);
      }
    n2 += __shfl_xor(n2, 32, 64);
    qn2[qb] = n2;
    qn2_keep[qb][tid] = n2;
    const int q = q0 + qb * 32 + r;
    if (!whole && split == 0 && h == 0 && q < P) ws.qn2[q] = n2;
  }
  // Every query of the workgroup is the zero vector (the padding rows behind a crop's masked pixels in a capacity-sized
  // batch, isr_prep_queries_batch): each logit is exactly 0, each chunk sum the number of its keys (v_exp_f32(0) = 1 and sums
  // of ones are exact), the maximum 0 at the range's first key.  The workgroup hands on what the loop below would have
  // produced — corr_finish's inputs when it owns the whole key range, the range's partials otherwise (round 3: the key-split
  // route computed these rows in full, so a crop batch cost twice as much with two key ranges as with one) — and leaves
  // without touching the keys.
  {
    bool nonzero = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) nonzero |= qn2[qb] != 0.f;
    if constexpr (SP != 0) {                    // an f32 number too small for plane 1 still lives in planes 2 and 3
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int s = SP; s < NFR; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) nonzero |= (bq[qb][s][e] & 0x7FFF) != 0;
    }
    if (!__syncthreads_or(nonzero ? 1 : 0)) {
      if (whole) {
        const float kn2z = kn2_all;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const int q = q0 + qb * 32 + r;
          if (h == 0 && q < P)
            corr_finish<NAT ? 2 : 1>(q, 0.f, 0.f, 0, false, (double)N, 0.0, DEFF, EABS, 0.f, kn2z, ws, LSE ? nullptr : idx_out,
                                     LSE ? nullptr : logp_out, lse_out);
        }
      } else {
        const int zc0 = split * range_chunks;
        const int zk0 = zc0 * kChunk, zk1 = min(N, zk0 + range_chunks * kChunk);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const int q = q0 + qb * 32 + r;
          if (h != 0 || q >= P) continue;
          for (int c = zc0; c * kChunk < zk1; ++c) ws.plc[(size_t)c * P + q] = (float)(min(zk1, (c + 1) * kChunk) - c * kChunk);
          const size_t off = (size_t)split * P + q;
          ws.pm[off] = 0.f;
          ws.pm2[off] = zk1 - zk0 > 1 ? 0.f : -__builtin_inff();
          ws.pbi[off] = zk0;
          ws.pbad[off] = 0;
        }
      }
      if (tid == 0) ws.flags[blockIdx.y * gridDim.x + blockIdx.x] = 0;
      if (probe && tid == 0) { ws.clk[0] = 0; ws.clk[1] = 0; }
      return;
    }
  }
  DirectState st[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    st[qb].m = -__builtin_inff(); st[qb].m2 = -__builtin_inff(); st[qb].l = 0.f; st[qb].tb = 0; st[qb].L = 0.0;
#if defined(ISR_ABL_SCREEN) && ISR_ABL_SCREEN == 1
    st[qb].thr = __builtin_inff();
#else
    st[qb].thr = 0.f;
#endif
    if constexpr (SKIP != 0) {
      // the tile-skip threshold: (a lower bound of the query's maximum) - kSkipT, raw logit units
      const int q = min(q0 + qb * 32 + r, P - 1);
      st[qb].thr = ws.lower ? ws.lower[(size_t)q * ws.lower_stride] - ws.skip_T * (NAT ? 0.6931471805599453f : 1.f) : ws.skip_default;
    }
  }
  // the screen's distance below the maximum, in raw logit units: 2 eps of corr_finish (same |q|^2, same max |k|^2) inflated by
  // 1 %, plus 1e-4 for v_exp_f32 (1 ulp), the 15 additions of ts and the rounding of M - dlt (|M| < 128: 8e-6)
  float dlt[QB];
  {
    const float kn2s = kn2_all;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
      dlt[qb] = 2.02f * ((float)(DEFF + 2) * 1.1920929e-7f * 1.0001f * __builtin_sqrtf(qn2[qb] * kn2s) +
                         EABS * (__builtin_sqrtf(qn2[qb]) + __builtin_sqrtf(kn2s))) + 1e-4f;
  }

  const int c0 = split * range_chunks;                 // first canonical chunk of this key range
  const int k0 = c0 * kChunk;
  const int k1 = min(N, k0 + range_chunks * kChunk);
  const int nstage = (k1 - k0 + TKS - 1) / TKS;
  const int nfull = (k1 - k0) / TKS;                  // stages whose TKS keys all exist

  // Key rows come through a raw buffer descriptor over this key range: one 32-bit offset per load,
  // and rows beyond the range read as zero in hardware (no predicates in the loop).
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(K + (size_t)k0 * ldk), 0, (k1 - k0) * ldk * 2, 0x00020000);
  // Split rows go global -> LDS without a register stop (buffer_load ... lds: a wave instruction fills 64 consecutive
  // 16-byte units of the stage image, so the swizzle sits on the SOURCE side — unit u of the image takes the chunk that
  // belongs there, key_slot being its own inverse inside a plane): 24 staging registers and the ds_write pass less in a
  // kernel whose 24 + 12 operand fragments leave none to spare.
#ifndef ISR_DIRECT_DMA
#define ISR_DIRECT_DMA 1      // plain rows (power-of-two chunk counts) through the same DMA staging: -1.5 % at D = 64 (profiles/r04_k1_dma_ab.txt); 0: registers + ds_write
#endif
  constexpr bool DMA = SP != 0 || (ISR_DIRECT_DMA != 0 && POW2);
  int koff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int ci = tid + i * kThreads;
    koff[i] = DMA ? ((ci / NCH) * ldk + 8 * slot(ci / NCH, ci % NCH)) * 2 : ((ci / NCH) * ldk + 8 * (ci % NCH)) * 2;
  }
  uint4 stg[DMA ? 1 : NLD];
  auto gload = [&](int stage, int dbuf = -1) {       // dbuf: the destination buffer when the caller knows it at compile time
    const int so = stage * TKS * ldk * 2;
    if (dbuf < 0) dbuf = stage & 1;
    if constexpr (DMA) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass of this template rejects the LDS address-space cast (and then drops the kernel's stub)
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)&lds[dbuf * CHUNKS + i * kThreads + wave * 64],
                                                 16, koff[i] + so, 0, 0, 0);
#endif
    } else {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(krs, koff[i] + so, 0, 0);
        stg[i] = *reinterpret_cast<const uint4*>(&v);
      }
    }
  };
  auto lwrite = [&](int buf) {
    if constexpr (!DMA) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int ci = tid + i * kThreads;
        const int row = ci / NCH, c = ci % NCH;
        lds[buf * CHUNKS + row * NCH + slot(row, c)] = stg[i];
      }
    }
  };
  // A fragments of key sub-tile `sub` of LDS buffer `buf`.  One register set: a fragment's ds_read
  // for the NEXT sub-tile is issued right behind the last MFMA that reads the current one, i.e. one
  // whole item (~300 cycles) before the MFMA that needs it.
  bf16x8 a[NFR];
  // split rows: block j of every plane sits at the lane's rb[j] plus a compile-time offset (the XOR term of key_slot depends
  // on r alone: 32 rows are a multiple of its period) — one address register per block, the rest in the ds_read's immediate
  // plain rows of up to four blocks: the same, one register per block (the swizzle term of `slot` depends on r alone there too)
  constexpr bool RB = SP != 0 || NFR <= 4;
  int rb[RB ? (SP ? SP : NFR) : 1];
  if constexpr (RB) {
#pragma unroll
    for (int j = 0; j < (SP ? SP : NFR); ++j) rb[j] = r * NCH + slot(r, 2 * j + h);
  }
  auto load_a = [&](int buf, int sub) {
    if constexpr (SP == 0 && RB) {
      const uint4* lp = lds + buf * CHUNKS;
#pragma unroll
      for (int s = 0; s < NFR; ++s) {
        const uint4 v = lp[rb[s] + sub * 32 * NCH];
        a[s] = *reinterpret_cast<const bf16x8*>(&v);
      }
    } else if constexpr (SP != 0) {
      const uint4* lp = lds + buf * CHUNKS;
#pragma unroll
      for (int i = 0; i < NFR; ++i) {
        const int pl = 2 - i / SP, j = i % SP;        // in the order tile_chain lets go of the planes: k3, then k2, then k1
        const uint4 v = lp[rb[j] + sub * 32 * NCH + pl * 2 * SP];
        a[pl * SP + j] = *reinterpret_cast<const bf16x8*>(&v);
      }
    } else {
      const int row = sub * 32 + r;
#pragma unroll
      for (int s = 0; s < NFR; ++s) {
        const uint4 v = lds[buf * CHUNKS + row * NCH + slot(row, 2 * s + h)];
        a[s] = *reinterpret_cast<const bf16x8*>(&v);
      }
    }
  };

  // Software pipeline: item w's epilogue runs under item w + 1's MFMAs.  The stage barrier sits two
  // items before the stage's end (every LDS read of the stage has been issued by then), so the next
  // stage's first fragments are a full item ahead too.
  constexpr int NSUB = TKS / 32, NW = NSUB * QB;
  static_assert(QB == 2 && NSUB % 2 == 0, "item schedule below is written for two query blocks per wave");
  gload(0);
  lwrite(0);
  if constexpr (DMA) {                                 // DMA: stage s + 2 is requested in the last item of stage s
    if (nstage > 1) gload(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  load_a(0, 0);
  f32x16 acc[2];
  acc[0] = tile_chain<DKU, SP, F16>(a, bq[0], splat16(0.f));

  // TRK: track the maxima.  Always, except in an LSE kernel whose wave holds only queries the Cauchy-Schwarz bound keeps
  // inside the direct sum's range (the loop exists twice there; which copy runs is wave-uniform).
  // buf_tag: the stage's LDS buffer (stage & 1) as a compile-time constant — full stages run in even / odd pairs — so that
  // every LDS address of the loop is a lane register plus an immediate; -1: taken from the stage index at run time.
  auto stage_body = [&](int stage, auto full_tag, auto trk_tag, auto buf_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr bool TRK = !LSE || decltype(trk_tag)::value;
    constexpr int BUF = decltype(buf_tag)::value;
    const int buf = BUF >= 0 ? BUF : (stage & 1);
    const bool has_next = stage + 1 < nstage;
    if constexpr (!DMA) { if (has_next) gload(stage + 1); }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int sub = w / QB, qb = w % QB;
      const int qbn = (w + 1) % QB;
      if (w == NW - 2) {                                  // block-uniform
        if (has_next) lwrite(buf ^ 1);
        // DMA: this wave's pieces of the next stage (requested NW - 1 items ago) must have LANDED before the barrier
        // publishes the buffer — the compiler does not order a buffer_load ... lds against the ds_reads behind the
        // barrier by itself (the f16 instantiation had no vmcnt wait in its loop at all: results changed from call to call)
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef ISR_ABL_NOBARRIER  // timing-only ablation: what the stage barrier costs (results are wrong without it)
        __syncthreads();
#endif
      }
      // DMA: every wave is past the barrier above, so nobody reads this stage's buffer any more: the stage after the next
      // goes into it now, a whole stage ahead of the barrier that publishes it (the compiler drains the DMA counter in front
      // of the next ds_read, an item away)
      // (the pieces of a stage issued in two halves an item apart: 3 % slower at D = 64 with f16 planes, profiles/README.md)
      if constexpr (DMA) { if (w == NW - 1 && stage + 2 < nstage) gload(stage + 2, buf); }
      const int kb = k0 + stage * TKS + sub * 32;
      if constexpr (SP != 0) {
        // Split rows: an item is its plane pairs as phases (RowFrags: six with bf16 planes, three with f16 planes), SP matrix
        // instructions each, fenced by sched_barriers; each phase carries its share of the previous item's epilogue (the
        // maxima, then a slice of the 16 exp + add in register order: the canonical sum) and, in the items that read
        // fragments last, the ds_reads of the key plane it has just let go of.
        if (FULL || kb < k1) {  // block-uniform
          if (!FULL && kb + 32 > k1) mask_tail(acc[w & 1], kb + 4 * h, k1);
          const f32x16& cur = acc[w & 1];
          f32x16& nxt = acc[(w + 1) & 1];
          // the reads are unconditional in the items that own them: past the range's last stage they fetch stale LDS
          // into fragments nobody multiplies (a branch here would cut the phase into basic blocks the scheduler cannot
          // interleave across)
          const bool reload = qb == 0;                 // folds once the item loop is unrolled
          const int rbuf = (w == NW - 2) ? (buf ^ 1) : buf, rsub = (w == NW - 2) ? 0 : sub + 1;
          auto reads = [&](int pl) {
            if (reload) {
              const uint4* lp = lds + rbuf * CHUNKS;
#pragma unroll
              for (int j = 0; j < SP; ++j) {
                const uint4 v = lp[rb[j] + rsub * 32 * NCH + pl * 2 * SP];
                a[pl * SP + j] = *reinterpret_cast<const bf16x8*>(&v);
              }
            }
          };
          if constexpr (SCR) {
          // the screen (see ISR_K1_SCREEN above): the tile's own sum ts grows through the phases, joins the chunk sum in the last
          // one, and decides behind the item whether the maxima look at the tile at all
          float ts = 0.f;
          f32x16 c = splat16(0.f);
          auto phase = [&](auto ph_tag) {
            constexpr int ph = decltype(ph_tag)::value;
#pragma unroll
#if defined(ISR_ABL_MFMA16)
            for (int j = 0; j + 1 < SP; j += 2)
              c = mfma16_pair_as_16x16x32<F16>(a[RF::PA[ph] * SP + j], a[RF::PA[ph] * SP + j + 1], bq[qbn][RF::PB[ph] * SP + j],
                                               bq[qbn][RF::PB[ph] * SP + j + 1], c);
            if constexpr (SP % 2 == 1) c = mfma16<F16>(a[RF::PA[ph] * SP + SP - 1], bq[qbn][RF::PB[ph] * SP + SP - 1], c);
#else
            for (int j = 0; j < SP; ++j) c = mfma16<F16>(a[RF::PA[ph] * SP + j], bq[qbn][RF::PB[ph] * SP + j], c);
#endif
            if constexpr (RF::RD[ph] >= 0) reads(RF::RD[ph]);
#pragma unroll
            for (int i = RF::E0[ph]; i < RF::E0[ph + 1]; ++i) {
              const float e = __builtin_amdgcn_exp2f(NAT ? cur[i] * kLog2e : cur[i]);
              ts = i == 0 ? e : ts + e;
            }
            if constexpr (ph + 1 == RF::NPH) st[qb].l += ts;
            constexpr int ne = RF::E0[ph + 1] - RF::E0[ph];
            constexpr int nv = (NAT ? 3 : 2) * ne - (RF::E0[ph] == 0 && ne > 0 ? 1 : 0) + (ph + 1 == RF::NPH ? 1 : 0);
            constexpr int gv = (nv + SP - 1) / SP;
#pragma unroll
            for (int j = 0; j < SP; ++j) {
#if defined(ISR_ABL_MFMA16)
              __builtin_amdgcn_sched_group_barrier(0x008, SP % 2 == 0 ? 2 : 1, 0);
#else
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#endif
              if (reload && RF::RD[ph] >= 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              if constexpr (gv > 0) __builtin_amdgcn_sched_group_barrier(0x002, gv, 0);
            }
            if constexpr (ph + 1 < RF::NPH) {
              asm volatile("" : "+v"(c), "+v"(ts));
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          phase(std::integral_constant<int, 0>{});
          phase(std::integral_constant<int, 1>{});
          phase(std::integral_constant<int, 2>{});
          if constexpr (RF::NPH == 6) {
            phase(std::integral_constant<int, 3>{});
            phase(std::integral_constant<int, 4>{});
            phase(std::integral_constant<int, 5>{});
          }
          nxt = c;
          asm volatile("" : "+v"(nxt), "+v"(st[qb].l), "+v"(ts));
          if constexpr (TRK) {
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(ts >= st[qb].thr) != 0ull, 0)) {   // wave-uniform
              const float t = tile_max(cur);
              st[qb].m2 = __builtin_amdgcn_fmed3f(st[qb].m, st[qb].m2, t);   // second largest of {m, m2, t} (m2 <= m)
              st[qb].tb = (t > st[qb].m) ? kb : st[qb].tb;                     // strict: the first tile to reach m keeps it
              st[qb].m = fmaxf(st[qb].m, t);
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(st[qb].m), __float_as_uint(st[qb].m), false, false);
              const float mq = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
              const float ax = mq - dlt[qb];
              st[qb].thr = __builtin_amdgcn_exp2f(NAT ? ax * kLog2e : ax);
            }
          }
          } else {
          float l = st[qb].l;
          f32x16 c = splat16(0.f);
          auto phase = [&](auto ph_tag) {
            constexpr int ph = decltype(ph_tag)::value;
#pragma unroll
#if defined(ISR_ABL_MFMA16)
            for (int j = 0; j + 1 < SP; j += 2)
              c = mfma16_pair_as_16x16x32<F16>(a[RF::PA[ph] * SP + j], a[RF::PA[ph] * SP + j + 1], bq[qbn][RF::PB[ph] * SP + j],
                                               bq[qbn][RF::PB[ph] * SP + j + 1], c);
            if constexpr (SP % 2 == 1) c = mfma16<F16>(a[RF::PA[ph] * SP + SP - 1], bq[qbn][RF::PB[ph] * SP + SP - 1], c);
#else
            for (int j = 0; j < SP; ++j) c = mfma16<F16>(a[RF::PA[ph] * SP + j], bq[qbn][RF::PB[ph] * SP + j], c);
#endif
            if constexpr (RF::RD[ph] >= 0) reads(RF::RD[ph]);
            if constexpr (ph == 0 && TRK) {
              const float t = tile_max(cur);
              st[qb].m2 = __builtin_amdgcn_fmed3f(st[qb].m, st[qb].m2, t);   // second largest of {m, m2, t} (m2 <= m)
              st[qb].tb = (t > st[qb].m) ? kb : st[qb].tb;                     // strict: the first tile to reach m keeps it
              st[qb].m = fmaxf(st[qb].m, t);
            }
#pragma unroll
            for (int i = RF::E0[ph]; i < RF::E0[ph + 1]; ++i) l += __builtin_amdgcn_exp2f(NAT ? cur[i] * kLog2e : cur[i]);
            // issue order inside the phase: each matrix instruction followed by its share of the phase's VALU work (and, in
            // the items that reload fragments, one of the freed plane's ds_reads)
            constexpr int nv = ((ph == 0 && TRK) ? 12 : 0) + (NAT ? 3 : 2) * (RF::E0[ph + 1] - RF::E0[ph]);
            constexpr int gv = (nv + SP - 1) / SP;
#pragma unroll
            for (int j = 0; j < SP; ++j) {
#if defined(ISR_ABL_MFMA16)
              __builtin_amdgcn_sched_group_barrier(0x008, SP % 2 == 0 ? 2 : 1, 0);
#else
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#endif
              if (reload && RF::RD[ph] >= 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              if constexpr (gv > 0) __builtin_amdgcn_sched_group_barrier(0x002, gv, 0);
            }
            if constexpr (ph + 1 < RF::NPH) {
              asm volatile("" : "+v"(c), "+v"(l), "+v"(st[qb].m), "+v"(st[qb].m2), "+v"(st[qb].tb));
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          phase(std::integral_constant<int, 0>{});
          phase(std::integral_constant<int, 1>{});
          phase(std::integral_constant<int, 2>{});
          if constexpr (RF::NPH == 6) {
            phase(std::integral_constant<int, 3>{});
            phase(std::integral_constant<int, 4>{});
            phase(std::integral_constant<int, 5>{});
          }
          st[qb].l = l;
          nxt = c;
          asm volatile("" : "+v"(nxt), "+v"(st[qb].l));
          }
        }
      } else
      if (FULL || kb < k1) {  // block-uniform
        if (!FULL && kb + 32 > k1) mask_tail(acc[w & 1], kb + 4 * h, k1);
        const f32x16& cur = acc[w & 1];
        f32x16& nxt = acc[(w + 1) & 1];
        nxt = tile_chain<DKU, SP, F16>(a, bq[qbn], splat16(0.f));
        if (qb == 0) {                                    // the chain above was the fragments' last reader
          if (w == NW - 2) { if (has_next) load_a(buf ^ 1, 0); }
          else load_a(buf, sub + 1);
        }
        if constexpr (SKIP != 0) {
          // The dense form of the screened route's rule (corr_sparse.hpp): the tile's maximum first; its 16 exponentials only
          // when some lane's maximum reaches its query's threshold L_q - T (wave-uniform branch), each lane adding its own
          // piece's sum only when ITS maximum does — the pieces, their order and their bits are corr_fp6_sparse_kernel's
          float t = tile_max(cur);
          constexpr int GS = (9 + NMF - 1) / NMF;
#pragma unroll
          for (int s = 0; s < NMF; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (qb == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, GS, 0);
          }
          asm volatile("" : "+v"(nxt), "+v"(t));
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(t >= st[qb].thr) != 0ull, 0)) {   // wave-uniform
            float ts = __builtin_amdgcn_exp2f(NAT ? cur[0] * kLog2e : cur[0]);
#pragma unroll
            for (int i = 1; i < 16; ++i) ts += __builtin_amdgcn_exp2f(NAT ? cur[i] * kLog2e : cur[i]);
            st[qb].l += (t >= st[qb].thr) ? ts : 0.f;
            st[qb].m2 = __builtin_amdgcn_fmed3f(st[qb].m, st[qb].m2, t);
            st[qb].tb = (t > st[qb].m) ? kb : st[qb].tb;
            st[qb].m = fmaxf(st[qb].m, t);
          }
        } else {
#if ISR_K1_SCREEN
        // the tile's own sum first (the canonical order of a chunk sum: tile sums in register order, added tile after tile)
        // DK = 1 (one matrix instruction per tile: the item is all VALU): four interleaved partial sums (registers i, i + 4,
        // i + 8, i + 12), then (c0 + c1) + (c2 + c3) — the same 15 additions as one chain, a quarter of its dependent length:
        // 17.98 -> 17.22 ms per 32-image launch at D = 16; at D = 64 / 128 and on the plane routes one chain is as fast or
        // 1 % faster (profiles/r04_k1_screen.txt, section 8)
        float ts;
        if constexpr (DK == 1) {
          float tc[4];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f(NAT ? cur[i] * kLog2e : cur[i]);
            tc[i & 3] = i < 4 ? e : tc[i & 3] + e;
          }
          ts = (tc[0] + tc[1]) + (tc[2] + tc[3]);
        } else {
          ts = __builtin_amdgcn_exp2f(NAT ? cur[0] * kLog2e : cur[0]);
#pragma unroll
          for (int i = 1; i < 16; ++i) ts += __builtin_amdgcn_exp2f(NAT ? cur[i] * kLog2e : cur[i]);
        }
        st[qb].l += ts;
        // issue order: the item's 33 VALU instructions (16 exp2, 15 + 1 add, the screen's compare; NAT: 16 mul more) spread
        // evenly behind the DK MFMAs of the next item; the empty asm ties the item's results to a fixed point of the
        // instruction stream (without it instruction selection sinks all eight epilogues below all eight MFMA chains)
        constexpr int G = ((NAT ? 49 : 33) + NMF - 1) / NMF;
#pragma unroll
        for (int s = 0; s < NMF; ++s) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (qb == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // this fragment's next ds_read
          __builtin_amdgcn_sched_group_barrier(0x002, G, 0);
        }
        asm volatile("" : "+v"(nxt), "+v"(st[qb].l), "+v"(ts));
#ifndef ISR_ABL_DNOMAX   // timing-only ablation (tools/ablate_direct.sh): no maxima at all
        if constexpr (TRK) {
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(ts >= st[qb].thr) != 0ull, 0)) {   // wave-uniform
            const float t = tile_max(cur);
            st[qb].m2 = __builtin_amdgcn_fmed3f(st[qb].m, st[qb].m2, t);   // second largest of {m, m2, t} (m2 <= m)
            st[qb].tb = (t > st[qb].m) ? kb : st[qb].tb;                     // strict: the first tile to reach m keeps it
            st[qb].m = fmaxf(st[qb].m, t);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(st[qb].m), __float_as_uint(st[qb].m), false, false);
            const float mq = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));   // [0]: the h = 0 lane's value on both lanes, [1]: the h = 1 lane's
            const float a = mq - dlt[qb];
#if !defined(ISR_ABL_SCREEN)   // timing-only ablations: 1 = thr stays +inf (no tile takes the slow path), 2 = thr stays 0 (every tile does)
            st[qb].thr = __builtin_amdgcn_exp2f(NAT ? a * kLog2e : a);
#else
            asm volatile("" :: "v"(a));
#endif
          }
        }
#endif
#else
#ifdef ISR_ABL_DNOMAX   // timing-only ablations (tools/ablate_direct.sh): keep the accumulator live
        asm volatile("" :: "v"(cur[0]), "v"(cur[15]));
        st[qb].m = 0.f;
#else
        if constexpr (TRK) {
          const float t = tile_max(cur);
          st[qb].m2 = __builtin_amdgcn_fmed3f(st[qb].m, st[qb].m2, t);   // second largest of {m, m2, t} (m2 <= m)
          st[qb].tb = (t > st[qb].m) ? kb : st[qb].tb;                     // strict: the first tile to reach m keeps it
          st[qb].m = fmaxf(st[qb].m, t);
        }
#endif
        float l = st[qb].l;
#ifdef ISR_ABL_DNOEXP
        asm volatile("" :: "v"(cur[1]), "v"(cur[14]));
        l = 1.f;
#else
#pragma unroll
        for (int i = 0; i < 16; ++i)
          l += __builtin_amdgcn_exp2f(NAT ? cur[i] * kLog2e : cur[i]);
#endif
        st[qb].l = l;
        // issue order: the item's 44 VALU instructions (7 max3 + max, med3, cmp, cndmask, max, 16 exp2,
        // 16 add; NAT: 16 mul more) spread evenly behind the DK MFMAs of the next item.  The empty asm
        // ties the item's results to a fixed point of the instruction stream: a stage is one basic
        // block, and without it instruction selection sinks all eight epilogues below all eight MFMA
        // chains (eight tiles live, 243 VGPRs, nothing overlapped).
        constexpr int G = ((NAT ? 60 : 44) - (TRK ? 0 : 12) + NMF - 1) / NMF;
#pragma unroll
        for (int s = 0; s < NMF; ++s) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          // this fragment's next ds_read
          if (qb == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, G, 0);
        }
        asm volatile("" : "+v"(nxt), "+v"(st[qb].l), "+v"(st[qb].m), "+v"(st[qb].m2), "+v"(st[qb].tb));
#endif
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // canonical chunks: kChunkStages stages each; the sums of a chunk leave the registers at its end
  bool over[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) over[qb] = false;
  auto run_chunks = [&](auto trk_tag) {
    int stage = 0;
    for (int c = c0; stage < nstage; ++c) {
      const int send = min(nstage, stage + CSTAGES);
      const int sfull = min(nfull, send);
      // a chunk starts at an even stage (kChunk / TKS stages per chunk, ranges of whole chunks)
      for (; stage + 1 < sfull; stage += 2) {
        stage_body(stage, std::true_type{}, trk_tag, std::integral_constant<int, 0>{});
        stage_body(stage + 1, std::true_type{}, trk_tag, std::integral_constant<int, 1>{});
      }
      for (; stage < sfull; ++stage) stage_body(stage, std::true_type{}, trk_tag, std::integral_constant<int, -1>{});
      if (stage < send) { stage_body(stage, std::false_type{}, trk_tag, std::integral_constant<int, -1>{}); ++stage; }   // only a range's last stage is partial
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float lc = st[qb].l + __shfl_xor(st[qb].l, 32, 64);
        over[qb] |= !(lc <= 3.0e38f);     // the same on both lanes of the query
        const int q = q0 + qb * 32 + r;
        if (whole) st[qb].L += (double)lc;                    // the first chunk: 0 + l_0 = l_0, as in the finalize merge
        else if (h == 0 && q < P) ws.plc[(size_t)c * P + q] = lc;
        st[qb].l = 0.f;
      }
    }
  };
  // LSE: does any query of this wave need its maximum to decide whether the direct sum applies?  (|s'| <= |q||k|_max < 99
  // log2 units keeps every term a normal f32 and the maximum above kLow: the full kernel calls such a query good too.)
  bool trk = true;
  if constexpr (LSE) {
    const float kn2l = kn2_all;
    bool unsure = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) unsure |= !((NAT ? kLog2e * kLog2e : 1.f) * qn2[qb] * kn2l < kLseBound2);
    trk = __any(unsure) != 0;
  }
  if (trk) run_chunks(std::true_type{});
  else run_chunks(std::false_type{});

  // ---- row recovery: one MFMA chain per distinct winning tile of the wave's queries
  bool any_bad_lane = false;
  if constexpr (LSE) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      // the full kernel's rule — overflow, or a maximum below kLow — on the maximum where the wave tracked it; where it did
      // not, the bound has shown that neither can happen
      const float M = fmaxf(st[qb].m, __shfl_xor(st[qb].m, 32, 64));
      const bool bad_q = over[qb] || (trk && !((NAT ? M * kLog2e : M) >= kLow));   // the same on both lanes of the query
      const int q = q0 + qb * 32 + r;
      if (whole) over[qb] = bad_q;
      else if (h == 0 && q < P) ws.pbad[(size_t)split * P + q] = bad_q ? 1 : 0;
      any_bad_lane |= bad_q && q < P;
    }
  } else
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    // the two lanes (h = 0, 1) of a query agree on (M, T): the maximum and the lowest tile reaching it
    const float mo = __shfl_xor(st[qb].m, 32, 64);
    const int tbo = __shfl_xor(st[qb].tb, 32, 64);
    const float M = fmaxf(st[qb].m, mo);
    const int T = (st[qb].m == M) ? ((mo == M) ? min(st[qb].tb, tbo) : st[qb].tb) : tbo;
    int cand = T;
    float cmax = -__builtin_inff(), c2 = -__builtin_inff();
    unsigned long long todo = __ballot(true);
    auto fetch = [&](int kb, bf16x8 (&dst)[NFR]) {
      int row = kb + r;
      row = row < N ? row : N - 1;
      const uint16_t* src = K + (size_t)row * ldk + 8 * h;
#pragma unroll
      for (int s = 0; s < NFR; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
    };
    bf16x8 a0[NFR], a1[NFR];
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
      f32x16 c = tile_chain<DKU, SP, F16>(a0, bq[qb], splat16(0.f));
      if (kb_cur + 32 > k1) mask_tail(c, kb_cur + 4 * h, k1);
      if (T == kb_cur) {
        // the two largest of this lane's 16 rows (with multiplicity) and the lowest row of the largest
        float a1v = -__builtin_inff(), a2v = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          a2v = __builtin_amdgcn_fmed3f(a1v, a2v, c[i]);
          a1v = fmaxf(a1v, c[i]);
        }
        cmax = a1v; c2 = a2v;
        int rr = 15;
#pragma unroll
        for (int i = 14; i >= 0; --i) rr = (c[i] == cmax) ? i : rr;   // lowest register = lowest key
        cand = kb_cur + 4 * h + (rr & 3) + 8 * (rr >> 2);
      }
      if (!more) break;
#pragma unroll
      for (int s = 0; s < NFR; ++s) a0[s] = a1[s];
      kb_cur = kb_nxt;
    }
    // runner-up of the query inside this key range.  The lane that owns the winner contributes its
    // second tile maximum and the second row of the winning tile, the other lane its best.
    const int co = __shfl_xor(cand, 32, 64);
    const float cmo = __shfl_xor(cmax, 32, 64);
    const bool other_wins = better(cmo, co, cmax, cand);
    const float mine = other_wins ? st[qb].m : fmaxf(st[qb].m2, c2);
    const float run = fmaxf(mine, __shfl_xor(mine, 32, 64));
    if (other_wins) { cand = co; cmax = cmo; }
    const float mlog2 = NAT ? cmax * kLog2e : cmax;
    const bool bad_q = over[qb] || !(mlog2 >= kLow);      // the same on both lanes of the query
    const int q = q0 + qb * 32 + r;
    if (whole) {
      st[qb].m = cmax; st[qb].m2 = run; st[qb].tb = cand;     // kept for the epilogue below
      over[qb] = bad_q;
    } else if (h == 0 && q < P) {
      const size_t off = (size_t)split * P + q;
      ws.pm[off] = cmax;
      ws.pm2[off] = run;
      ws.pbi[off] = cand;
      ws.pbad[off] = bad_q ? 1 : 0;
    }
    any_bad_lane |= bad_q && q < P;
  }
  const int any_bad = __syncthreads_or(any_bad_lane ? 1 : 0);
  if (tid == 0) {
    const int ent = blockIdx.y * gridDim.x + blockIdx.x;
    ws.flags[ent] = any_bad;
    if (any_bad) ws.blist[atomicAdd(&ws.rcount[1], 1)] = ent;      // the fallback kernel's work list
  }
  if (probe && tid == 0) {
    ws.clk[0] = (long long)__builtin_amdgcn_s_memtime() - t_sclk0;
    ws.clk[1] = (long long)__builtin_amdgcn_s_memrealtime() - t_ref0;
  }
  if (!whole) return;
  // ---- one key range: finish the good queries here (corr_finalize_kernel only revisits workgroups with a
  // bad query, and needs the per-query marks only then)
  const float kn2 = kn2_all;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q0 + qb * 32 + r;
    if (h != 0 || q >= P) continue;
    if (any_bad) {
      ws.pbad[q] = over[qb] ? 1 : 0;
      // corr_finalize_kernel finishes the bad queries of this block and reads |q|^2 for the margin test from the
      // workspace (this launch stored it nowhere else: the split == 0 store above is the key-split route's)
      if (over[qb]) ws.qn2[q] = qn2_keep[qb][tid];
    }
    if (!over[qb])
      corr_finish<NAT ? 2 : 1>(q, LSE ? 0.f : st[qb].m, LSE ? 0.f : st[qb].m2, LSE ? 0 : st[qb].tb, false, st[qb].L, 0.0, DEFF, EABS,
                               qn2_keep[qb][tid], kn2, ws, LSE ? nullptr : idx_out, LSE ? nullptr : logp_out, lse_out);
  }
}
