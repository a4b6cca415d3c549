// corr_sparse.hpp — K1 for bf16 descriptors of D = 64 behind a rigorous low-precision SCREEN (round 5).  Included by
// corr_argmax.hip inside its anonymous namespace after corr_direct.hpp (shares DirectState, tile_max, corr_finish, CorrWs).
//
// corr_bf16_direct_kernel is bound by VALU issue on the P x N exponentials (52 of a tile's ~75 issue slots) and cannot get
// faster in that formulation.  But almost none of those exponentials can be seen in the result: with M the query's maximum
// logit (log2 units), a term more than T = 21 + ceil(log2 N) below M is, summed over all N keys, below 2^-21 of the sum —
// 5e-7 in lse, under the f32 spacing of every lse above 4.  So the CANONICAL sum of this route is defined as
//     l = sum over the (16-key lane group, 32-key tile) pieces whose largest logit reaches  L_q - T,
// L_q <= M a per-query LOWER bound of the maximum that depends on (query, keys) only (below).  The arg-max, the runner-up of
// the margin test and every term the sum keeps live in pieces that reach L_q - T; what is skipped is decided per QUERY, so a
// query's outputs stay a function of (query, keys) alone.
//
// What makes this pay: a piece can be PROVEN to lie below L_q - T without computing it.  Rows are also held as block-scaled
// FP6 (e2m3, one E8M0 scale per 32 elements): v_mfma_scale_f32_32x32x64_f8f6f4 forms a whole 32 x 32 x 64 tile of
// approximate logits s~ in ONE instruction at a quarter of the four bf16 instructions' cycles, and
//     |s - s~| <= |q| |dk| + |dq| |k~| + (f32 accumulation)  =: E_q       (dk = k - k~, dq = q - q~; Cauchy-Schwarz)
// with the norms computed exactly when the rows are quantised.  A tile whose s~ stay below L_q - T - E_q for every lane is
// skipped after 1 matrix instruction + 9 VALU; the others (the tile of a query's winner, and the few a noise tail reaches:
// 5 % + 1.5 % on |k| = 8 planted data) are formed on the bf16 matrix cores from the ORIGINAL rows, and only those results enter
// sums and maxima: indices, logp and lse never see an FP6 number.
//
// L_q: pass 0 (corr_fp6_lower_kernel) finds, per query, A 32-key tile holding the largest s~ up to the 13 mantissa bits the
// tile's number displaces in the running maximum (among equal values the bit pattern decides: the highest-numbered tile for
// positive s~, e.g. the LAST tile for a zero query — a deterministic function of the two quantised rows either way), redoes
// that tile in bf16 and takes its largest logit: a true logit of the query, hence a lower bound of its maximum — on data with
// a clear winner it IS the maximum.  Nothing downstream assumes which of several tied tiles it is: pass 1 enters every piece in
// tile order, so "the first tile reaching the maximum" is found as in the dense kernel.
// Worst case (flat logits: nothing can be skipped): corr_fp6_sparse_kernel screens its first stage ahead of its main loop and hands
// a 256-query block with more than a quarter of those items flagged to the dense tile-skip kernel
// (corr_bf16_direct_kernel<.., SKIP = 1>), which applies the same rule to the same bf16 logits: identical bits either way, and
// the call then costs pass 0 + the dense kernel (1.4 x the unscreened one).  That is also what happens on the bench's own data
// (|k| = 5: every term of the sums counts to 5e-7) — the route is an opt-in for softmaxes peaked beyond f32 resolution.
#pragma once

using i32x8 = __attribute__((ext_vector_type(8))) int;

constexpr int kQ6Row = 64;          // bytes of a quantised row of 64 elements: per 32-element half 24 B of codes, 1 B scale, 7 B zero
constexpr int kScreenTBase = 21;    // T = kScreenTBase + ceil(log2 N)
constexpr int kLowStride = 10;      // floats pass 0 leaves per query: L_q, its tile T (int bits), {piece maximum, piece sum, runner-up, winner key} of lane halves 0, 1
constexpr int kScreenMaxN = 1 << 18; // pass 0 carries the tile index in 13 mantissa bits

// T of a call: a piece more than T below L_q is left out of the sum; all N keys' worth of such pieces stay below 2^-kScreenTBase of it
inline int screen_T(int N) {
  int tl = 0;
  while ((1 << tl) < N) ++tl;
  return kScreenTBase + tl;
}
// which calls take the screened route: the dtype asks for it, the rows are 64 wide and the tile index fits pass 0's 13 bits;
// every other ISR_DTYPE_BF16_LOG2_SCREENED call runs the unscreened log2-domain kernels
inline bool screened_route(int dtype, int N, int D) { return dtype == ISR_DTYPE_BF16_LOG2_SCREENED && D == 64 && N <= kScreenMaxN; }

// ------------------------------------------------------------------------------------------ quantisation
// One thread per 32-element half of a row.  e2m3 codes c = 0..31 of |x| / 2^e: c / 8 below 2, 2 + (c - 16) / 4 below 4, 4 + (c - 24) / 2
// up to 7.5 (c = 31); bit 5 the sign; 2^e the smallest power of two with max |x| / 2^e <= 7.5.  The norms are those of the
// values the matrix instruction will see: |x|^2, |x - x~|^2, |x~|^2 (f32; the callers inflate).
// KEYS: the maxima of |dk|^2 and |k~|^2 over all rows go to kmax[0], kmax[1] by integer atomic max (order-free).
// QUERIES: per row {|q|, |dq|} to nrm[2 row], nrm[2 row + 1].
template <bool QUERY>
__global__ __launch_bounds__(256) void corr_quant_fp6_kernel(const uint16_t* __restrict__ X, int R, int ld, uint8_t* __restrict__ out,
                                                             float* __restrict__ nrm, uint32_t* __restrict__ kmax, CorrWs ws) {
  if (gated_off(ws)) return;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < 2l * R;
  const long row = live ? (i >> 1) : 0;
  const int half = (int)(i & 1);
  const uint16_t* src = X + row * ld + 32 * half;
  float x[32];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const uint4 u = *reinterpret_cast<const uint4*>(src + 8 * v);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[8 * v + 2 * j] = __uint_as_float(w[j] << 16);
      x[8 * v + 2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
    }
  }
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) mx = fmaxf(mx, fabsf(x[j]));
  // e = ceil(log2(mx / 7.5)), clamped to E8M0's range; an all-zero half takes the smallest scale (its codes are 0)
  int e = -127;
  if (mx > 0.f) {
    int p;
    const float f = frexpf(mx * (1.f / 7.5f), &p);          // mx / 7.5 = f 2^p, f in [0.5, 1)
    e = (f > 0.5f) ? p : p - 1;
    if (mx > 7.5f * ldexpf(1.f, e)) ++e;                    // the rounding of the product above
    e = e < -127 ? -127 : (e > 127 ? 127 : e);
  }
  const float inv = ldexpf(1.f, -e), sc = ldexpf(1.f, e);    // (e = 127 with a value near bf16's maximum: inv is subnormal — still exact)
  uint32_t d[6] = {0, 0, 0, 0, 0, 0};
  float n2 = 0.f, d2 = 0.f, t2 = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float y = fminf(fabsf(x[j]) * inv, 7.5f);
    int c;
    float v;
    if (y < 2.f) { c = (int)rintf(y * 8.f); v = (float)c * 0.125f; }
    else if (y < 4.f) { c = 16 + (int)rintf((y - 2.f) * 4.f); v = 2.f + (float)(c - 16) * 0.25f; }
    else { c = 24 + (int)rintf((y - 4.f) * 2.f); c = c > 31 ? 31 : c; v = 4.f + (float)(c - 24) * 0.5f; }
    const float xq = copysignf(v * sc, x[j]);
    const float dx = x[j] - xq;                               // exact: both are short-mantissa numbers a few binades apart
    n2 = __builtin_fmaf(x[j], x[j], n2);
    d2 = __builtin_fmaf(dx, dx, d2);
    t2 = __builtin_fmaf(xq, xq, t2);
    const uint32_t code = (uint32_t)c | ((__float_as_uint(x[j]) >> 31) << 5);
    const int pos = 6 * j, w = pos >> 5, o = pos & 31;
    d[w] |= code << o;
    if (o > 26) d[w + 1] |= code >> (32 - o);
  }
  if (live) {
    uint4* dst = reinterpret_cast<uint4*>(out + row * kQ6Row + 32 * half);
    dst[0] = make_uint4(d[0], d[1], d[2], d[3]);
    dst[1] = make_uint4(d[4], d[5], (uint32_t)(e + 127), 0u);
  }
  // the row's norms: this thread's half + its neighbour's
  n2 += __shfl_xor(n2, 1, 64);
  d2 += __shfl_xor(d2, 1, 64);
  t2 += __shfl_xor(t2, 1, 64);
  if (!live || half != 0) return;
  if (QUERY) {
    nrm[2 * row] = __builtin_sqrtf(n2) * 1.00001f;
    nrm[2 * row + 1] = __builtin_sqrtf(d2) * 1.00001f;
  } else {
    atomicMax(&kmax[0], __float_as_uint(d2));
    atomicMax(&kmax[1], __float_as_uint(t2));
  }
}

// E_q = (|q| max|dk| + |dq| max|k~|) (1 + 1e-4)  +  what the matrix instruction's own f32 accumulation of the 64 exact products can
// add (2^-16 |q~||k~| is 128 x the f32 bound)  +  the bf16 chain's own error against the exact logit (eps of the margin test)
__device__ __forceinline__ float screen_error(float qn, float dqn, float dk2max, float kt2max) {
  const float dk = __builtin_sqrtf(dk2max) * 1.00001f, kt = __builtin_sqrtf(kt2max) * 1.00001f;
  return (qn * dk + dqn * kt) * 1.0001f + (qn + dqn) * kt * (1.53e-5f + 66.f * 1.1920929e-7f * 1.01f) + 1e-6f;
}

// one 32 x 32 x 64 tile of approximate logits: keys (A) and queries (B) as e2m3 with their block scales in register 6's low byte
__device__ __forceinline__ f32x16 mfma_fp6(const i32x8& a, const i32x8& b) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, splat16(0.f), 2, 2, 0, a[6], 0, b[6]);
}

// ------------------------------------------------------------------------------------------ shared staging
// The quantised keys stream through LDS like 64-byte bf16 rows (four 16-byte chunks per row, XOR-swizzled: key_slot<4, 0>),
// by buffer_load ... lds, TKQ keys per stage, two buffers.
#ifndef ISR_Q6_TKQ
#define ISR_Q6_TKQ 256
#endif
constexpr int kTKQ = ISR_Q6_TKQ;                       // keys per stage of the quantised stream
constexpr int kQ6Chunks = kTKQ * 4;                    // 16-byte chunks per stage
constexpr int kQ6Nld = kQ6Chunks / kThreads;           // DMA instructions per thread and stage

struct Q6Stream {
  __amdgpu_buffer_rsrc_t krs;
  int koff[kQ6Nld];
  uint4* lds;
  int wave;
  __device__ __forceinline__ void init(const uint8_t* K6, int k0, int k1, uint4* lds_, int tid, int wave_) {
    krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(K6 + (size_t)k0 * kQ6Row), 0, (k1 - k0) * kQ6Row, 0x00020000);
    lds = lds_;
    wave = wave_;
#pragma unroll
    for (int i = 0; i < kQ6Nld; ++i) {
      const int u = tid + i * kThreads, row = u >> 2, slot = u & 3;
      koff[i] = row * kQ6Row + 16 * key_slot<4, 0>(row, slot);        // the swizzle on the SOURCE side (its own inverse)
    }
  }
  __device__ __forceinline__ void gload(int stage, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < kQ6Nld; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)&lds[buf * kQ6Chunks + i * kThreads + wave * 64],
                                               16, koff[i] + stage * kTKQ * kQ6Row, 0, 0, 0);
#endif
  }
  // this lane's fragment of key sub-tile `sub`: the 32-byte half h of row sub * 32 + r (chunks 2h, 2h + 1)
  __device__ __forceinline__ i32x8 frag(int buf, int sub, int r, int h) const {
    const int row = sub * 32 + r;
    const uint4 lo = lds[buf * kQ6Chunks + row * 4 + key_slot<4, 0>(row, 2 * h)];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    u32x4 hi = *reinterpret_cast<const u32x4*>(&lds[buf * kQ6Chunks + row * 4 + key_slot<4, 0>(row, 2 * h + 1)]);
    // all four dwords of the second chunk are "used": left alone, the compiler narrows the read to ds_read_b96 (the matrix
    // instruction reads six registers and the scale), which the swizzle is not conflict-free for — 39 % of the kernels' LDS
    // cycles were bank conflicts (profiles/r05_k1_screen.txt, section 4)
    asm volatile("" : "+v"(hi));
    return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
  }
};

// One stage of the quantised stream for a wave's QB 32-query blocks, software-pipelined over the flat list of items
// w = sub * QB + qb taken two at a time: the two matrix instructions of pair p + 1 are issued before the results of pair p are
// looked at, and a sub-tile's fragment is requested a whole sub-tile (QB items) before its first use.  A fragment feeds QB
// matrix instructions: QB = 4 halves the LDS reads per item (at QB = 2 four SIMDs' ds_reads alone fill the LDS pipe at the
// matrix-instruction rate).
// consume(c, qb, kb): this lane's 16 approximate logits of query block qb against keys kb .. kb + 31.
// All kTKQ keys of the stage exist (q6_stage_partial takes a range's last, partial stage).
template <int QB, class Consume>
__device__ __forceinline__ void q6_stage(const Q6Stream& ks, int buf, int kbase, int r, int h, const i32x8 (&bq6)[QB], Consume&& consume) {
  constexpr int NSUB = kTKQ / 32, NP = NSUB * QB / 2;
  static_assert(QB % 2 == 0, "items are taken in pairs of query blocks");
  i32x8 af[2];
  af[0] = ks.frag(buf, 0, r, h);
  af[1] = ks.frag(buf, 1, r, h);
  f32x16 c0 = mfma_fp6(af[0], bq6[0]), c1 = mfma_fp6(af[0], bq6[1]);
  f32x16 n0 = c0, n1 = c1;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int sub = (2 * p) / QB, qb = (2 * p) % QB;
    const int sub1 = (2 * (p + 1)) / QB, qb1 = (2 * (p + 1)) % QB;
    if (p + 1 < NP) {
      n0 = mfma_fp6(af[sub1 & 1], bq6[qb1]);
      n1 = mfma_fp6(af[sub1 & 1], bq6[qb1 + 1]);
      // the pair just issued opens sub-tile sub1: sub-tile sub1 - 1's fragment has had its last reader, sub1 + 1's goes there
      if (qb1 == 0 && sub1 + 1 < NSUB) af[(sub1 + 1) & 1] = ks.frag(buf, sub1 + 1, r, h);
    }
    consume(c0, qb, kbase + sub * 32);
    consume(c1, qb + 1, kbase + sub * 32);
    c0 = n0; c1 = n1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the key range's last, partial stage (N not a multiple of kTKQ): plain item by item, sub-tiles at or beyond N untouched
template <int QB, class Consume>
__device__ __forceinline__ void q6_stage_partial(const Q6Stream& ks, int buf, int kbase, int N, int r, int h, const i32x8 (&bq6)[QB],
                                                 Consume&& consume) {
#pragma nounroll
  for (int sub = 0; sub < kTKQ / 32 && kbase + sub * 32 < N; ++sub) {
    const i32x8 a = ks.frag(buf, sub, r, h);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const f32x16 c = mfma_fp6(a, bq6[qb]);
      consume(c, qb, kbase + sub * 32);
    }
  }
}

__device__ __forceinline__ i32x8 load_q6(const uint8_t* Q6, int row, int h) {
  const uint4* p = reinterpret_cast<const uint4*>(Q6 + (size_t)row * kQ6Row + 32 * h);
  const uint4 lo = p[0], hi = p[1];
  return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
}

// the exact (bf16 MFMA, C = 0) logits of key tile kb for a 32-query block: this lane's 16 rows
__device__ __forceinline__ f32x16 exact_tile(const uint16_t* __restrict__ K, int N, int ldk, int kb, int r, int h, const bf16x8 (&bq)[4]) {
  int row = kb + r;
  row = row < N ? row : N - 1;
  const uint16_t* src = K + (size_t)row * ldk + 8 * h;
  bf16x8 a[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  f32x16 c = splat16(0.f);
#pragma unroll
  for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bq[s], c, 0, 0, 0);
  if (kb + 32 > N) mask_tail(c, kb + 4 * h, N);
  return c;
}

// ------------------------------------------------------------------------------------------ pass 0: L_q
// grid = query blocks of 512 (4 waves x kQB0 x 32 queries), the whole key range per workgroup.
#ifndef ISR_Q6_QB0
#define ISR_Q6_QB0 4
#endif
#ifndef ISR_Q6_QB1
#define ISR_Q6_QB1 2
#endif
constexpr int kQB0 = ISR_Q6_QB0;                              // 32-query blocks per wave in pass 0
constexpr int kQPB0 = kWaves * kQB0 * 32;                      // queries per workgroup
#ifndef ISR_Q6_W0
#define ISR_Q6_W0 3
#endif
#ifndef ISR_Q6_W1
#define ISR_Q6_W1 (ISR_Q6_QB1 <= 2 ? 3 : 2)
#endif
__global__ __launch_bounds__(kThreads, ISR_Q6_W0) void corr_fp6_lower_kernel(const uint8_t* __restrict__ Q6, const uint8_t* __restrict__ K6,
                                                                     const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K, int P, int N,
                                                                     int ldq, int ldk, float* __restrict__ lower, CorrWs ws) {
  __shared__ uint4 lds[2 * kQ6Chunks];
  if (gated_off(ws)) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  constexpr int QB = kQB0;
  const int q0 = (blockIdx.x * kWaves + wave) * (QB * 32);
  i32x8 bq6[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) bq6[qb] = load_q6(Q6, min(q0 + qb * 32 + r, P - 1), h);
  const int nstage = (N + kTKQ - 1) / kTKQ;
  Q6Stream ks;
  ks.init(K6, 0, N, lds, tid, wave);
  ks.gload(0, 0);
  if (nstage > 1) ks.gload(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // per lane and query block: the largest approximate logit so far WITH ITS TILE in the 13 low mantissa bits (one v_and_or +
  // one v_max per item instead of compare / select / max: an approximate logit loses nothing it has by being perturbed by
  // 2^-10 of itself, and the tile it names is redone exactly whichever it is)
  float m[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) m[qb] = -__builtin_inff();
  const unsigned himask = 0xFFFFE000u;
  const int nfull = N / kTKQ;                               // stages whose kTKQ keys all exist
  // rows past N are zero codes: s~ = 0 there (a tile of the loop still holds a real key, so L_q stays a true logit)
  auto consume = [&](const f32x16& c, int qb, int kb) __attribute__((always_inline)) {
#if defined(ISR_ABL_P0) && (ISR_ABL_P0 & 1)      // timing-only ablation: no VALU on the tile
    asm volatile("" :: "v"(c));
    m[qb] = 0.f;
#else
    const float t = tile_max(c);
    m[qb] = fmaxf(m[qb], __uint_as_float((__float_as_uint(t) & himask) | (unsigned)(kb >> 5)));
#endif
  };
  for (int stage = 0; stage < nstage; ++stage) {
    const int buf = stage & 1;
    if (stage < nfull) q6_stage<QB>(ks, buf, stage * kTKQ, r, h, bq6, consume);
    else q6_stage_partial<QB>(ks, buf, stage * kTKQ, N, r, h, bq6, consume);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the next stage have landed
#if !(defined(ISR_ABL_P0) && (ISR_ABL_P0 & 4))   // timing-only ablation: no stage barrier
    __syncthreads();                                          // ... and everybody is done reading this one
#endif
#if !(defined(ISR_ABL_P0) && (ISR_ABL_P0 & 2))   // timing-only ablation: no key traffic after the first two stages
    if (stage + 2 < nstage) ks.gload(stage + 2, buf);
#endif
  }
  // the tile of the query's largest approximate logit, redone exactly: one bf16 chain per distinct tile of the block's 32 queries;
  // a lane keeps the 16 logits of ITS query's tile and evaluates them once behind the loop
  bf16x8 bq[4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q0 + qb * 32 + r;
    {
      const uint16_t* src = Q + (size_t)min(q, P - 1) * ldq + 8 * h;
#pragma unroll
      for (int s = 0; s < 4; ++s) bq[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
    }
    const float M = fmaxf(m[qb], __shfl_xor(m[qb], 32, 64));
    const int T = (int)(__float_as_uint(M) & ~himask) << 5;
    f32x16 cs = splat16(0.f);
    unsigned long long todo = __ballot(true);
    while (todo) {                                            // wave-uniform: <= 32 trips
      const int kb = __shfl(T, __ffsll(todo) - 1, 64);
      todo &= ~__ballot(T == kb);
      const f32x16 c = exact_tile(K, N, ldk, kb, r, h, bq);
      if (T == kb) cs = c;
    }
    // this lane's piece of the tile (16 keys): maximum, sum of exponentials in pass 1's order, runner-up, the winner's key
    float a1v = -__builtin_inff(), a2v = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      a2v = __builtin_amdgcn_fmed3f(a1v, a2v, cs[i]);
      a1v = fmaxf(a1v, cs[i]);
    }
    int rr = 15;
#pragma unroll
    for (int i = 14; i >= 0; --i) rr = (cs[i] == a1v) ? i : rr;   // lowest register = lowest key
    float ts = __builtin_amdgcn_exp2f(cs[0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) ts += __builtin_amdgcn_exp2f(cs[i]);
    // per query: L_q and the tile; per lane half: {piece maximum, piece sum, runner-up, key of the maximum} — pass 1 meets the
    // tile again and takes its pieces from here instead of fetching the rows a second time (4 of 5 flagged items on planted
    // data), and its row recovery finds the winner already known when the final maximum sits in this tile
    const float L = fmaxf(a1v, __shfl_xor(a1v, 32, 64));
    if (q < P) {
      float* o = lower + (size_t)q * kLowStride;
      if (h == 0) { o[0] = L; o[1] = __int_as_float(T); }
      o[2 + 4 * h] = a1v;
      o[3 + 4 * h] = ts;
      o[4 + 4 * h] = a2v;
      o[5 + 4 * h] = __int_as_float(T + 4 * h + (rr & 3) + 8 * (rr >> 2));
    }
  }
}

// ------------------------------------------------------------------------------------------ pass 1: screen + exact pieces
// Everything that reaches sums, maxima and indices comes from bf16 MFMA chains with C = 0 (the logits of the dense kernels).
// ws.hand: a workgroup whose first stage is mostly flagged marks its 256-query block(s) there and leaves; the dense tile-skip
// kernel behind this one owns the marked blocks (same rule, same logits: the same bits).
constexpr int kHandMinStages = 8;                             // key ranges shorter than this many stages (2 048 keys) are never handed over
constexpr int kQB1 = ISR_Q6_QB1;                              // 32-query blocks per wave in pass 1
constexpr int kQPB1 = kWaves * kQB1 * 32;                      // queries per workgroup: kNB1 of the dense kernels' 256-query blocks
constexpr int kNB1 = kQPB1 / 256;
__global__ __launch_bounds__(kThreads, ISR_Q6_W1) void corr_fp6_sparse_kernel(const uint8_t* __restrict__ Q6, const uint8_t* __restrict__ K6,
                                                                      const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K, int P, int N,
                                                                      int ldq, int ldk, const float* __restrict__ qnrm, const uint32_t* __restrict__ kmax,
                                                                      CorrWs ws, int32_t* __restrict__ idx_out, float* __restrict__ logp_out,
                                                                      float* __restrict__ lse_out) {
  __shared__ uint4 lds[2 * kQ6Chunks];
  __shared__ int bad_half[2];
  __shared__ int hand_cnt;
  if (gated_off(ws)) return;
  constexpr int QB = kQB1, NFR = 4, DEFF = 64;
  static_assert(kQPB1 % 256 == 0 && kNB1 <= 2, "a workgroup covers whole 256-query blocks of the fallback / finalize kernels");
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.x * kWaves + wave) * (QB * 32);
  const bool probe = blockIdx.x == 0;
  const long long t_sclk0 = probe ? (long long)__builtin_amdgcn_s_memtime() : 0;
  const long long t_ref0 = probe ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  if (tid < 2) bad_half[tid] = 0;
  if (tid == 0) hand_cnt = 0;

  auto load_q = [&](int qb, bf16x8 (&dst)[NFR]) __attribute__((always_inline)) {            // the bf16 fragments of query block qb (its rows are L2-warm)
    const uint16_t* src = Q + (size_t)min(q0 + qb * 32 + r, P - 1) * ldq + 8 * h;
#pragma unroll
    for (int s = 0; s < NFR; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  };
  i32x8 bq6[QB];
  float qn2[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    bf16x8 t[NFR];
    load_q(qb, t);
    bq6[qb] = load_q6(Q6, min(q0 + qb * 32 + r, P - 1), h);
    float n2 = 0.f;
#pragma unroll
    for (int s = 0; s < NFR; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = elem_f32<false>((uint16_t)t[s][e]);
        n2 = __builtin_fmaf(v, v, n2);
      }
    n2 += __shfl_xor(n2, 32, 64);
    qn2[qb] = n2;
  }
  // a workgroup of zero vectors (padding rows of a capacity-sized batch): every logit 0, every piece kept (0 >= 0 - T), each
  // chunk sum the number of its keys — corr_finish's inputs without touching the keys (as corr_bf16_direct_kernel does)
  {
    bool nonzero = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) nonzero |= qn2[qb] != 0.f;
    if (!__syncthreads_or(nonzero ? 1 : 0)) {
      const float kn2z = kn2_max(ws);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + qb * 32 + r;
        if (h == 0 && q < P) corr_finish<1>(q, 0.f, 0.f, 0, false, (double)N, 0.0, DEFF, 0.f, 0.f, kn2z, ws, idx_out, logp_out, lse_out);
      }
      if (tid < kNB1 && (kNB1 * blockIdx.x + tid) * 256 < P) ws.flags[kNB1 * blockIdx.x + tid] = 0;
      if (probe && tid == 0) { ws.clk[0] = 0; ws.clk[1] = 0; }
      return;
    }
  }
  // thresholds: thr_x = L_q - T decides what a piece's exact maximum must reach to enter the sum; thr_s = thr_x - E_q what its
  // approximate maximum must stay below for the piece to be skipped unseen
  const float T = ws.skip_T;                                  // screen_T(N), the same number the dense tile-skip kernel is given
  float thr_x[QB], thr_s[QB];
  int tstar[QB];                                              // pass 0's tile of the query, and this lane's two exact values there
  float pt[QB], pts[QB];
  {
    const float dk2 = __uint_as_float(kmax[0]), kt2 = __uint_as_float(kmax[1]);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int q = min(q0 + qb * 32 + r, P - 1);
      const float* lo = ws.lower + (size_t)q * kLowStride;
      const float L = lo[0];
      tstar[qb] = __float_as_int(lo[1]);
      pt[qb] = lo[2 + 4 * h];
      pts[qb] = lo[3 + 4 * h];
      thr_x[qb] = L - T;
      thr_s[qb] = thr_x[qb] - screen_error(qnrm[2 * q], qnrm[2 * q + 1], dk2, kt2) - 2e-6f * fabsf(L);
#if defined(ISR_ABL_P1) && (ISR_ABL_P1 & 1)     // timing-only ablation: nothing fails the screen
      thr_s[qb] = __builtin_inff();
#endif
    }
  }
  float sm[QB], sm2[QB], sl[QB];                              // per lane: largest / second largest piece maximum, the chunk's sum
  int stb[QB];                                                // the first tile that reached sm
  double sL[QB];                                              // the canonical f64 sum of the chunk sums
  bool over[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { sm[qb] = -__builtin_inff(); sm2[qb] = -__builtin_inff(); sl[qb] = 0.f; stb[qb] = 0; sL[qb] = 0.0; over[qb] = false; }

  const int nstage = (N + kTKQ - 1) / kTKQ;
  constexpr int CST = kChunk / kTKQ;       // stages per canonical chunk
  static_assert(kChunk % kTKQ == 0, "a canonical chunk is a whole number of stages");
  Q6Stream ks;
  ks.init(K6, 0, N, lds, tid, wave);
  ks.gload(0, 0);
  if (nstage > 1) ks.gload(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int redone = 0;                                            // items this wave redid exactly (wave-uniform)
  const int nfull = N / kTKQ;
  // Flat logits: when more than a quarter of the block's FIRST stage's items would have to be fetched and redone, the screen is
  // not paying for itself; the block goes to the dense tile-skip kernel behind this one, which forms the same pieces from the
  // same logits by the same rule.  A speed decision only: the bits do not depend on it.  (A plain loop over the stage ahead of
  // the main loop — 1.3 % of the screening done twice — so that the main loop keeps a single exit.)
  if (nstage >= kHandMinStages) {                             // block-uniform
    int marked = 0;
#pragma nounroll
    for (int sub = 0; sub < kTKQ / 32; ++sub) {
      const i32x8 a = ks.frag(0, sub, r, h);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float t6 = tile_max(mfma_fp6(a, bq6[qb]));
        marked += __builtin_amdgcn_ballot_w64(t6 >= thr_s[qb] && sub * 32 != tstar[qb]) != 0ull ? 1 : 0;
      }
    }
    if (lane == 0) atomicAdd(&hand_cnt, marked);
    __syncthreads();
    if (hand_cnt * 4 > kWaves * (kTKQ / 32) * QB) {
      if (tid < kNB1 && (kNB1 * blockIdx.x + tid) * 256 < P) ws.hand[kNB1 * blockIdx.x + tid] = 1;
      if (tid == 0) atomicAdd(&ws.redone[1], (unsigned long long)kNB1);
      return;                                                 // (nothing of this wave's is in flight: vmcnt(0) above)
    }
  }
  // An item that fails the screen is only MARKED (one scalar instruction: its bit in the stage's item mask); the marked items of
  // a stage are redone behind the stage's screening loop by ONE copy of the heavy code (inlined at each of the 16 unrolled
  // sites, the heavy path made the kernel 36 KB of code and every visit to it a run of instruction-cache misses: ~5 500
  // cycles per item).  The rows of marked item j + 1 are requested before item j's exponentials are taken.
  bf16x8 ka[NFR], qa[NFR];                                   // the current item's key and query fragments
  unsigned heavy = 0u, star = 0u;                            // wave-uniform: bit (sub * QB + qb) of the current stage's marked items
  auto request = [&](int qb, int kb) __attribute__((always_inline)) {
    int row = kb + r;
    row = row < N ? row : N - 1;
    const uint16_t* src = K + (size_t)row * ldk + 8 * h;
#pragma unroll
    for (int s = 0; s < NFR; ++s) ka[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
    load_q(qb, qa);
  };
  constexpr unsigned kItemsOfQb0 = QB == 1 ? 0xFFFFFFFFu : QB == 2 ? 0x55555555u : QB == 4 ? 0x11111111u : 0u;   // bits j with j % QB == 0
  static_assert(kItemsOfQb0 != 0u, "QB is 1, 2 or 4");
  // The marked items of the stage, in ITEM order (= tile order per query block): a piece enters its lane's sum and maxima at the
  // position the dense kernel gives it, whichever way it is obtained — `star` items (every flagged lane sits in its query's
  // pass-0 tile) from the stored pieces, `heavy` items by fetching the rows.  (Round-5 stress test: with the stored pieces applied
  // inline and the fetched ones behind the stage, a zero query's "first tile reaching the maximum" came out as its LAST tile.)
  auto redo_marked = [&](int kbase) __attribute__((always_inline)) {
    unsigned all = heavy | star;
    if (all == 0u) return;
    if (heavy != 0u) { const int j0 = __builtin_ctz(heavy); request(j0 % QB, kbase + (j0 / QB) * 32); }
    while (all != 0u) {                                      // wave-uniform
      const int j = __builtin_ctz(all);
      all &= all - 1u;
      const int qbj = j % QB, kb = kbase + (j / QB) * 32;
      if ((heavy >> j) & 1u) {
        ++redone;
        f32x16 c = splat16(0.f);
#pragma unroll
        for (int s = 0; s < NFR; ++s) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s], qa[s], c, 0, 0, 0);
        heavy &= heavy - 1u;                                 // (j is heavy's lowest bit: heavy items are met in order)
        if (heavy != 0u) {                                   // the next fetched item's rows travel under this one's epilogue
          const int jn = __builtin_ctz(heavy);
          request(jn % QB, kbase + (jn / QB) * 32);
        }
        if (kb + 32 > N) mask_tail(c, kb + 4 * h, N);
        const float t = tile_max(c);
#if defined(ISR_ABL_P1) && (ISR_ABL_P1 & 2)     // timing-only ablation: no exponentials
        float ts = c[0];
#else
        float ts = __builtin_amdgcn_exp2f(c[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) ts += __builtin_amdgcn_exp2f(c[i]);
#endif
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          if (qbj == qb) {                                   // wave-uniform
            sl[qb] += (t >= thr_x[qb]) ? ts : 0.f;            // the canonical rule, per lane (16 keys of the tile)
            sm2[qb] = __builtin_amdgcn_fmed3f(sm[qb], sm2[qb], t);
            stb[qb] = (t > sm[qb]) ? kb : stb[qb];
            sm[qb] = fmaxf(sm[qb], t);
          }
      } else {
        // the exact values of the lanes whose query has this tile as its pass-0 tile are at hand (the same chain, the same
        // order of the sum); every other lane was shown by the screen to hold nothing that counts here
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          if (qbj == qb) {                                   // wave-uniform
            if (kb == tstar[qb]) {
              sl[qb] += (pt[qb] >= thr_x[qb]) ? pts[qb] : 0.f;
              sm2[qb] = __builtin_amdgcn_fmed3f(sm[qb], sm2[qb], pt[qb]);
              stb[qb] = (pt[qb] > sm[qb]) ? kb : stb[qb];
              sm[qb] = fmaxf(sm[qb], pt[qb]);
            }
          }
      }
    }
    star = 0u;
  };
  int kbase_cur = 0;
  auto consume = [&](const f32x16& c6, int qb, int kb) __attribute__((always_inline)) {
    const float t6 = tile_max(c6);
    const bool f = t6 >= thr_s[qb];
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(f) != 0ull, 0)) {     // wave-uniform
      const unsigned bit = 1u << (((kb - kbase_cur) >> 5) * QB + qb);
#if defined(ISR_ABL_P1) && (ISR_ABL_P1 & 8)     // timing-only ablation: every flagged item is treated as a pass-0 tile
      star |= bit;
#else
      if (__builtin_amdgcn_ballot_w64(f && kb != tstar[qb]) == 0ull) {
        // every flagged lane sits in its query's pass-0 tile: the stored pieces serve.  At once when no earlier item of this
        // query block waits for the handler (then this IS its position in the order), behind the waiting ones otherwise.
        if (((heavy | star) & (kItemsOfQb0 << qb)) == 0u) {
          if (kb == tstar[qb]) {
            sl[qb] += (pt[qb] >= thr_x[qb]) ? pts[qb] : 0.f;
            sm2[qb] = __builtin_amdgcn_fmed3f(sm[qb], sm2[qb], pt[qb]);
            stb[qb] = (pt[qb] > sm[qb]) ? kb : stb[qb];
            sm[qb] = fmaxf(sm[qb], pt[qb]);
          }
        } else {
          star |= bit;
        }
      } else {
        heavy |= bit;
      }
#endif
    }
  };
  static_assert((kTKQ / 32) * QB <= 32, "the stage's item mask is one 32-bit word");
  for (int stage = 0; stage < nstage; ++stage) {
    const int buf = stage & 1;
    kbase_cur = stage * kTKQ;
    if (stage < nfull) q6_stage<QB>(ks, buf, stage * kTKQ, r, h, bq6, consume);
    else q6_stage_partial<QB>(ks, buf, stage * kTKQ, N, r, h, bq6, consume);
    redo_marked(stage * kTKQ);
    if ((stage + 1) % CST == 0 || stage + 1 == nstage) {      // the end of a canonical chunk: its sum leaves the registers
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float lc = sl[qb] + __shfl_xor(sl[qb], 32, 64);
        over[qb] |= !(lc <= 3.0e38f);
        sL[qb] += (double)lc;
        sl[qb] = 0.f;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (stage + 2 < nstage) ks.gload(stage + 2, buf);
  }

  // ---- row recovery (as corr_bf16_direct_kernel): the winner's row inside the first tile that reached the maximum
  bool any_bad_lane = false;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float mo = __shfl_xor(sm[qb], 32, 64);
    const int tbo = __shfl_xor(stb[qb], 32, 64);
    const float M = fmaxf(sm[qb], mo);
    const int Tt = (sm[qb] == M) ? ((mo == M) ? min(stb[qb], tbo) : stb[qb]) : tbo;
    int cand = Tt;
    float cmax = M, c2 = -__builtin_inff();
    if (idx_out) {
      bf16x8 bq[NFR];
      load_q(qb, bq);
      // the usual case: the final maximum sits in pass 0's tile, whose winner and runner-up pass 0 left per lane half
      const bool known = Tt == tstar[qb];
      {
        const float* lo = ws.lower + (size_t)min(q0 + qb * 32 + r, P - 1) * kLowStride;
        cmax = known ? pt[qb] : -__builtin_inff();
        c2 = known ? lo[4 + 4 * h] : -__builtin_inff();
        cand = known ? __float_as_int(lo[5 + 4 * h]) : Tt;
      }
      unsigned long long todo = __ballot(!known);
      while (todo) {                                          // wave-uniform trip count (<= 32; usually 0)
        const int kb = __shfl(Tt, __ffsll(todo) - 1, 64);
        todo &= ~__ballot(Tt == kb);
        const f32x16 c = exact_tile(K, N, ldk, kb, r, h, bq);
        if (Tt == kb) {
          float a1v = -__builtin_inff(), a2v = -__builtin_inff();
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            a2v = __builtin_amdgcn_fmed3f(a1v, a2v, c[i]);
            a1v = fmaxf(a1v, c[i]);
          }
          cmax = a1v; c2 = a2v;
          int rr = 15;
#pragma unroll
          for (int i = 14; i >= 0; --i) rr = (c[i] == cmax) ? i : rr;
          cand = kb + 4 * h + (rr & 3) + 8 * (rr >> 2);
        }
      }
    }
    const int co = __shfl_xor(cand, 32, 64);
    const float cmo = __shfl_xor(cmax, 32, 64);
    const bool other_wins = better(cmo, co, cmax, cand);
    const float mine = other_wins ? sm[qb] : fmaxf(sm2[qb], c2);
    const float run = fmaxf(mine, __shfl_xor(mine, 32, 64));
    if (other_wins) { cand = co; cmax = cmo; }
    const bool bad_q = over[qb] || !(cmax >= kLow);
    const int q = q0 + qb * 32 + r;
    sm[qb] = cmax; sm2[qb] = run; stb[qb] = cand;
    over[qb] = bad_q;
    any_bad_lane |= bad_q && q < P;
  }
  // the bad-query protocol of the dense kernels, per 256-query block (kNB1 = 2: waves 0, 1 hold block 2 b, waves 2, 3 block 2 b + 1)
  const int myblk = (wave * QB * 32) / 256;
  if (__any(any_bad_lane) && lane == 0) atomicOr(&bad_half[myblk], 1);
  __syncthreads();
  const int any_bad = bad_half[myblk];
  if (tid < kNB1 && (kNB1 * blockIdx.x + tid) * 256 < P) {
    const int ent = kNB1 * blockIdx.x + tid;
    ws.flags[ent] = bad_half[tid];
    if (bad_half[tid]) ws.blist[atomicAdd(&ws.rcount[1], 1)] = ent;
  }
  if (lane == 0 && redone) atomicAdd(ws.redone, (unsigned long long)redone);     // diagnostics: tile items redone exactly
  if (probe && tid == 0) {
    ws.clk[0] = (long long)__builtin_amdgcn_s_memtime() - t_sclk0;
    ws.clk[1] = (long long)__builtin_amdgcn_s_memrealtime() - t_ref0;
  }
  const float kn2 = kn2_max(ws);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q0 + qb * 32 + r;
    if (h != 0 || q >= P) continue;
    if (any_bad) {
      ws.pbad[q] = over[qb] ? 1 : 0;
      if (over[qb]) ws.qn2[q] = qn2[qb];
    }
    if (!over[qb])
      corr_finish<1>(q, sm[qb], sm2[qb], stb[qb], false, sL[qb], 0.0, DEFF, 0.f, qn2[qb], kn2, ws, idx_out, logp_out, lse_out);
  }
}
