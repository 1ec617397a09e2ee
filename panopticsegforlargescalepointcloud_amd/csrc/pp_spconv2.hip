// K4 (pipelined): the dense-offset sparse convolution with the memory latency taken off the MFMA critical path.
//
// Same contract, tiling and summation order as k_spconv_fwd (pp_spconv.hip): a wave owns 32 output rows x NTW*16 output
// channels, A fragments are gathered straight from the rows named by the kernel map, B fragments come pre-packed.
// What changed (measured on the un-pipelined kernel, profiles/r01_g_pmc_dense64.md: MFMA pipe busy 48 %, 37 % of wave
// cycles parked in s_waitcnt, every wave of a CU waiting for its loads at the same time):
//   * prologue: the wave's 27 x 32 neighbour indices are fetched with 14 back-to-back coalesced loads and parked in
//     LDS (3.5 KiB per wave, wave-private, no barrier); the same pass builds two 27-bit scalar masks of the offsets
//     that are occupied for the upper / lower 16-row tile;
//   * the main loop walks only the occupied offsets (s_ff1 on the mask) -- no index loads, no empty iterations;
//   * software pipeline of depth 1 in registers: the A gathers and B fragment loads of step n+1 are issued before the
//     MFMAs of step n, so each wave overlaps its own memory latency with its own matrix work instead of relying on
//     other waves being out of phase.
#include <stdlib.h>

#include "pp_spconv.h"

#define F2_MAXK 28

template <int NTW, int T>
__global__ __launch_bounds__(256, 2) void k_spconv_fwd2(SpconvArgs a) {
  constexpr int R = 16 * T;  // rows per wave
  __shared__ int32_t s_idx[4][F2_MAXK][R];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const unsigned bid = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row_base = ((int64_t)bid * 4 + wave) * R;
  if (row_base >= a.n_out) return;  // wave-uniform; waves never synchronise
  const int jt0 = blockIdx.y * NTW;
  int32_t(*idx)[R] = s_idx[wave];

  // ---- prologue: indices -> LDS, per-tile occupancy masks -> SGPRs
  unsigned m[T];
#pragma unroll
  for (int tt = 0; tt < T; ++tt) m[tt] = 0;
  {
    constexpr int KPL = 64 / R >= 1 ? 64 / R : 1;       // offsets fetched per load instruction (R = 32: 2, R = 64: 1)
    constexpr int NL = (F2_MAXK + KPL - 1) / KPL;       // load instructions
    const int rr = lane % R, kh = lane / R;
    const int64_t row = row_base + rr;
    const bool rv = row < a.n_out;
    int v[NL];
    // branch-free: clamped addresses, validity applied afterwards (a branch per load would serialise the loads)
    const int64_t rowc = rv ? row : a.n_out - 1;
    if (a.nbr) {
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = KPL * kk + kh;
        const int kc = k < a.K ? k : a.K - 1;
        v[kk] = a.nbr[(int64_t)kc * a.n_out + rowc];
      }
#pragma unroll
      for (int kk = 0; kk < NL; ++kk)
        if (!rv || KPL * kk + kh >= a.K) v[kk] = -1;
    } else {
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) v[kk] = (rv && KPL * kk + kh < a.K) ? (int)row : -1;
    }
#pragma unroll
    for (int kk = 0; kk < NL; ++kk) {
      if (KPL * kk + kh < F2_MAXK) idx[KPL * kk + kh][rr] = v[kk];
      const unsigned long long b = __ballot(v[kk] >= 0);
#pragma unroll
      for (int h = 0; h < KPL; ++h)
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
          m[tt] |= (((b >> (h * R + tt * 16)) & 0xFFFFull) ? 1u : 0u) << (KPL * kk + h);
    }
  }
  unsigned rem = 0;
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    m[tt] = __builtin_amdgcn_readfirstlane(m[tt]);
    rem |= m[tt];
  }

  f32x4 acc[T][NTW];
#pragma unroll
  for (int tt = 0; tt < T; ++tt)
#pragma unroll
    for (int jt = 0; jt < NTW; ++jt) acc[tt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int cin = a.c0 + a.c1;
  const int S0 = a.c0 >> 4, S = cin >> 4;
  const float* wl = a.wp + (int64_t)jt0 * 256 + lane * 4;

  if (rem) {
    int k = __builtin_ctz(rem);
    int s = 0;
    int ar[T];
#pragma unroll
    for (int tt = 0; tt < T; ++tt) ar[tt] = idx[k][tt * 16 + i];
    f32x4 A[T], B[NTW];
    // loads of step (KK, SS) with neighbour rows RR[] into (X[], Y[]): unconditional, clamped addresses
#define F2_LOAD(X, Y, KK, SS, RR)                                                           \
  {                                                                                         \
    const float* src_ = (SS) < S0 ? a.in0 : a.in1;                                          \
    const int cs_ = (SS) < S0 ? a.c0 : a.c1;                                                \
    const int ss_ = (SS) < S0 ? (SS) : (SS)-S0;                                             \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) {                                      \
      const int r_ = RR[tt] < 0 ? 0 : RR[tt];                                               \
      X[tt] = *(const f32x4*)(src_ + (int64_t)r_ * cs_ + ss_ * 16 + q * 4);                 \
    }                                                                                       \
    const float* w_ = wl + ((int64_t)(KK)*S + (SS)) * a.NT * 256;                           \
    _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt) {                                    \
      const int jc_ = jt0 + jt < a.NT ? jt : 0;                                             \
      Y[jt] = *(const f32x4*)(w_ + jc_ * 256);                                              \
    }                                                                                       \
  }
#define F2_ZERO(X, RR) \
  _Pragma("unroll") for (int tt = 0; tt < T; ++tt) X[tt] = RR[tt] >= 0 ? X[tt] : (f32x4){0.f, 0.f, 0.f, 0.f};
    F2_LOAD(A, B, k, s, ar);
    F2_ZERO(A, ar);
    for (;;) {
      // next step
      int kn = k, sn = s + 1;
      int arn[T];
#pragma unroll
      for (int tt = 0; tt < T; ++tt) arn[tt] = ar[tt];
      if (sn == S) {
        sn = 0;
        rem &= rem - 1;
        kn = rem ? __builtin_ctz(rem) : -1;
        if (kn >= 0) {
#pragma unroll
          for (int tt = 0; tt < T; ++tt) arn[tt] = idx[kn][tt * 16 + i];
        }
      }
      // loads of the next step are unconditional (the last iteration re-reads its own step) so that no branch or
      // select sits between them and the MFMAs below: their s_waitcnt lands at the register rotation after the MFMAs
      f32x4 An[T], Bn[NTW];
      {
        const int kl = kn >= 0 ? kn : k, sl = kn >= 0 ? sn : s;
        F2_LOAD(An, Bn, kl, sl, arn);
      }
      // matrix work of the current step.  One guarded block per 16-row tile (a block touches only its own accumulators:
      // the three-way both/upper/lower split made hipcc shuffle all accumulators through copies at every step)
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
        if ((m[tt] >> k) & 1u) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int jt = 0; jt < NTW; ++jt)
              acc[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt][t], B[jt][t], acc[tt][jt], 0, 0, 0);
        }
      }
      if (kn < 0) break;
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
        A[tt] = An[tt];
        ar[tt] = arn[tt];
      }
      F2_ZERO(A, ar);
#pragma unroll
      for (int jt = 0; jt < NTW; ++jt) B[jt] = Bn[jt];
      k = kn;
      s = sn;
    }
#undef F2_LOAD
#undef F2_ZERO
  }

  // epilogue: lane (col = i, row group = q) holds rows 4q+r of each 16-row tile
#pragma unroll
  for (int jt = 0; jt < NTW; ++jt) {
    const int col = (jt0 + jt) * 16 + i;
    if (jt0 + jt < a.NT && col < a.cout) {
      const float sc = a.scale ? a.scale[col] : 1.f;
      const float sh = a.shift ? a.shift[col] : 0.f;
#pragma unroll
      for (int rt = 0; rt < T; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row_base + rt * 16 + q * 4 + r;
          if (row < a.n_out) {
            float v = acc[rt][jt][r] * sc + sh;
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.residual) v += a.residual[row * a.cout + col];
            a.out[row * a.cout + col] = v;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// v3: the same pipeline with (almost) no vector-ALU instruction left in the main loop.
//
// Measured on gfx950 (profiles/microbench/mfma_peak.hip): VALU instructions do NOT overlap with MFMAs of the same SIMD,
// not even from other waves -- 16 MFMAs + 64 v_fma per iteration run at 107 TFLOP/s instead of 155, at 1, 2 or 4 waves
// per SIMD alike.  v2 spends ~100 VALU instructions per step on 64-bit address arithmetic, zero-selects for missing
// neighbours and register rotation, which caps it at ~50 % of the MFMA peak even on a fully occupied, L2-resident map.
// Here
//   * A gathers and B fragment loads are buffer loads: 32-bit per-lane byte offset + a scalar descriptor whose base
//     carries the channel step, so a load costs no VALU instruction at all;
//   * a missing neighbour is the byte offset 0xFFFFFFFF: the hardware bounds check returns zeros, no select;
//   * LDS holds the byte offsets (row * row_bytes) computed once in the prologue; a new offset costs one v_or per tile;
//   * the step loop is unrolled by two over ping-pong register sets, so there is no rotation copy;
//   * everything else (step bookkeeping, occupancy tests, descriptor updates) is scalar.
// Requires n_in * cin_per_source * 4 < 4 GiB and c1 == 0 or c1 == c0 (the caller falls back to v2 otherwise).
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define F3_MISSING 0xFFFFFFFFu
// F3_ABLATE (profiling builds only, profiles/ablate_conv.sh): 1 = no step loop (prologue + epilogue), 2 = loads but no
// MFMAs, 3 = no feature gathers, 4 = no weight loads.  Results are wrong by construction; never defined in the product.
#ifndef F3_ABLATE
#define F3_ABLATE 0
#endif
// waves per workgroup of k_spconv_fwd3 (waves never synchronise; the workgroup is only the dispatch granule)
#ifndef F3_WPB
#define F3_WPB 4
#endif

template <int NTW, int T, bool BF16, int D>
__global__ __launch_bounds__(64 * F3_WPB, 2) void k_spconv_fwd3(SpconvArgs a, unsigned a_bytes, unsigned w_bytes) {
  constexpr int R = 16 * T;  // rows per wave
  __shared__ unsigned s_off[F3_WPB][F2_MAXK][R];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const unsigned bid = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row_base = ((int64_t)bid * F3_WPB + wave) * R;
  if (row_base >= a.n_out) return;  // wave-uniform; waves never synchronise
  const int jt0 = blockIdx.y * NTW;
  unsigned(*off)[R] = s_off[wave];
  const unsigned row_bytes = (unsigned)a.c0 * 4u;

  // ---- prologue: neighbour rows -> byte offsets in LDS, per-tile occupancy masks -> SGPRs
  unsigned m[T];
  {
    // lanes (kh, rr): rr = row slot of the wave, kh = which half of the offsets (KPL = 2 halves when a wave has 32 rows).
    // Offsets are split in contiguous halves (k = kk + NL * kh), every lane keeps one presence bit per kk, a 4-step
    // OR over the 16 lanes of a row tile and T * KPL readlanes give the per-tile masks -- the earlier form decoded
    // every ballot with ~6 scalar instructions per (offset, tile), a quarter of the wave's non-MFMA issue slots
    constexpr int KPL = 64 / R >= 1 ? 64 / R : 1;
    constexpr int NL = (F2_MAXK + KPL - 1) / KPL;
    const int rr = lane % R, kh = lane / R;
    const bool rv = row_base + rr < a.n_out;
    // tile schedule: the wave's 32 "slots" may name any output rows (rows with similar neighbour masks are scheduled
    // into the same tile by the coordinate manager); without one, slot = row
    const int64_t slot = rv ? row_base + rr : a.n_out - 1;
    const int64_t row = a.row_order ? (int64_t)a.row_order[slot] : slot;
    int v[NL];
    const int64_t rowc = row;
    if (a.nbr) {
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = kk + NL * kh;
        const int kc = k < a.K ? k : a.K - 1;
        v[kk] = a.nbr[(int64_t)kc * a.n_out + rowc];
      }
#pragma unroll
      for (int kk = 0; kk < NL; ++kk)
        if (!rv || kk + NL * kh >= a.K) v[kk] = -1;
    } else {
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) v[kk] = (rv && kk + NL * kh < a.K) ? (int)row : -1;
    }
    unsigned ml = 0;
#pragma unroll
    for (int kk = 0; kk < NL; ++kk) {
      const int k = kk + NL * kh;
      if (k < F2_MAXK) off[k][rr] = v[kk] >= 0 ? (unsigned)v[kk] * row_bytes : F3_MISSING;
      ml |= (v[kk] >= 0 ? 1u : 0u) << kk;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) ml |= (unsigned)__shfl_xor((int)ml, o);
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
      m[tt] = 0;
#pragma unroll
      for (int h = 0; h < KPL; ++h)
        m[tt] |= (unsigned)__builtin_amdgcn_readlane((int)ml, h * R + tt * 16) << (NL * h);
    }
  }
  unsigned rem = 0;
  unsigned kmask = 0xFFFFFFFFu;
  if (a.split > 1) {
    const int k0 = (int)blockIdx.z * a.K / a.split, k1 = ((int)blockIdx.z + 1) * a.K / a.split;
    kmask = (k1 >= 32 ? 0xFFFFFFFFu : (1u << k1) - 1u) & ~((1u << k0) - 1u);
  }
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    m[tt] = __builtin_amdgcn_readfirstlane(m[tt]) & kmask;
    rem |= m[tt];
  }

  f32x4 acc[T][NTW];
#pragma unroll
  for (int tt = 0; tt < T; ++tt)
#pragma unroll
    for (int jt = 0; jt < NTW; ++jt) acc[tt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (rem && F3_ABLATE != 1) {
    const int S0 = a.c0 >> 4, S = (a.c0 + a.c1) >> 4;
    const unsigned q16 = (unsigned)q * 16u;
    const unsigned lane16 = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (int)w_bytes, 0x00020000);
    const unsigned w_step = (unsigned)a.NT * 1024u;  // bytes of packed weights per (k, s)
    const unsigned w_jt0 = (unsigned)jt0 * 1024u;

    // load side runs one step ahead of the compute side.  Its state (rem, kl, sl) is scalar; descriptor base and weight
    // offset are recomputed from it every step (carrying them across iterations makes hipcc treat them as divergent and
    // emit waterfall loops around the buffer loads).  Scalar instructions are not free either: they take MFMA issue
    // slots (profiles/microbench/mfma_peak.hip), so the bookkeeping is kept short.
    int kl = __builtin_ctz(rem), sl = 0;
    unsigned vo[T];
#pragma unroll
    for (int tt = 0; tt < T; ++tt) vo[tt] = off[kl][tt * 16 + i] | q16;
    f32x4 A0[T], B0[NTW], A1[T], B1[NTW];

#define F3_LOADS(AX, BX)                                                                                       \
  {                                                                                                            \
    const float* src_ = sl < S0 ? a.in0 + sl * 16 : a.in1 + (sl - S0) * 16;                                    \
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)src_, 0, (int)a_bytes, 0x00020000); \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt)                                                           \
        AX[tt] = F3_ABLATE == 3 ? (f32x4){1.f, 2.f, 3.f, (float)vo[tt]}                                         \
                                : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)vo[tt], 0, 0)); \
    const unsigned wso_ = (unsigned)(kl * S + sl) * w_step + w_jt0;                                            \
    _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)                                                         \
        BX[jt] = F3_ABLATE == 4 ? (f32x4){1.f, 2.f, 3.f, (float)wso_}                                           \
                                : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(lane16 + jt * 1024u), (int)wso_, 0)); \
  }
    // advance the load side; VALID = false when no step is left (the state then still names a valid step)
#define F3_ADVANCE(VALID)                                             \
  {                                                                   \
    VALID = 1;                                                        \
    ++sl;                                                             \
    if (sl == S) {                                                    \
      sl = 0;                                                         \
      rem &= rem - 1;                                                 \
      if (rem) {                                                      \
        kl = __builtin_ctz(rem);                                      \
        _Pragma("unroll") for (int tt = 0; tt < T; ++tt) vo[tt] = off[kl][tt * 16 + i] | q16; \
      } else                                                          \
        VALID = 0;                                                    \
    }                                                                 \
  }
    // BF16: the same registers, rounded to bfloat16 (k = 4q + t of lane (i, q) is exactly the operand layout of
    // v_mfma_f32_16x16x16_bf16), so one MFMA replaces the four fp32 ones; fp32 accumulation either way
#define F3_MFMAS(AX, BX, KC)                                                                              \
  if constexpr (F3_ABLATE == 2) {                                                                         \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) asm volatile("" ::"v"(AX[tt]));                      \
    _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt) asm volatile("" ::"v"(BX[jt]));                    \
  } else if constexpr (BF16) {                                                                                   \
    s16x4 bh_[NTW];                                                                                       \
    _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt) bh_[jt] = pp_bf16x4(BX[jt]);                       \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) {                                                    \
      if ((m[tt] >> (KC)) & 1u) {                                                                         \
        const s16x4 ah_ = pp_bf16x4(AX[tt]);                                                              \
        _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)                                                \
            acc[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah_, bh_[jt], acc[tt][jt], 0, 0, 0);  \
      }                                                                                                   \
    }                                                                                                     \
  } else {                                                                                                \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) {                                                    \
      if ((m[tt] >> (KC)) & 1u) {                                                                         \
        _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)                                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                 \
                acc[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AX[tt][t], BX[jt][t], acc[tt][jt], 0, 0, 0); \
      }                                                                                                   \
    }                                                                                                     \
  }
    int more;  // an int, not a bool: hipcc keeps bools as 64-bit lane masks (4 scalar instructions per test)
    // The loads are unconditional: after the last step the load side simply re-reads a valid step.  (A branch around
    // them makes hipcc merge the two paths' outstanding-load counts and wait for the NEW loads before the MFMAs.)
    if constexpr (D == 1) {
      F3_LOADS(A0, B0);
      int kc = kl;
      F3_ADVANCE(more);
      for (;;) {
        F3_LOADS(A1, B1);
        F3_MFMAS(A0, B0, kc);
        if (!more) break;
        kc = kl;
        F3_ADVANCE(more);
        F3_LOADS(A0, B0);
        F3_MFMAS(A1, B1, kc);
        if (!more) break;
        kc = kl;
        F3_ADVANCE(more);
      }
    } else if constexpr (D == 3) {
      // as D == 1, but the load side advances (LDS read of the next offsets) BEFORE the MFMAs of the current step, so
      // that read is covered by them instead of sitting between two steps
      int k0, k1, e0, e1, nx;
      F3_LOADS(A0, B0);
      k0 = kl;
      F3_ADVANCE(nx);
      for (;;) {
        F3_LOADS(A1, B1);
        k1 = kl;
        e1 = nx;
        F3_ADVANCE(more);
        nx = e1 ? more : 0;
        F3_MFMAS(A0, B0, k0);
        if (!e1) break;
        F3_LOADS(A0, B0);
        k0 = kl;
        e0 = nx;
        F3_ADVANCE(more);
        nx = e0 ? more : 0;
        F3_MFMAS(A1, B1, k1);
        if (!e0) break;
      }
    } else {
      // two steps in flight (three register sets): narrow layers have too few MFMAs per step to cover a gather
      int nsteps = __builtin_popcount(rem) * S;
      f32x4 A2[T], B2[NTW];
      F3_LOADS(A0, B0);
      int k0 = kl, k1, k2;
      F3_ADVANCE(more);
      F3_LOADS(A1, B1);
      k1 = kl;
      F3_ADVANCE(more);
      for (;;) {
        F3_LOADS(A2, B2);
        k2 = kl;
        F3_ADVANCE(more);
        F3_MFMAS(A0, B0, k0);
        if (--nsteps == 0) break;
        F3_LOADS(A0, B0);
        k0 = kl;
        F3_ADVANCE(more);
        F3_MFMAS(A1, B1, k1);
        if (--nsteps == 0) break;
        F3_LOADS(A1, B1);
        k1 = kl;
        F3_ADVANCE(more);
        F3_MFMAS(A2, B2, k2);
        if (--nsteps == 0) break;
      }
    }
#undef F3_LOADS
#undef F3_ADVANCE
#undef F3_MFMAS
  }

  if (a.split > 1) {  // raw partial sums; the epilogue runs in k_spconv_split_reduce
    float* __restrict__ part = a.part + (int64_t)blockIdx.z * a.n_out * a.cout;
#pragma unroll
    for (int jt = 0; jt < NTW; ++jt) {
      const int col = (jt0 + jt) * 16 + i;
      if (jt0 + jt < a.NT && col < a.cout) {
#pragma unroll
        for (int rt = 0; rt < T; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t slot = row_base + rt * 16 + q * 4 + r;
            if (slot < a.n_out) {
              const int64_t row = a.row_order ? (int64_t)a.row_order[slot] : slot;
              part[row * a.cout + col] = acc[rt][jt][r];
            }
          }
      }
    }
    return;
  }
  // epilogue: lane (col = i, row group = q) holds rows 4q+r of each 16-row tile
#pragma unroll
  for (int jt = 0; jt < NTW; ++jt) {
    const int col = (jt0 + jt) * 16 + i;
    if (jt0 + jt < a.NT && col < a.cout) {
      const float sc = a.scale ? a.scale[col] : 1.f;
      const float sh = a.shift ? a.shift[col] : 0.f;
#pragma unroll
      for (int rt = 0; rt < T; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t slot = row_base + rt * 16 + q * 4 + r;
          if (slot < a.n_out) {
            const int64_t row = a.row_order ? (int64_t)a.row_order[slot] : slot;
            float v = acc[rt][jt][r] * sc + sh;
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.residual) v += a.residual[row * a.cout + col];
            a.out[row * a.cout + col] = v;
          }
        }
      }
    }
  }
}

template <int T, bool BF16>
static int launch3_t(const SpconvArgs& a, int ntw, unsigned groups, unsigned a_bytes, unsigned w_bytes, hipStream_t s) {
  dim3 grid(pp_blocks(a.n_out, 16 * T * F3_WPB), groups, (unsigned)(a.split > 1 ? a.split : 1));
  // PP_DENSE_DEPTH: loop variant for launches with <= PP_DENSE_DEPTH_NTW (default 2) column tiles per wave:
  // 1 = one step in flight, 2 = two steps in flight, 3 (default) = one step in flight with the load side advanced
  // before the MFMAs (16->16 at 2.5 M rows: 374 / 375 / 343 us); wider launches always use 1
  static int depth_env = -1;
  static int depth_ntw = 2;
  if (depth_env < 0) {
    depth_env = getenv("PP_DENSE_DEPTH") ? atoi(getenv("PP_DENSE_DEPTH")) : 3;
    if (getenv("PP_DENSE_DEPTH_NTW")) depth_ntw = atoi(getenv("PP_DENSE_DEPTH_NTW"));
  }
  const bool deep = depth_env == 2 && ntw <= depth_ntw;
  const bool early = depth_env == 3 && ntw <= depth_ntw;
  switch (ntw + (deep ? 10 : 0) + (early ? 20 : 0)) {
    case 21: hipLaunchKernelGGL((k_spconv_fwd3<1, T, BF16, 3>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 22: hipLaunchKernelGGL((k_spconv_fwd3<2, T, BF16, 3>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 23: hipLaunchKernelGGL((k_spconv_fwd3<3, T, BF16, 3>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 24: hipLaunchKernelGGL((k_spconv_fwd3<4, T, BF16, 3>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 11: hipLaunchKernelGGL((k_spconv_fwd3<1, T, BF16, 2>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 12: hipLaunchKernelGGL((k_spconv_fwd3<2, T, BF16, 2>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 13: hipLaunchKernelGGL((k_spconv_fwd3<3, T, BF16, 2>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 14: hipLaunchKernelGGL((k_spconv_fwd3<4, T, BF16, 2>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 1: hipLaunchKernelGGL((k_spconv_fwd3<1, T, BF16, 1>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 2: hipLaunchKernelGGL((k_spconv_fwd3<2, T, BF16, 1>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 3: hipLaunchKernelGGL((k_spconv_fwd3<3, T, BF16, 1>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    case 4: hipLaunchKernelGGL((k_spconv_fwd3<4, T, BF16, 1>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes); break;
    default: pp_set_error("pp_spconv_fwd3: ntw %d out of range", ntw); return PP_ERR_INVALID;
  }
  return PP_OK;
}

// sum of the split-K partials in a fixed order + the epilogue of the convolution (deterministic, one pass)
__global__ __launch_bounds__(256) void k_spconv_split_reduce(SpconvArgs a) {
  const int64_t total = a.n_out * a.cout;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int col = (int)(e % a.cout);
  float v = a.part[e];
  for (int z = 1; z < a.split; ++z) v += a.part[(int64_t)z * total + e];
  v = v * (a.scale ? a.scale[col] : 1.f) + (a.shift ? a.shift[col] : 0.f);
  if (a.relu) v = fmaxf(v, 0.f);
  if (a.residual) v += a.residual[e];
  a.out[e] = v;
}

int pp_spconv_split_reduce_launch(const SpconvArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_spconv_split_reduce, dim3(pp_blocks(a.n_out * a.cout, 256)), dim3(256), 0, s, a);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

bool pp_spconv_fwd3_ok(const SpconvArgs& a, int64_t n_in) {
  if (a.c1 != 0 && a.c1 != a.c0) return false;
  if (n_in <= 0) return false;
  return (double)n_in * a.c0 * 4.0 < 4294967000.0 && (double)a.K * (a.c0 + a.c1) * a.NT * 64.0 < 4294967000.0;
}

int pp_spconv_fwd3_launch(const SpconvArgs& a, int64_t n_in, int ntw, unsigned groups, hipStream_t s) {
  static int t_env = -1;
  if (t_env < 0) {
    const char* e = getenv("PP_DENSE_T");
    t_env = e ? atoi(e) : 0;
  }
  // rows per wave: 64 (T = 4) on launches with <= PP_DENSE_T4_NTW (default 2) column tiles per wave -- narrow layers are
  // bound by per-step work, which 64 rows halve per row (end to end 186.9 -> 183.7 ms) -- and 32 (T = 2) on wider ones,
  // where the extra accumulators cost occupancy (48->48: 660 vs 690 us), and on launches below PP_DENSE_T4_ROWS (2 M)
  // rows, where halving the number of waves costs more (one rank's share at 8 GPUs: 34.1 vs 34.6 ms); PP_DENSE_T forces
  // one value
  static const int t4_ntw = getenv("PP_DENSE_T4_NTW") ? atoi(getenv("PP_DENSE_T4_NTW")) : 2;
  static const int64_t t4_rows = getenv("PP_DENSE_T4_ROWS") ? atoll(getenv("PP_DENSE_T4_ROWS")) : 2000000;
  const int T = t_env ? t_env : (ntw <= t4_ntw && a.n_out >= t4_rows ? 4 : 2);
  const unsigned a_bytes = (unsigned)((uint64_t)n_in * a.c0 * 4u);
  const unsigned w_bytes = (unsigned)((uint64_t)a.K * ((a.c0 + a.c1) / 16) * a.NT * 1024u);
  if (a.bf16)
    return T == 4 ? launch3_t<4, true>(a, ntw, groups, a_bytes, w_bytes, s)
                  : launch3_t<2, true>(a, ntw, groups, a_bytes, w_bytes, s);
  return T == 4 ? launch3_t<4, false>(a, ntw, groups, a_bytes, w_bytes, s)
                : launch3_t<2, false>(a, ntw, groups, a_bytes, w_bytes, s);
}

template <int T>
static int launch_t(const SpconvArgs& a, int ntw, unsigned groups, hipStream_t s) {
  dim3 grid(pp_blocks(a.n_out, 64 * T), groups);
  switch (ntw) {
    case 1: hipLaunchKernelGGL((k_spconv_fwd2<1, T>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_spconv_fwd2<2, T>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_spconv_fwd2<3, T>), grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_spconv_fwd2<4, T>), grid, dim3(256), 0, s, a); break;
    default: pp_set_error("pp_spconv_fwd2: ntw %d out of range", ntw); return PP_ERR_INVALID;
  }
  return PP_OK;
}

// rows per wave: 32 (T = 2) or 64 (T = 4); PP_DENSE_T overrides the per-shape choice (A/B measurements)
int pp_spconv_fwd2_launch(const SpconvArgs& a, int ntw, unsigned groups, hipStream_t s) {
  static int t_env = -1;
  if (t_env < 0) {
    const char* e = getenv("PP_DENSE_T");
    t_env = e ? atoi(e) : 0;
  }
  const int T = t_env ? t_env : 2;
  return T == 4 ? launch_t<4>(a, ntw, groups, s) : launch_t<2>(a, ntw, groups, s);
}
