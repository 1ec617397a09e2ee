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
#include "pp_spconv.h"

#define F2_MAXK 28

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

// OR of a value over the 16 lanes of every row (a 16-row MFMA tile): four DPP rotations inside the row instead of four
// ds_bpermute round trips through the LDS crossbar (each with its own wait)
__device__ __forceinline__ unsigned pp_row_or16(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);  // row_ror:8
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);  // row_ror:4
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);  // row_ror:2
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);  // row_ror:1
  return v;
}
// F3_ABLATE (profiling builds only, profiles/ablate_conv.sh): 1 = no step loop (prologue + epilogue), 2 = loads but no
// MFMAs, 3 = no feature gathers, 4 = no weight loads.  Results are wrong by construction; never defined in the product.
#ifndef F3_ABLATE
#define F3_ABLATE 0
#endif
// waves per workgroup of k_spconv_fwd3 (waves never synchronise; the workgroup is only the dispatch granule)
#ifndef F3_WPB
#define F3_WPB 4
#endif

// C4: the 4-channel input layer (rows of 16 bytes).  A step is a GROUP of four kernel offsets: lane (i, q) loads the whole row
// of offset 4 g + q with one dwordx4; the MFMA's k index is the offset inside the group, its four issues are the channels.
// DS: fused 1x1 shortcut (SpconvArgs::ds_*): after the offset loop the wave multiplies ITS OWN rows of ds_in (no gather: a
// same-level map's output row is the input row) with the packed 1x1 weights into a second set of accumulators.
// S1: the input has ONE 16-channel step (c0 == 16, no second source): a step is a kernel offset, the source descriptor is
// loop-invariant and the step bookkeeping shrinks to the offset mask (the general loop carries (offset, channel step) state
// and rebuilds the descriptor every step: ~25 scalar instructions per step that the 16-channel layers -- 16 MFMAs per step
// at most -- do not hide).
template <int NTW, int T, bool BF16, int D, bool C4 = false, bool DS = false, bool S1 = false>
__global__ __launch_bounds__(64 * F3_WPB, 2) void k_spconv_fwd3(SpconvArgs a, unsigned a_bytes, unsigned w_bytes, unsigned flags) {
  constexpr int R = 16 * T;  // rows per wave
  __shared__ unsigned s_off[F3_WPB][F2_MAXK][R];
  // D == 6 (LDS-staged feature tiles): per wave a ring of RD6 steps x T tiles x 2 KiB (16 rows x 128 bytes)
  constexpr int RD6 = NTW <= 2 ? 3 : 2;
  __shared__ f32x4 s_ring[D == 6 ? F3_WPB * RD6 * T * 128 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const unsigned bid = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row_base = ((int64_t)bid * F3_WPB + wave) * R;
  if (row_base >= a.n_out) return;  // wave-uniform; waves never synchronise
  const int jt0 = blockIdx.y * NTW;
  unsigned(*off)[R] = s_off[wave];
  const unsigned row_bytes = (unsigned)a.c0 * 4u;
  // row * row_bytes: v_mul_lo_u32 runs at a quarter of the vector rate (on the pipe the fp32 MFMAs share); the 24-bit form
  // is full rate and exact while the input has fewer than 2^24 rows (flags bit 0, set by the launcher)
  const bool m24 = (flags & 1u) != 0u;
#define F3_ROWOFF(R_) (m24 ? __umul24((unsigned)(R_), row_bytes) : (unsigned)(R_) * row_bytes)

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
    // slot order of a cross-level map (pp_maporder.hip): the map is stored slot-major, row_order[slot] names the output
    // row the slot writes; same-level maps have slot = row and no row_order
    const int64_t slot = rv ? row_base + rr : a.n_out - 1;
    const int64_t row = a.row_order ? (int64_t)a.row_order[slot] : slot;
    int v[NL];
    const int64_t rowc = slot;
    unsigned ml = 0;
    if (a.t8 == 2) {
      // compact same-level map: the table is filled with MISSING (every lane its own half of the offsets, as below), then the
      // wave's present entries -- one contiguous run of the entry array -- are loaded 64 at a time and dropped into their
      // (offset, row) slots; LDS operations of a wave execute in program order.  ~31 bytes per row at 6.8 pairs per row
      // instead of the dense map's 108, and 4 batches of loads instead of 28 per lane.
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = kk + NL * kh;
        if (k < F2_MAXK) off[k][rr] = F3_MISSING;
      }
      const unsigned mrow = (rv && kh == 0) ? a.cm_mask[slot] : 0u;
      const int64_t chunks = (a.n_out + 31) >> 5;
      const int64_t c0 = row_base >> 5, c1 = c0 + R / 32 < chunks ? c0 + R / 32 : chunks;
      const int e0 = a.cm_start[c0], e1 = a.cm_start[c1];
      const unsigned r0 = (unsigned)row_base & 63u;
      for (int eb = e0; eb < e1; eb += 256) {
        int v4[4];
        unsigned t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = eb + u * 64 + lane;
          const int ec = e < e1 ? e : e1 - 1;  // (loads stay in range; the store below is predicated)
          v4[u] = a.nbr[ec];
          t4[u] = a.cm_tag[ec];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = eb + u * 64 + lane;
          const unsigned prod = m24 ? __umul24((unsigned)v4[u], row_bytes) : (unsigned)v4[u] * row_bytes;
          if (e < e1) off[t4[u] >> 6][(t4[u] & 63u) - r0] = prod;
        }
      }
      const unsigned mk = pp_row_or16(mrow);
#pragma unroll
      for (int tt = 0; tt < T; ++tt) m[tt] = (unsigned)__builtin_amdgcn_readlane((int)mk, tt * 16);
    } else if (a.t8) {
      // 8-wide transposed map: the lane's row has <= 8 neighbours, entry j (= coarse row | class << 28) belongs to the offset
      // the row's parity class names.
      // Every (k, row) slot is filled with MISSING first (each lane its own half of the offsets, as below), then one lane
      // per row overwrites the <= 8 real ones; LDS operations of a wave execute in program order.
      int e8[8];
      if (flags & 4u) {  // the 8-wide map fits 32-bit byte offsets: buffer loads, one scalar offset per entry
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)a.nbr, 0, (int)(8u * (unsigned)a.n_out * 4u), 0x00020000);
        const unsigned kstep = (unsigned)a.n_out * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j) e8[j] = __builtin_amdgcn_raw_buffer_load_b32(rn, (int)((unsigned)rowc * 4u), (int)(j * kstep), 0);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) e8[j] = a.nbr[(int64_t)j * a.n_out + rowc];
      }
      unsigned cls = 0;  // every entry of the row carries the row's parity class in bits 28..30
#pragma unroll
      for (int j = 0; j < 8; ++j) cls |= e8[j] >= 0 ? (unsigned)e8[j] >> 28 : 0u;
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = kk + NL * kh;
        if (k < F2_MAXK) off[k][rr] = F3_MISSING;
      }
      unsigned mk = 0;
      // (class, j) -> offset index: per axis 1 when the row is even there, else 0 / 2 by the entry's bit.  Entries that do not
      // exist (or belong to the other lane half) are written to the spare row 27 of the table: no branch per entry
      const unsigned kx = (cls & 1u) ? 0u : 1u, ky = (cls & 2u) ? 0u : 3u, kz = (cls & 4u) ? 0u : 9u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = rv && (((unsigned)j & ~cls) == 0u) && e8[j] >= 0;
        const unsigned k = kx + ((j & 1) ? 2u : 0u) * (cls & 1u) + ky + ((j & 2) ? 6u : 0u) * ((cls >> 1) & 1u) + kz +
                           ((j & 4) ? 18u : 0u) * ((cls >> 2) & 1u);
        const unsigned prod = m24 ? __umul24((unsigned)e8[j] & PP_ROW_MASK, row_bytes) : ((unsigned)e8[j] & PP_ROW_MASK) * row_bytes;
        off[(ok && kh == 0) ? k : 27u][rr] = (ok && kh == 0) ? prod : F3_MISSING;
        mk |= ok ? 1u << k : 0u;
      }
      mk = pp_row_or16(mk);
#pragma unroll
      for (int tt = 0; tt < T; ++tt) m[tt] = (unsigned)__builtin_amdgcn_readlane((int)mk, tt * 16);
    } else {
    if (a.nbr) {
      // buffer loads: per-lane 32-bit offset (row and this lane's half of the offsets) + a scalar offset per kk -- no 64-bit
      // vector address arithmetic in front of the 14 / 28 loads (flags bit 1: the map is smaller than 4 GiB, else plain loads)
      if (flags & 2u) {
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)a.nbr, 0, (int)((unsigned)a.K * (unsigned)a.n_out * 4u), 0x00020000);
        const unsigned vrow = ((unsigned)rowc + (unsigned)(NL * kh) * (unsigned)a.n_out) * 4u;
        const unsigned kstep = (unsigned)a.n_out * 4u;
#pragma unroll
        for (int kk = 0; kk < NL; ++kk) v[kk] = __builtin_amdgcn_raw_buffer_load_b32(rn, (int)vrow, (int)(kk * kstep), 0);  // beyond K: zeros, masked below
      } else {
#pragma unroll
        for (int kk = 0; kk < NL; ++kk) {
          const int k = kk + NL * kh;
          const int kc = k < a.K ? k : a.K - 1;
          v[kk] = a.nbr[(int64_t)kc * a.n_out + rowc];
        }
      }
#pragma unroll
      for (int kk = 0; kk < NL; ++kk)
        if (!rv || kk + NL * kh >= a.K) v[kk] = -1;
    } else {
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) v[kk] = (rv && kk + NL * kh < a.K) ? (int)row : -1;
    }
    // (two copies of the loop so that the multiply form is chosen once, not per entry; the product of a missing entry is
    // computed and discarded by a select -- no branch per offset)
    if (m24) {
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = kk + NL * kh;
        const unsigned prod = __umul24((unsigned)v[kk], row_bytes);
        if (k < F2_MAXK) off[k][rr] = v[kk] >= 0 ? prod : F3_MISSING;
        ml |= (v[kk] >= 0 ? 1u : 0u) << kk;
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = kk + NL * kh;
        const unsigned prod = (unsigned)v[kk] * row_bytes;
        if (k < F2_MAXK) off[k][rr] = v[kk] >= 0 ? prod : F3_MISSING;
        ml |= (v[kk] >= 0 ? 1u : 0u) << kk;
      }
    }
    ml = pp_row_or16(ml);
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
      m[tt] = 0;
#pragma unroll
      for (int h = 0; h < KPL; ++h)
        m[tt] |= (unsigned)__builtin_amdgcn_readlane((int)ml, h * R + tt * 16) << (NL * h);
    }
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
    const int S0 = C4 ? 1 : a.c0 >> 4, S = C4 ? 1 : (a.c0 + a.c1) >> 4;
    constexpr unsigned WT = C4 ? 256u : 1024u;  // bytes of packed weights per (k, s, column tile)
    const unsigned q16 = (unsigned)q * (C4 ? 4u : 16u);
    const unsigned lane16 = (unsigned)lane * (C4 ? 4u : 16u);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (int)w_bytes, 0x00020000);
    const unsigned w_step = (unsigned)a.NT * WT;  // bytes of packed weights per (k, s)
    const unsigned w_jt0 = (unsigned)jt0 * WT;

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
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) {                                                         \
      if constexpr (C4)                                                                                        \
        AX[tt][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra_, (int)vo[tt], 0, 0));   \
      else                                                                                                     \
        AX[tt] = F3_ABLATE == 3 ? (f32x4){1.f, 2.f, 3.f, (float)vo[tt]}                                         \
                                : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)vo[tt], 0, 0)); \
    }                                                                                                          \
    const unsigned wso_ = (unsigned)(kl * S + sl) * w_step + w_jt0;                                            \
    _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt) {                                                       \
      if constexpr (C4)                                                                                        \
        BX[jt][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, (int)(lane16 + jt * WT), (int)wso_, 0)); \
      else                                                                                                     \
        BX[jt] = F3_ABLATE == 4 ? (f32x4){1.f, 2.f, 3.f, (float)wso_}                                           \
                                : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(lane16 + jt * WT), (int)wso_, 0)); \
    }                                                                                                          \
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
            acc[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh_[jt], ah_, acc[tt][jt], 0, 0, 0);  \
      }                                                                                                   \
    }                                                                                                     \
  } else {                                                                                                \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) {                                                    \
      if ((m[tt] >> (KC)) & 1u) {                                                                         \
        _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)                                                \
            _Pragma("unroll") for (int t = 0; t < (C4 ? 1 : 4); ++t)                                      \
                acc[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(BX[jt][t], AX[tt][t], acc[tt][jt], 0, 0, 0); \
      }                                                                                                   \
    }                                                                                                     \
  }
    int more = 1;  // an int, not a bool: hipcc keeps bools as 64-bit lane masks (4 scalar instructions per test)
    // The loads are unconditional: after the last step the load side simply re-reads a valid step.  (A branch around
    // them makes hipcc merge the two paths' outstanding-load counts and wait for the NEW loads before the MFMAs.)
    if constexpr (C4) {
      // ---- the 4-channel input layer, FOUR kernel offsets per step (round 4).  A row is 16 bytes: lane (i, q) loads the whole
      // row of neighbour offset 4 g + q of its row i with ONE dwordx4, so the MFMA's k index is the offset inside the group and
      // its four issues t are the four channels: D^T[col][row] += sum_q W[4g+q][t][col] * in[nbr_{4g+q}(row)][t].  7 steps instead
      // of 27, each with T 16-byte gathers instead of 4 T 4-byte ones (the layer was bound by its load instructions: 0.06 of
      // the MFMA peak).  The weights keep the mode-4 packing: lane (col i, q) reads W[4g+q][t][col] as four dwords.
      (void)sl; (void)more; (void)vo; (void)kl;
      if (a.split > 1) {
        // split-K: a group may straddle this block's offset range; the offsets outside it are made absent in the table (the
        // group mask below only drops whole groups).  LDS operations of a wave execute in program order
#pragma unroll 1
        for (int k = 0; k < F2_MAXK; ++k)
          if (!((kmask >> k) & 1u))
            for (int r = lane; r < R; r += 64) off[k][r] = F3_MISSING;
      }
      const unsigned wrow = (unsigned)a.NT * 256u;                       // bytes of packed weights per offset
      const unsigned vb = (unsigned)q * wrow + (unsigned)i * 4u;         // lane part of the weight address
      const __amdgpu_buffer_rsrc_t rc4 = __builtin_amdgcn_make_buffer_rsrc((void*)a.in0, 0, (int)a_bytes, 0x00020000);
      unsigned remg = 0;
#pragma unroll
      for (int g = 0; g < 7; ++g) remg |= ((rem >> (4 * g)) & 15u) ? 1u << g : 0u;
#define C4_LOADS(G, AX, BX)                                                                                       \
  {                                                                                                               \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt)                                                              \
        AX[tt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rc4, (int)off[4 * (G) + q][tt * 16 + i], 0, 0)); \
    _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt) _Pragma("unroll") for (int t = 0; t < 4; ++t)              \
        BX[jt][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                               \
            rw, (int)(vb + t * 64u), (int)((unsigned)(4 * (G) * a.NT + jt0 + jt) * 256u), 0));                    \
  }
#define C4_MFMAS(G, AX, BX)                                                                                       \
  _Pragma("unroll") for (int tt = 0; tt < T; ++tt) {                                                              \
    if ((m[tt] >> (4 * (G))) & 15u) {                                                                             \
      _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt) _Pragma("unroll") for (int t = 0; t < 4; ++t)            \
          acc[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(BX[jt][t], AX[tt][t], acc[tt][jt], 0, 0, 0);         \
    }                                                                                                             \
  }
      int g0 = __builtin_ctz(remg), g1 = g0;
      unsigned rg = remg & (remg - 1u);
      C4_LOADS(g0, A0, B0);
      for (;;) {
        const int h1 = rg != 0u;
        g1 = h1 ? __builtin_ctz(rg) : g0;
        rg &= rg - 1u;
        C4_LOADS(g1, A1, B1);
        C4_MFMAS(g0, A0, B0);
        if (!h1) break;
        const int h0 = rg != 0u;
        g0 = h0 ? __builtin_ctz(rg) : g1;
        rg &= rg - 1u;
        C4_LOADS(g0, A0, B0);
        C4_MFMAS(g1, A1, B1);
        if (!h0) break;
      }
#undef C4_LOADS
#undef C4_MFMAS
    } else if constexpr (S1) {
      static_assert(!C4 && D == 3, "S1 is the depth-3 loop of the 16-channel layers");
      const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.in0, 0, (int)a_bytes, 0x00020000);
#define S1_LOADS(AX, BX, KK)                                                                                     \
  {                                                                                                              \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt)                                                             \
        AX[tt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra1, (int)vo[tt], 0, 0));       \
    const unsigned wso_ = (unsigned)(KK) * w_step + w_jt0;                                                       \
    _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)                                                           \
        BX[jt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(lane16 + jt * WT), (int)wso_, 0)); \
  }
      // kn / vo name the next offset, or stay on the current one when none is left (have = 0)
#define S1_ADV()                                                                                 \
  {                                                                                              \
    have = r != 0u ? 1 : 0;                                                                      \
    kn = have ? __builtin_ctz(r) : kn;                                                           \
    r &= r - 1u;                                                                                 \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) vo[tt] = off[kn][tt * 16 + i] | q16;        \
  }
      unsigned r = rem & (rem - 1u);
      int kc = kl, kn = kl, have = 0;
      S1_LOADS(A0, B0, kc);
      S1_ADV();
      for (;;) {
        S1_LOADS(A1, B1, kn);
        const int k0 = kc, d0 = !have;
        kc = kn;
        S1_ADV();
        F3_MFMAS(A0, B0, k0);
        if (d0) break;
        S1_LOADS(A0, B0, kn);
        const int k1 = kc, d1 = !have;
        kc = kn;
        S1_ADV();
        F3_MFMAS(A1, B1, k1);
        if (d1) break;
      }
#undef S1_LOADS
#undef S1_ADV
      (void)more; (void)sl;
    } else if constexpr (D == 1) {
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
      int k0 = 0, k1 = 0, e0 = 0, e1 = 0, nx = 0;
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
    } else if constexpr (D == 5) {
      // ---- register ring of depth 3: the operand loads of step n + 2 are issued before the MFMAs of step n.
      // PMC on the depth-1 loop (profiles/r02_pmc_c16_s1.md): a wave spends 36 % of its life in s_waitcnt -- a step's loads
      // are waited for one step (~1300 cycles incl. the other waves' MFMAs) after their issue, and a gathered row takes
      // ~2000 cycles to arrive while the texture path is busy.  hipcc's own three-set pipelines (round 2) needed 117 - 171
      // VGPRs and lost the gain to occupancy; here the loads are inline assembly with hand-counted waits (vmcnt is in issue
      // order: every step issues exactly T + NTW loads, so "at most 2 (T + NTW) outstanding" = the oldest step has landed) and
      // the three register sets cost exactly 3 (T + NTW) x 4 VGPRs.  hipcc does not count asm loads: no wait of its own is
      // emitted for them, and nothing else in the loop is a vector memory operation.
      static_assert(!C4, "the ring loop serves the 16-channel-step layers");
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      const unsigned long long pw_ = (unsigned long long)a.wp;
      const u32x4_t dw_ = {(unsigned)pw_, (unsigned)(pw_ >> 32) & 0xFFFFu, w_bytes, 0x00020000u};
      unsigned vw[NTW];
#pragma unroll
      for (int jt = 0; jt < NTW; ++jt) vw[jt] = lane16 + (unsigned)jt * WT;
      f32x4 A2[T], B2[NTW];
#define R_LOADS(AX, BX)                                                                                              \
  {                                                                                                                  \
    const float* src_ = sl < S0 ? a.in0 + sl * 16 : a.in1 + (sl - S0) * 16;                                          \
    const unsigned long long pa_ = (unsigned long long)src_;                                                         \
    const u32x4_t da_ = {(unsigned)pa_, (unsigned)(pa_ >> 32) & 0xFFFFu, a_bytes, 0x00020000u};                      \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt)                                                                 \
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(AX[tt]) : "v"(vo[tt]), "s"(da_));              \
    const unsigned wso_ = (unsigned)(kl * S + sl) * w_step + w_jt0;                                                  \
    _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)                                                               \
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(BX[jt]) : "v"(vw[jt]), "s"(dw_), "s"(wso_)); \
  }
      // the step the load side names goes into (AX, BX); KV = its offset, EV = whether it is a real step (after the last one
      // the load side keeps naming a valid address and the loads are simply repeated: the counts stay exact)
#define R_STAGE(AX, BX, KV, EV)      \
  {                                  \
    R_LOADS(AX, BX);                 \
    KV = kl;                         \
    EV = nx;                         \
    if (nx) { F3_ADVANCE(nx); }      \
  }
      // wait until the two youngest steps are the only loads in flight; the operands are named so that no MFMA moves above
#define R_WAIT(AX, BX)                                                                                               \
  {                                                                                                                  \
    asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(2 * (T + NTW)));                                                       \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) asm volatile("" : "+v"(AX[tt]));                               \
    _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt) asm volatile("" : "+v"(BX[jt]));                             \
  }
      int nx = 1, k0 = 0, k1 = 0, k2 = 0, e0 = 0, e1 = 0, e2 = 0;
      R_STAGE(A0, B0, k0, e0);
      R_STAGE(A1, B1, k1, e1);
      // ONE loop exit (three exits make hipcc keep the accumulators in different registers per exit and reconcile them with
      // 16-register copies around every tile branch: 152 VGPRs): steps behind the last real one run with an empty tile mask
#define R_MFMAS(AX, BX, KC, EV)                                  \
  {                                                              \
    unsigned mk_[T];                                             \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) mk_[tt] = m[tt]; \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) m[tt] = EV ? m[tt] : 0u; \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt)             \
        _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt) asm volatile("" : "+v"(acc[tt][jt])); \
    F3_MFMAS(AX, BX, KC);                                        \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) m[tt] = mk_[tt]; \
  }
      for (;;) {
        R_STAGE(A2, B2, k2, e2);
        R_WAIT(A0, B0);
        R_MFMAS(A0, B0, k0, e0);
        R_STAGE(A0, B0, k0, e0);
        R_WAIT(A1, B1);
        R_MFMAS(A1, B1, k1, e1);
        R_STAGE(A1, B1, k1, e1);
        R_WAIT(A2, B2);
        R_MFMAS(A2, B2, k2, e2);
        if (!e0) break;
      }
#undef R_MFMAS
      asm volatile("s_waitcnt vmcnt(0)");  // the repeated loads behind the last step: their registers die here
      (void)more;
#undef R_LOADS
#undef R_STAGE
#undef R_WAIT
    } else if constexpr (D == 6) {
      // ---- LDS-staged feature tiles (what north_star names; rows of >= 128 bytes, i.e. inputs with a multiple of 32 channels).
      // The depth-1 loop gathers the A operand in MFMA fragment shape: one instruction = 16 rows x 64 bytes, so a 128-byte
      // row is fetched as two HALF cache lines by two instructions.  profiles/r04_gather_forms.txt: that form costs the CU's
      // texture path 38.6 cycles per KiB from L2 and 179 from HBM, full lines (8 rows x 128 bytes per instruction) 20.5 / 90
      // -- the cost is per LINE touched, whatever part of it is used.  Full-line loads do not land in fragment layout, so they go
      // to LDS (buffer_load ... lds: no VGPRs, no ds_write pass) and the fragments are read back with ds_read_b128.
      //   step      = (offset k, 128-byte segment g of the row = two 16-channel steps)
      //   staging   = per tile 2 instructions (rows 0-7, 8-15; lane = (row l >> 3, chunk l & 7)); the lane reads source chunk
      //               (l & 7) ^ ((l >> 3) & 6) so that the linear LDS image is XOR-swizzled and the fragment reads -- lane (i, q)
      //               takes chunk 4 s + q of row i -- are bank-conflict free in every 16-lane group of ds_read_b128
      //   ring      = RD6 steps per wave, RD6 - 1 in flight; weights in RD6 register sets; all loads are inline assembly with
      //               counted waits (2 T + 2 NTW vector memory operations per step, in issue order)
      static_assert(!C4 && T == 2, "the LDS-staged loop: 32 rows per wave (its ring is 4 KiB per step)");
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      constexpr int LPS = 2 * T + 2 * NTW;
      const int G0 = S0 >> 1, G = S >> 1;
      const unsigned long long pw_ = (unsigned long long)a.wp;
      const u32x4_t dw_ = {(unsigned)pw_, (unsigned)(pw_ >> 32) & 0xFFFFu, w_bytes, 0x00020000u};
      unsigned vw[NTW];
#pragma unroll
      for (int jt = 0; jt < NTW; ++jt) vw[jt] = lane16 + (unsigned)jt * WT;
      const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
      f32x4* ringw = s_ring + wave_u * (RD6 * T * 128);
      const unsigned ring_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) f32x4*)ringw;
      const unsigned r8 = (unsigned)lane >> 3;
      const unsigned dma_c = (((unsigned)lane & 7u) ^ (r8 & 6u)) << 4;
      const unsigned rdb = ((unsigned)(i >> 3) << 10) + ((unsigned)(i & 7) << 7);
      const unsigned sw6 = (unsigned)(i & 6);
      const f32x4* rp0 = (const f32x4*)((const char*)ringw + rdb + (((unsigned)q ^ sw6) << 4));
      const f32x4* rp1 = (const f32x4*)((const char*)ringw + rdb + ((((unsigned)q + 4u) ^ sw6) << 4));
      unsigned vo2[T][2];
      int gl = 0;
#define L_OFFS()                                                                                          \
  _Pragma("unroll") for (int tt = 0; tt < T; ++tt) _Pragma("unroll") for (int j = 0; j < 2; ++j)           \
      vo2[tt][j] = off[kl][tt * 16 + 8 * j + r8] | dma_c;
      L_OFFS();
#define L_ADVANCE(VALID)           \
  {                                \
    VALID = 1;                     \
    ++gl;                          \
    if (gl == G) {                 \
      gl = 0;                      \
      rem &= rem - 1;              \
      if (rem) {                   \
        kl = __builtin_ctz(rem);   \
        L_OFFS();                  \
      } else                       \
        VALID = 0;                 \
    }                              \
  }
#define L_LOADS(SLOT, BX)                                                                                            \
  {                                                                                                                  \
    const float* src_ = gl < G0 ? a.in0 + gl * 32 : a.in1 + (gl - G0) * 32;                                          \
    const unsigned long long pa_ = (unsigned long long)src_;                                                         \
    const u32x4_t da_ = {(unsigned)pa_, (unsigned)(pa_ >> 32) & 0xFFFFu, a_bytes, 0x00020000u};                      \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                 \
      const unsigned lds_ = ring_lds + (unsigned)(((SLOT) * T + tt) * 2048 + j * 1024);                              \
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"                       \
                   ::"v"(vo2[tt][j]), "s"(lds_), "s"(da_) : "memory");                                               \
    }                                                                                                                \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                               \
      const unsigned wso_ = (unsigned)(kl * S + 2 * gl + s2) * w_step + w_jt0;                                       \
      _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)                                                             \
          asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(BX[s2][jt]) : "v"(vw[jt]), "s"(dw_), "s"(wso_)); \
    }                                                                                                                \
  }
#define L_STAGE(SLOT, BX, KV, EV)  \
  {                                \
    L_LOADS(SLOT, BX);             \
    KV = kl;                       \
    EV = nx;                       \
    if (nx) { L_ADVANCE(nx); }     \
  }
#define L_WAIT(BX)                                                                                                    \
  {                                                                                                                   \
    asm volatile("s_waitcnt vmcnt(%c0)" ::"n"((RD6 - 1) * LPS) : "memory");                                           \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)               \
        asm volatile("" : "+v"(BX[s2][jt]));                                                                          \
  }
#define L_MFMAS(SLOT, BX, KC, EV)                                                                                     \
  {                                                                                                                   \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)               \
        asm volatile("" : "+v"(acc[tt][jt]));                                                                         \
    _Pragma("unroll") for (int tt = 0; tt < T; ++tt) {                                                                \
      if (EV && ((m[tt] >> (KC)) & 1u)) {                                                                             \
        if constexpr (BF16) {                                                                                         \
          /* the two 16-channel steps of the segment as ONE v_mfma_f32_16x16x32_bf16 (gfx950: twice the k of the CDNA3 form): */ \
          /* lane (i, q) supplies k = 8 q + t, t < 4 from the first step, t >= 4 from the second -- the same map for both operands */ \
          const bf16x8_t a8_ = pp_bf16x8(rp0[((SLOT) * T + tt) * 128], rp1[((SLOT) * T + tt) * 128]);                  \
          _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt)                                                          \
              acc[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pp_bf16x8(BX[0][jt], BX[1][jt]), a8_, acc[tt][jt], 0, 0, 0); \
        } else {                                                                                                      \
          _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                          \
            const f32x4 av_ = (s2 ? rp1 : rp0)[((SLOT) * T + tt) * 128];                                              \
            _Pragma("unroll") for (int jt = 0; jt < NTW; ++jt) _Pragma("unroll") for (int t = 0; t < 4; ++t)          \
                acc[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(BX[s2][jt][t], av_[t], acc[tt][jt], 0, 0, 0);      \
          }                                                                                                           \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  }
      f32x4 BA[2][NTW], BB[2][NTW], BC[2][NTW];
      int nx = 1, k0 = 0, k1 = 0, k2 = 0, e0 = 0, e1 = 0, e2 = 0;
      if constexpr (RD6 == 3) {
        L_STAGE(0, BA, k0, e0);
        L_STAGE(1, BB, k1, e1);
        for (;;) {
          L_STAGE(2, BC, k2, e2);
          L_WAIT(BA);
          L_MFMAS(0, BA, k0, e0);
          L_STAGE(0, BA, k0, e0);
          L_WAIT(BB);
          L_MFMAS(1, BB, k1, e1);
          L_STAGE(1, BB, k1, e1);
          L_WAIT(BC);
          L_MFMAS(2, BC, k2, e2);
          if (!e0) break;
        }
      } else {
        L_STAGE(0, BA, k0, e0);
        for (;;) {
          L_STAGE(1, BB, k1, e1);
          L_WAIT(BA);
          L_MFMAS(0, BA, k0, e0);
          L_STAGE(0, BA, k0, e0);
          L_WAIT(BB);
          L_MFMAS(1, BB, k1, e1);
          if (!e0) break;
        }
        (void)BC; (void)k2; (void)e2;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      (void)more; (void)sl; (void)vo;
#undef L_OFFS
#undef L_ADVANCE
#undef L_LOADS
#undef L_STAGE
#undef L_WAIT
#undef L_MFMAS
    }
#undef F3_LOADS
#undef F3_ADVANCE
#undef F3_MFMAS
  }

  f32x4 acc2[DS ? T : 1][DS ? NTW : 1];
  if constexpr (DS) {
#pragma unroll
    for (int tt = 0; tt < T; ++tt)
#pragma unroll
      for (int jt = 0; jt < NTW; ++jt) acc2[tt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int S2 = a.ds_c >> 4;
    const unsigned ds_row = (unsigned)a.ds_c * 4u;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.ds_in, 0, (int)((unsigned)a.n_out * ds_row), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdw = __builtin_amdgcn_make_buffer_rsrc((void*)a.ds_wp, 0, (int)((unsigned)S2 * (unsigned)a.NT * 1024u), 0x00020000);
    unsigned o2[T];
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
      const int64_t r = row_base + tt * 16 + i;
      o2[tt] = r < a.n_out ? (unsigned)r * ds_row + (unsigned)q * 16u : F3_MISSING;
    }
    const unsigned lane16d = (unsigned)lane * 16u;
#pragma unroll 1
    for (int s2 = 0; s2 < S2; ++s2) {
      f32x4 A2[T], B2[NTW];
#pragma unroll
      for (int tt = 0; tt < T; ++tt)
        A2[tt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, (int)o2[tt], s2 * 64, 0));
#pragma unroll
      for (int jt = 0; jt < NTW; ++jt)
        B2[jt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdw, (int)(lane16d + jt * 1024u),
                                                                                 (int)((unsigned)(s2 * a.NT + jt0) * 1024u), 0));
      if constexpr (BF16) {
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
          const s16x4 ah = pp_bf16x4(A2[tt]);
#pragma unroll
          for (int jt = 0; jt < NTW; ++jt)
            acc2[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pp_bf16x4(B2[jt]), ah, acc2[tt][jt], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
#pragma unroll
          for (int jt = 0; jt < NTW; ++jt)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              acc2[tt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(B2[jt][t], A2[tt][t], acc2[tt][jt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue.  The MFMAs above compute out^T = W^T in^T (the weight fragment as the A operand, the gathered rows as B --
  // the same registers, products commute, the k order of every sum is unchanged: bit-identical to in W), so lane (j, q) holds
  // out[row j of the tile][16 jt + 4 q .. + 3]: 16 contiguous bytes.  One 16-byte store (and residual load) per (tile, column
  // tile) and ONE row address per tile, instead of four 4-byte accesses with an address each: the 16-channel layers spent a
  // tenth of their vector-ALU time -- the pipe the fp32 MFMAs run on -- in this part.
  const bool vec = (a.cout & 3) == 0;
  float* __restrict__ dst = a.split > 1 ? a.part + (int64_t)blockIdx.z * a.n_out * a.cout : a.out;
#pragma unroll
  for (int rt = 0; rt < T; ++rt) {
    const int64_t slot = row_base + rt * 16 + i;
    if (slot < a.n_out) {
      const int64_t row = a.row_order ? (int64_t)a.row_order[slot] : slot;
      const int64_t e0 = row * a.cout + (jt0 * 16 + q * 4);
#pragma unroll
      for (int jt = 0; jt < NTW; ++jt) {
        const int col = (jt0 + jt) * 16 + q * 4;
        if (jt0 + jt < a.NT && col < a.cout) {
          f32x4 v = acc[rt][jt];
          if (a.split > 1) {  // raw partial sums; scale / shift / ReLU / residual are applied by k_spconv_split_reduce
            if (vec) *(f32x4*)(dst + e0 + jt * 16) = v;
            else
              for (int r = 0; r < 4; ++r)
                if (col + r < a.cout) dst[e0 + jt * 16 + r] = v[r];
            continue;
          }
          if (vec) {
            if (a.scale) v *= *(const f32x4*)(a.scale + col);
            if (a.shift) v += *(const f32x4*)(a.shift + col);
            if (a.relu) v = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            if (a.residual) v += *(const f32x4*)(a.residual + e0 + jt * 16);
            if constexpr (DS) {
              f32x4 w = acc2[rt][jt];
              if (a.ds_scale) w *= *(const f32x4*)(a.ds_scale + col);
              if (a.ds_shift) w += *(const f32x4*)(a.ds_shift + col);
              v += w;
            }
            *(f32x4*)(dst + e0 + jt * 16) = v;
          } else {  // output widths that are not a multiple of 4 (none of the published networks): element by element
            for (int r = 0; r < 4; ++r) {
              if (col + r >= a.cout) break;
              float x = v[r] * (a.scale ? a.scale[col + r] : 1.f) + (a.shift ? a.shift[col + r] : 0.f);
              if (a.relu) x = fmaxf(x, 0.f);
              if (a.residual) x += a.residual[e0 + jt * 16 + r];
              if constexpr (DS)
                x += acc2[rt][jt][r] * (a.ds_scale ? a.ds_scale[col + r] : 1.f) + (a.ds_shift ? a.ds_shift[col + r] : 0.f);
              dst[e0 + jt * 16 + r] = x;
            }
          }
        }
      }
    }
  }
}

template <int T, bool BF16>
static int launch3_t(const SpconvArgs& a, int ntw, int depth, unsigned groups, unsigned a_bytes, unsigned w_bytes, unsigned flags, hipStream_t s) {
  dim3 grid(pp_blocks(a.n_out, 16 * T * F3_WPB), groups, (unsigned)(a.split > 1 ? a.split : 1));
  const bool s1 = a.c0 == 16 && a.c1 == 0;
  if (a.ds_in) {  // fused shortcut: the per-shape loop variant only (3 for <= 2 column tiles, 1 otherwise)
    if (s1 && ntw <= 2) {
      if (ntw == 1) hipLaunchKernelGGL((k_spconv_fwd3<1, T, BF16, 3, false, true, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags);
      else hipLaunchKernelGGL((k_spconv_fwd3<2, T, BF16, 3, false, true, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags);
      return PP_OK;
    }
    switch (ntw) {
      case 1: hipLaunchKernelGGL((k_spconv_fwd3<1, T, BF16, 3, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
      case 2: hipLaunchKernelGGL((k_spconv_fwd3<2, T, BF16, 3, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
      case 3:
        if (depth == 5) hipLaunchKernelGGL((k_spconv_fwd3<3, T, BF16, 5, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags);
        else hipLaunchKernelGGL((k_spconv_fwd3<3, T, BF16, 1, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags);
        break;
      case 4:
        if (depth == 5) hipLaunchKernelGGL((k_spconv_fwd3<4, T, BF16, 5, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags);
        else hipLaunchKernelGGL((k_spconv_fwd3<4, T, BF16, 1, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags);
        break;
      case 5: hipLaunchKernelGGL((k_spconv_fwd3<5, T, BF16, 1, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
      case 6: hipLaunchKernelGGL((k_spconv_fwd3<6, T, BF16, 1, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
      default: pp_set_error("pp_spconv_fwd3: ntw %d out of range", ntw); return PP_ERR_INVALID;
    }
    return PP_OK;
  }
  if constexpr (!BF16) {
    if (a.c0 == 4) {  // the input layer: one column-tile count per launch is enough (cout = 16 in every published model)
      switch (ntw) {
        case 1: hipLaunchKernelGGL((k_spconv_fwd3<1, T, false, 3, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
        case 2: hipLaunchKernelGGL((k_spconv_fwd3<2, T, false, 3, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
        case 3: hipLaunchKernelGGL((k_spconv_fwd3<3, T, false, 3, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
        case 4: hipLaunchKernelGGL((k_spconv_fwd3<4, T, false, 3, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
        default: pp_set_error("pp_spconv_fwd3: ntw %d out of range", ntw); return PP_ERR_INVALID;
      }
      return PP_OK;
    }
  }
  // depth: 1 = one step in flight; 3 = one step in flight with the load side advanced (LDS read of the next offsets)
  // before the MFMAs of the current step -- pays on launches with <= 2 column tiles per wave (16->16 at 2.5 M rows:
  // 374 -> 343 us), nothing on wider ones
  if (s1 && depth == 3 && ntw <= 2 && !a.ds_in) {
    if (ntw == 1) hipLaunchKernelGGL((k_spconv_fwd3<1, T, BF16, 3, false, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags);
    else hipLaunchKernelGGL((k_spconv_fwd3<2, T, BF16, 3, false, false, true>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags);
    return PP_OK;
  }
  if (depth == 6) {  // LDS-staged feature tiles: 32 rows per wave only (pp_spconv_fwd3_launch has checked the shape)
    if constexpr (T == 2) {
      switch (ntw) {
        case 1: hipLaunchKernelGGL((k_spconv_fwd3<1, 2, BF16, 6>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
        case 2: hipLaunchKernelGGL((k_spconv_fwd3<2, 2, BF16, 6>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
        case 3: hipLaunchKernelGGL((k_spconv_fwd3<3, 2, BF16, 6>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
        case 4: hipLaunchKernelGGL((k_spconv_fwd3<4, 2, BF16, 6>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
        default: pp_set_error("pp_spconv_fwd3: ntw %d out of range for the LDS-staged loop", ntw); return PP_ERR_INVALID;
      }
      return PP_OK;
    } else {
      pp_set_error("pp_spconv_fwd3: the LDS-staged loop runs with 32 rows per wave");
      return PP_ERR_INVALID;
    }
  }
#define F3_CASE(N, D) \
  case 10 * D + N: hipLaunchKernelGGL((k_spconv_fwd3<N, T, BF16, D>), grid, dim3(64 * F3_WPB), 0, s, a, a_bytes, w_bytes, flags); break;
  switch (10 * depth + ntw) {
    F3_CASE(1, 1) F3_CASE(2, 1) F3_CASE(3, 1) F3_CASE(4, 1) F3_CASE(5, 1) F3_CASE(6, 1)
    F3_CASE(1, 3) F3_CASE(2, 3) F3_CASE(3, 3) F3_CASE(4, 3)
    F3_CASE(1, 5) F3_CASE(2, 5) F3_CASE(3, 5) F3_CASE(4, 5)
    default: pp_set_error("pp_spconv_fwd3: ntw %d / depth %d out of range", ntw, depth); return PP_ERR_INVALID;
  }
#undef F3_CASE
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
  if (a.c0 % 16 != 0 && !(a.c0 == 4 && a.c1 == 0 && !a.bf16)) return false;
  return (double)n_in * a.c0 * 4.0 < 4294967000.0 && (double)a.K * (a.c0 + a.c1) * a.NT * 64.0 < 4294967000.0;
}

// T = 16-row tiles per wave (2 or 4), depth = loop variant (1 or 3); the caller picks them (spconv_fwd_impl)
int pp_spconv_fwd3_launch(const SpconvArgs& a, int64_t n_in, int ntw, unsigned groups, int T, int depth, hipStream_t s) {
  const unsigned a_bytes = (unsigned)((uint64_t)n_in * a.c0 * 4u);
  const unsigned w_bytes = a.c0 == 4 ? (unsigned)((uint64_t)a.K * a.NT * 256u)
                                     : (unsigned)((uint64_t)a.K * ((a.c0 + a.c1) / 16) * a.NT * 1024u);
  // bit 0: fewer than 2^24 input rows (24-bit multiplies); bit 1 / 2: the dense / 8-wide map fits 32-bit byte offsets (buffer loads)
  const unsigned flags = (n_in < (int64_t(1) << 24) ? 1u : 0u) |
                         (!a.t8 && a.nbr && (double)a.K * (double)a.n_out * 4.0 < 4294967000.0 ? 2u : 0u) |
                         (a.t8 == 1 && (double)a.n_out * 32.0 < 4294967000.0 ? 4u : 0u);
  if (depth == 6 && (T != 2 || a.c0 % 32 != 0 || a.c1 % 32 != 0 || a.ds_in || ntw > 4)) {
    pp_set_error("pp_spconv_fwd3: the LDS-staged loop needs 32 rows per wave, channel counts that are multiples of 32 and <= 4 column tiles");
    return PP_ERR_INVALID;
  }
  if (T != 2 && T != 4) {
    pp_set_error("pp_spconv_fwd3: rows per wave must be 32 or 64");
    return PP_ERR_INVALID;
  }
  if (a.bf16)
    return T == 4 ? launch3_t<4, true>(a, ntw, depth, groups, a_bytes, w_bytes, flags, s)
                  : launch3_t<2, true>(a, ntw, depth, groups, a_bytes, w_bytes, flags, s);
  return T == 4 ? launch3_t<4, false>(a, ntw, depth, groups, a_bytes, w_bytes, flags, s)
                : launch3_t<2, false>(a, ntw, depth, groups, a_bytes, w_bytes, flags, s);
}
