// K4 (layers with >= 32 input and output channels): the dense-offset sparse convolution on the bf16 matrix pipe with fp32
// results -- "X3".
//
// Why.  k_spconv_fwd3 (pp_spconv2.hip) multiplies fp32 operands with v_mfma_f32_16x16x4_f32, which runs at the fp32 VECTOR
// rate (157 TFLOP/s, 32 cycles per SIMD for 2048 flops); its >= 48-channel layers keep that pipe 0.74 - 0.77 busy (the 32-channel ones 0.65), i.e.
// they are bound by it.  v_mfma_f32_16x16x32_bf16 does 16384 flops in ~17 cycles.  An fp32 number is
// EXACTLY the sum of three bfloat16 numbers (8 + 8 + 8 significand bits: hi = the top 16 bits of x, mid = the top 16 bits
// of x - hi, lo = the rest; all three subtractions are exact), so
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) + [a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b|: dropped]
// six bf16 products (each exact in fp32), accumulated in fp32 by the MFMA: the error per product is below one fp32
// rounding, the sum over a K = 32 chunk is rounded once instead of 32 times.  Measured against float64 the error is of the
// size of the fp32-MFMA kernel's (0.6 - 1.4 x; tests/test_hip_ops.py::test_spconv_x3_*, bench.py's self-check).  Six bf16 MFMAs
// cost 6 x 17 = 102 cycles per (16 rows x 16 columns x 32 channels) against 8 x 32 = 256 cycles of fp32 MFMAs.
//
// What that needs.  At that rate the operand traffic of a 32 x 64 register tile no longer fits the CU's texture path
// (every wave loading its own weight fragments: 12 KiB per step), so
//   * the WORKGROUP (4 waves = 128 output rows) walks the occupied offsets of ITS rows in lockstep; the weight slice of a
//     step (offset k, 32 input channels, this launch's column tiles: NTW x 3 KiB of pre-split bf16 planes) is fetched ONCE
//     per workgroup with buffer_load ... lds (no VGPRs, no ds_write) into a double-buffered LDS stage, one barrier per step;
//   * the gathered rows stay fp32 in HBM (nothing else in the network changes): a wave gathers its two 16-row tiles as
//     before (zero-cost buffer loads, hardware zeros for missing neighbours) one step ahead, splits them in registers
//     (5.5 vector-ALU instructions per element, paid once per tile and step, used by NTW column tiles) and feeds
//     v_mfma_f32_16x16x32_bf16; a wave whose tiles lack the step's offset only takes part in the staging;
//   * weights are split once, at packing time (pp_pack_weight appends the section: [k][g][jt][plane][lane][8 bf16]).
// Same contract, prologue and epilogue as k_spconv_fwd3: per output row the sum runs over the occupied offsets in
// ascending order, inside an offset over the 32-channel groups, inside a group over the six products in a fixed order --
// independent of the map form, the row order and the batch, so results are bit-identical across those.
#include <stdlib.h>
#include <string.h>

#include "pp_spconv.h"

// The weight staging writes m0 in inline assembly and lists it as clobbered, so that the compiler never keeps a value of its own
// live in m0 across the statement (it re-materialises m0 before every instruction of its own that reads it).  clang warns about any
// reserved register on a clobber list; the clobber is what is wanted here.
#pragma clang diagnostic ignored "-Winline-asm"

#define X3_MAXK 28
#ifndef X3_WPB
#define X3_WPB 4  // waves per workgroup (A/B builds: 8 = 256 rows share a staged weight slice)
#endif
#define X3_MISSING 0xFFFFFFFFu
#ifndef X3_WAVES
#define X3_WAVES 4  // waves per SIMD the register budget is set for (A/B builds: 3)
#endif

// X3_ABLATE (profiling builds only, profiles/build_x3_variants.sh; results are wrong by construction): 1 = no row gathers,
// 2 = no operand split, 3 = no MFMAs, 4 = no weight staging, 5 = no per-step barrier
#ifndef X3_ABLATE
#define X3_ABLATE 0
#endif

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned x3_row_or16(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);  // row_ror:8
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);  // row_ror:4
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);  // row_ror:2
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);  // row_ror:1
  return v;
}

// 8 floats -> three bf16x8 planes with x == hi + mid + lo exactly (truncating splits: every remainder keeps the sign and
// fits the bits that are left).  MODE 1 (bfloat16 compute, configs[4]): one plane, round to nearest even.
struct X3Planes {
  bf16x8_t p0, p1, p2;
};
__device__ __forceinline__ unsigned x3_hi16(float x1, float x0) {  // (bits(x1) & 0xFFFF0000) | (bits(x0) >> 16)
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
__device__ __forceinline__ X3Planes x3_split(f32x4 lo4, f32x4 hi4) {
#if defined(X3_ABLATE) && X3_ABLATE == 2
  {
    X3Planes p;
    p.p0 = __builtin_bit_cast(bf16x8_t, lo4);
    p.p1 = __builtin_bit_cast(bf16x8_t, hi4);
    p.p2 = __builtin_bit_cast(bf16x8_t, lo4);
    return p;
  }
#endif
  float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  float r[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[e]) & 0xFFFF0000u);
    r[e] = x[e] - h;
    const float m = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r[e]) & 0xFFFF0000u);
    l[e] = r[e] - m;
  }
  u32x4_t a, b, c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = x3_hi16(x[2 * j + 1], x[2 * j]);
    b[j] = x3_hi16(r[2 * j + 1], r[2 * j]);
    c[j] = x3_hi16(l[2 * j + 1], l[2 * j]);
  }
  X3Planes p;
  p.p0 = __builtin_bit_cast(bf16x8_t, a);
  p.p1 = __builtin_bit_cast(bf16x8_t, b);
  p.p2 = __builtin_bit_cast(bf16x8_t, c);
  return p;
}

__device__ __forceinline__ X3Planes x3_round(f32x4 lo4, f32x4 hi4) {  // MODE 1: one plane, round to nearest even
  X3Planes p;
  p.p0 = pp_bf16x8(lo4, hi4);
  p.p1 = p.p0;
  p.p2 = p.p0;
  return p;
}

#if X3_ABLATE == 3
__device__ __forceinline__ f32x4 x3_mfma(bf16x8_t b, bf16x8_t a, f32x4 c) {
  asm volatile("" ::"v"(b), "v"(a));
  return c;
}
#else
#define x3_mfma(B, A, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(B, A, C, 0, 0, 0)
#endif

// MODE 0: fp32 results from exactly split operands (three planes, six products).  MODE 1: bfloat16 compute (configs[4],
// pp_spconv_fwd_bf16): both operands rounded to nearest-even bfloat16 -- the weights at packing time (third section of the
// packed buffer), the gathered rows in registers (v_cvt_pk_bf16_f32) -- ONE v_mfma_f32_16x16x32_bf16 per tile, column tile
// and 32 channels, fp32 accumulation; same workgroup-synchronous walk, one 1 KiB weight plane per column tile staged.
// T = 16-row tiles per wave: 2 (T = 4, 64 rows per wave, compiles but is not instantiated: it lost on every layer it was tried on).
template <int NTW, bool DS, int MODE, int T>
__global__ __launch_bounds__(64 * X3_WPB, X3_WAVES) void k_spconv_x3(SpconvArgs a, unsigned a_bytes, unsigned w_bytes, unsigned flags) {
  constexpr int R = 16 * T;
  static_assert(T == 2 || T == 4, "32 or 64 rows per wave");
  __shared__ unsigned s_off[X3_WPB][X3_MAXK][R];
  constexpr int PL = MODE == 1 ? 1 : 3;      // weight planes per column tile
  __shared__ f32x4 s_wb[2][NTW * PL * 64];  // weight stage: per column tile PL planes x 64 lanes x 16 bytes
  __shared__ unsigned s_u[X3_WPB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 15, q = lane >> 4;
  const unsigned bid = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row_base = ((int64_t)bid * X3_WPB + wave) * R;  // (may lie behind the last row: the wave then only stages weights)
  const int jt0 = blockIdx.y * NTW;
  unsigned(*off)[R] = s_off[wave];
  const unsigned row_bytes = (unsigned)a.c0 * 4u;
  const bool m24 = (flags & 1u) != 0u;

  // ---- prologue (as k_spconv_fwd3): neighbour rows -> byte offsets in LDS, per-tile occupancy masks -> SGPRs
  unsigned m[T];
  {
    constexpr int KPL = 64 / R, NL = X3_MAXK / KPL;
    const int rr = lane % R, kh = lane / R;
    const bool rv = row_base + rr < a.n_out;
    const int64_t slot = rv ? row_base + rr : a.n_out - 1;
    const int64_t row = a.row_order ? (int64_t)a.row_order[slot] : slot;
    if (a.t8 == 2) {
      // compact same-level map (pp_map_compact): table filled with MISSING, then the wave's present entries -- one contiguous run
      // of the entry array -- dropped into their (offset, row) slots; LDS operations of a wave execute in program order
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) off[kk + NL * kh][rr] = X3_MISSING;
      const unsigned mrow = (rv && kh == 0) ? a.cm_mask[slot] : 0u;
      if (row_base < a.n_out) {
        const int64_t chunks = (a.n_out + 31) >> 5;
        const int64_t c0 = row_base >> 5, c1 = c0 + R / 32 < chunks ? c0 + R / 32 : chunks;
        const int e0 = a.cm_start[c0], e1 = a.cm_start[c1];
        for (int eb = e0; eb < e1; eb += 256) {
          int v4[4];
          unsigned t4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = eb + u * 64 + lane;
            const int ec = e < e1 ? e : e1 - 1;
            v4[u] = a.nbr[ec];
            t4[u] = a.cm_tag[ec];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = eb + u * 64 + lane;
            const unsigned prod = m24 ? __umul24((unsigned)v4[u], row_bytes) : (unsigned)v4[u] * row_bytes;
            if (e < e1) off[t4[u] >> 6][(t4[u] & 63u) - ((unsigned)row_base & 63u)] = prod;  // (tag = offset << 6 | output row & 63)
          }
        }
      }
      const unsigned mk = x3_row_or16(mrow);
#pragma unroll
      for (int tt = 0; tt < T; ++tt) m[tt] = (unsigned)__builtin_amdgcn_readlane((int)mk, tt * 16);
    } else if (a.t8) {
      int e8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e8[j] = a.nbr[(int64_t)j * a.n_out + slot];
      unsigned cls = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) cls |= e8[j] >= 0 ? (unsigned)e8[j] >> 28 : 0u;
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) off[kk + NL * kh][rr] = X3_MISSING;
      unsigned mk = 0;
      const unsigned kx = (cls & 1u) ? 0u : 1u, ky = (cls & 2u) ? 0u : 3u, kz = (cls & 4u) ? 0u : 9u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = rv && (((unsigned)j & ~cls) == 0u) && e8[j] >= 0;
        const unsigned k = kx + ((j & 1) ? 2u : 0u) * (cls & 1u) + ky + ((j & 2) ? 6u : 0u) * ((cls >> 1) & 1u) + kz +
                           ((j & 4) ? 18u : 0u) * ((cls >> 2) & 1u);
        const unsigned rowj = (unsigned)e8[j] & PP_ROW_MASK;
        const unsigned prod = m24 ? __umul24(rowj, row_bytes) : rowj * row_bytes;
        off[(ok && kh == 0) ? k : 27u][rr] = (ok && kh == 0) ? prod : X3_MISSING;
        mk |= ok ? 1u << k : 0u;
      }
      mk = x3_row_or16(mk);
#pragma unroll
      for (int tt = 0; tt < T; ++tt) m[tt] = (unsigned)__builtin_amdgcn_readlane((int)mk, tt * 16);
    } else {
      int v[NL];
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const int k = kk + NL * kh;
        if (a.nbr) {
          const int kc = k < a.K ? k : a.K - 1;
          v[kk] = a.nbr[(int64_t)kc * a.n_out + slot];
          if (!rv || k >= a.K) v[kk] = -1;
        } else {
          v[kk] = (rv && k < a.K) ? (int)row : -1;
        }
      }
      unsigned ml = 0;
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) {
        const unsigned prod = m24 ? __umul24((unsigned)v[kk], row_bytes) : (unsigned)v[kk] * row_bytes;
        off[kk + NL * kh][rr] = v[kk] >= 0 ? prod : X3_MISSING;
        ml |= (v[kk] >= 0 ? 1u : 0u) << kk;
      }
      ml = x3_row_or16(ml);
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
        m[tt] = 0;
#pragma unroll
        for (int h = 0; h < KPL; ++h) m[tt] |= (unsigned)__builtin_amdgcn_readlane((int)ml, h * R + tt * 16) << (NL * h);
      }
    }
  }
  unsigned kmask = 0xFFFFFFFFu;
  if (a.split > 1) {
    const int k0 = (int)blockIdx.z * a.K / a.split, k1 = ((int)blockIdx.z + 1) * a.K / a.split;
    kmask = (k1 >= 32 ? 0xFFFFFFFFu : (1u << k1) - 1u) & ~((1u << k0) - 1u);
  }
  unsigned rem = 0;
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    m[tt] = __builtin_amdgcn_readfirstlane(m[tt]) & kmask;
    rem |= m[tt];
  }
  if (lane == 0) s_u[wave] = rem;
  __syncthreads();
  unsigned U = 0;
#pragma unroll
  for (int w = 0; w < X3_WPB; ++w) U |= s_u[w];
  U = (unsigned)__builtin_amdgcn_readfirstlane((int)U);

  f32x4 acc[T][NTW];
#pragma unroll
  for (int tt = 0; tt < T; ++tt)
#pragma unroll
    for (int jt = 0; jt < NTW; ++jt) acc[tt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (U) {  // workgroup-uniform
    const int S0 = a.c0 >> 4, S = (a.c0 + a.c1) >> 4, G = (S + 1) >> 1;
    const unsigned q16 = (unsigned)q * 16u;
    // the pre-split section of the packed weights follows the fp32 one: [k][g][jt][plane][lane][8 bf16]
    // (MODE 1: the bfloat16 section behind that one: [k][g][jt][lane][8 bf16])
    const unsigned x3_bytes = (unsigned)a.K * (unsigned)G * (unsigned)a.NT * 3072u;
    const unsigned long long pw_ = (unsigned long long)a.wp + w_bytes + (MODE == 1 ? x3_bytes : 0u);
    const unsigned wx_bytes = MODE == 1 ? x3_bytes / 3u : x3_bytes;
    const u32x4_t dw_ = {(unsigned)pw_, (unsigned)(pw_ >> 32) & 0xFFFFu, wx_bytes, 0x00020000u};
    const unsigned slice = (unsigned)a.NT * (PL * 1024u);  // bytes per (k, g)
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned wb_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) f32x4*)&s_wb[0][0];

    // stage the weight slice of step (k, g) into buffer b: 3 NTW pieces of 1 KiB, piece j by wave j % 4
#ifndef X3_STAGE_ARITH
#define X3_STAGE_ARITH 1  // LDS-DMA staging: 1 = piece index computed from the wave number, 0 (A/B builds) = one compare + branch per piece
#endif
#ifndef X3_AHEAD
#define X3_AHEAD 1        // gathered rows in flight: 1 = one step ahead; 2 (A/B builds) = two steps ahead, two register sets in ping-pong --
                          // measured neutral (-0.1 % over 11 layer shapes, profiles/r06_x3_ab_ahead.txt): not bound by the gathers' latency
#endif
#ifndef X3_TILE_ORDER
#define X3_TILE_ORDER 1   // MFMAs of a column tile: 1 = tile by tile, accumulating in place, 0 (A/B builds) = the two tiles alternating
#endif
#ifndef X3_STAGE_MODE
#define X3_STAGE_MODE 0  // 0: buffer_load ... lds; 1 (A/B builds): loads into registers at the start of a step, ds_write at its end
#endif
    constexpr int NPW = (PL * NTW + X3_WPB - 1) / X3_WPB;  // pieces per wave
    f32x4 wreg[X3_STAGE_MODE == 1 ? NPW : 1];
    (void)wreg;
#define X3_STAGE_W(KK, GG, BUF)                                                                              \
  {                                                                                                          \
    const unsigned so_ = ((unsigned)(KK) * (unsigned)G + (unsigned)(GG)) * slice + (unsigned)jt0 * (PL * 1024u); \
    if (X3_STAGE_MODE == 1) {                                                                                \
      const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)pw_, 0, (int)wx_bytes, 0x00020000); \
      _Pragma("unroll") for (int jj = 0; jj < NPW; ++jj)                                                     \
          wreg[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                        \
              rw_, (int)lane16, (int)(so_ + (unsigned)(jj * X3_WPB + wave) * 1024u), 0));                    \
    } else if (X3_STAGE_ARITH) {                                                                             \
      /* piece jj * X3_WPB + wave of this wave: scalar arithmetic on the wave number, no branch per piece */ \
      _Pragma("unroll") for (int jj = 0; jj < NPW; ++jj) {                                                   \
        const int j_ = jj * X3_WPB + wave;                                                                   \
        if ((jj + 1) * X3_WPB <= PL * NTW || j_ < PL * NTW) {                                                \
          const unsigned lds_ = wb_lds + (unsigned)(BUF) * (NTW * PL * 1024u) + (unsigned)j_ * 1024u;        \
          const unsigned sj_ = so_ + (unsigned)j_ * 1024u;                                                   \
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"           \
                       ::"v"(lane16), "s"(lds_), "s"(dw_), "s"(sj_) : "memory", "m0");                       \
        }                                                                                                    \
      }                                                                                                      \
    } else {                                                                                                 \
      _Pragma("unroll") for (int j = 0; j < PL * NTW; ++j) {                                                 \
        if ((j % X3_WPB) == wave) {                                                                          \
          const unsigned lds_ = wb_lds + (unsigned)(BUF) * (NTW * PL * 1024u) + (unsigned)j * 1024u;              \
          const unsigned sj_ = so_ + (unsigned)j * 1024u;                                                    \
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"           \
                       ::"v"(lane16), "s"(lds_), "s"(dw_), "s"(sj_) : "memory", "m0");                             \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
  }
    // (register staging: the pieces loaded at the start of the step go to LDS right before its barrier)
#define X3_STAGE_FLUSH(BUF)                                                                                  \
  if (X3_STAGE_MODE == 1) {                                                                                  \
    _Pragma("unroll") for (int jj = 0; jj < NPW; ++jj)                                                       \
        if (jj * X3_WPB + wave < PL * NTW) s_wb[BUF][(jj * X3_WPB + wave) * 64 + lane] = wreg[jj];            \
  }
    // gather the two 16-channel halves of group g of offset k for the tile pair PR (tiles 2 PR, 2 PR + 1; missing half /
    // neighbour: hardware zeros)
#define X3_GATHER2(KK, GG, AX, PR)                                                                             \
  {                                                                                                            \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                            \
      const int sl_ = 2 * (GG) + h;                                                                            \
      const int sc_ = sl_ < S ? sl_ : S - 1;                                                                   \
      const float* src_ = sc_ < S0 ? a.in0 + sc_ * 16 : a.in1 + (sc_ - S0) * 16;                               \
      const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)src_, 0, (int)(sl_ < S ? a_bytes : 0u), 0x00020000); \
      _Pragma("unroll") for (int tt = 2 * (PR); tt < 2 * (PR) + 2; ++tt)                                       \
          AX[tt][h] = X3_ABLATE == 1 ? (f32x4){1.f, 2.f, 3.f, (float)off[KK][tt * 16 + i]}                      \
                                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)(off[KK][tt * 16 + i] | q16), 0, 0)); \
    }                                                                                                          \
  }
    // ---- the workgroup's step sequence: (offset k of U ascending) x (group g); two positions are live: the step being
    // computed (k0, g0) and the next one (weights being staged, rows in flight).  (Rows two steps ahead were tried: they fit 128
    // VGPRs only with spills, and at three waves per SIMD the kernel lost 5 %, profiles/r05_x3_ab_variants.txt.)
#ifndef X3_BDEPTH
#define X3_BDEPTH 2  // weight fragments of one column tile (1) or two (2: the next tile's LDS reads run beside the MFMAs) in registers
#endif
    // X3_ABLATE (profiling builds only, profiles/build_x3_variants.sh; results are wrong by construction): 1 = no row gathers,
    // 2 = no operand split, 3 = no MFMAs, 4 = no weight staging, 5 = no per-step barrier
#define X3_ADV(KI, GI, UR, OK)            \
  {                                       \
    if (GI + 1 < G) {                     \
      ++GI;                               \
    } else if (UR) {                      \
      KI = __builtin_ctz(UR);             \
      UR &= UR - 1u;                      \
      GI = 0;                             \
    } else {                              \
      OK = 0;                             \
    }                                     \
  }
    int k0 = __builtin_ctz(U), g0 = 0, buf = 0;
    unsigned Ur = U & (U - 1u);
    int k1 = k0, g1 = g0, ok1 = 1;
    X3_ADV(k1, g1, Ur, ok1);
    f32x4 AA[T][2];
    X3_STAGE_W(k0, g0, 0);
    X3_STAGE_FLUSH(0);
#pragma unroll
    for (int pr = 0; pr < T / 2; ++pr) X3_GATHER2(k0, g0, AA, pr);
#define X3_SIX(ACC, PL, B0, B1, B2)                                                       \
  ACC = x3_mfma(B2, PL.p0, ACC);                 \
  ACC = x3_mfma(B0, PL.p2, ACC);                 \
  ACC = x3_mfma(B1, PL.p1, ACC);                 \
  ACC = x3_mfma(B1, PL.p0, ACC);                 \
  ACC = x3_mfma(B0, PL.p1, ACC);                 \
  ACC = x3_mfma(B0, PL.p0, ACC);
    // one step: per tile pair -- split its rows of (k0, g0) held in AX, refill AX with the rows of step (kg, gg) if there is one
    // (okg), multiply; the next step's weights are staged beside the first pair's MFMAs
    auto x3_step = [&](f32x4 (&AX)[T][2], const int kg, const int gg, const int okg) __attribute__((always_inline)) {
#pragma unroll
      for (int pr = 0; pr < T / 2; ++pr) {
        const int ta = 2 * pr, tb = 2 * pr + 1;
        const unsigned act0 = (m[ta] >> k0) & 1u, act1 = (m[tb] >> k0) & 1u;
        X3Planes P0, P1;
        if (act0) P0 = MODE == 1 ? x3_round(AX[ta][0], AX[ta][1]) : x3_split(AX[ta][0], AX[ta][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (act1) P1 = MODE == 1 ? x3_round(AX[tb][0], AX[tb][1]) : x3_split(AX[tb][0], AX[tb][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (pr == 0 && ok1 && X3_ABLATE != 4) X3_STAGE_W(k1, g1, buf ^ 1);
        if (okg) X3_GATHER2(kg, gg, AX, pr);
        if (act0 | act1) {
          const f32x4* wb = &s_wb[buf][0];
          bf16x8_t Bf[2][PL];
          if (X3_BDEPTH == 2) {
#pragma unroll
            for (int p = 0; p < PL; ++p) Bf[0][p] = __builtin_bit_cast(bf16x8_t, wb[p * 64 + lane]);
          }
#pragma unroll
          for (int jt = 0; jt < NTW; ++jt) {
            if (X3_BDEPTH == 2) {
              if (jt + 1 < NTW) {
#pragma unroll
                for (int p = 0; p < PL; ++p)
                  Bf[(jt + 1) & 1][p] = __builtin_bit_cast(bf16x8_t, wb[(jt + 1) * (PL * 64) + p * 64 + lane]);
              }
            } else {
#pragma unroll
              for (int p = 0; p < PL; ++p) Bf[jt & 1][p] = __builtin_bit_cast(bf16x8_t, wb[jt * (PL * 64) + p * 64 + lane]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE == 1) {
              const bf16x8_t B0 = Bf[jt & 1][0];
              if (act0) acc[ta][jt] = x3_mfma(B0, P0.p0, acc[ta][jt]);
              if (act1) acc[tb][jt] = x3_mfma(B0, P1.p0, acc[tb][jt]);
            } else {
              const bf16x8_t B0 = Bf[jt & 1][0], B1 = Bf[jt & 1][PL > 1 ? 1 : 0], B2 = Bf[jt & 1][PL > 2 ? 2 : 0];
              if (X3_TILE_ORDER == 1) {
                // tile by tile: every chain accumulates IN PLACE.  The alternating form below merges three code paths per
                // accumulator; the compiler then writes the products into fresh registers and copies them back -- 4 v_mov_b64
                // behind an s_nop after EVERY column tile and 12 more at the loop's back edge, each waiting for the matrix pipe
                // to drain (round 6, read off the ISA)
                if (act0) {
                  X3_SIX(acc[ta][jt], P0, B0, B1, B2)
                }
                if (act1) {
                  X3_SIX(acc[tb][jt], P1, B0, B1, B2)
                }
              } else
              if (act0 & act1) {
                // the two tiles alternate: consecutive MFMAs never wait for each other's accumulator
                acc[ta][jt] = x3_mfma(B2, P0.p0, acc[ta][jt]);
                acc[tb][jt] = x3_mfma(B2, P1.p0, acc[tb][jt]);
                acc[ta][jt] = x3_mfma(B0, P0.p2, acc[ta][jt]);
                acc[tb][jt] = x3_mfma(B0, P1.p2, acc[tb][jt]);
                acc[ta][jt] = x3_mfma(B1, P0.p1, acc[ta][jt]);
                acc[tb][jt] = x3_mfma(B1, P1.p1, acc[tb][jt]);
                acc[ta][jt] = x3_mfma(B1, P0.p0, acc[ta][jt]);
                acc[tb][jt] = x3_mfma(B1, P1.p0, acc[tb][jt]);
                acc[ta][jt] = x3_mfma(B0, P0.p1, acc[ta][jt]);
                acc[tb][jt] = x3_mfma(B0, P1.p1, acc[tb][jt]);
                acc[ta][jt] = x3_mfma(B0, P0.p0, acc[ta][jt]);
                acc[tb][jt] = x3_mfma(B0, P1.p0, acc[tb][jt]);
              } else if (act0) {
                X3_SIX(acc[ta][jt], P0, B0, B1, B2)
              } else {
                X3_SIX(acc[tb][jt], P1, B0, B1, B2)
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
#if X3_AHEAD == 2
    // Rows TWO steps ahead (round 6, A/B builds): the gathers of step n + 2 are issued right after the rows of step n have been split,
    // into the registers that split freed -- two register sets in ping-pong, the loop unrolled by two, no copies.  Fits without
    // spills since the in-place MFMA chains freed 11 VGPRs (103 -> 118 at four column tiles, four waves per SIMD kept).  Before the
    // barrier a wave waits for everything but the four gathers it issued last (vmcnt counts in order: the staging loads of step
    // n + 1 were issued before them).  Result: 19.31 against 19.34 ms over eleven layer shapes -- neutral; round 5's attempt had to
    // give up a wave per SIMD for it and lost 5 %.  The kernel is not waiting for its rows.
    static_assert(T == 2, "the two-steps-ahead form holds two tiles per wave");
    int k2 = k1, g2 = g1, ok2 = ok1;
    if (ok1) X3_ADV(k2, g2, Ur, ok2);
    f32x4 AB[T][2];
    if (ok1) X3_GATHER2(k1, g1, AB, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#define X3_NEXT()                                                    \
  {                                                                  \
    X3_STAGE_FLUSH(buf ^ 1);                                         \
    if (ok2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            \
    if (X3_ABLATE != 5) __syncthreads();                             \
    k0 = k1; g0 = g1; k1 = k2; g1 = g2; ok1 = ok2;                   \
    if (ok2) X3_ADV(k2, g2, Ur, ok2);                                \
    buf ^= 1;                                                        \
  }
    for (;;) {
      x3_step(AA, k2, g2, ok1 & ok2);
      if (!ok1) break;
      X3_NEXT();
      x3_step(AB, k2, g2, ok1 & ok2);
      if (!ok1) break;
      X3_NEXT();
    }
#undef X3_NEXT
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (;;) {
      x3_step(AA, k1, g1, ok1);
      if (!ok1) break;
      // before the barrier: this wave's share of the next step's weights has landed (and its rows of the next step)
      X3_STAGE_FLUSH(buf ^ 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (X3_ABLATE != 5) __syncthreads();
      k0 = k1;
      g0 = g1;
      X3_ADV(k1, g1, Ur, ok1);
      buf ^= 1;
    }
#endif
#undef X3_SIX
#undef X3_ADV
#undef X3_GATHER2
#undef X3_STAGE_W
#undef X3_STAGE_FLUSH
  }

  // ---- fused 1x1 shortcut (fp32 MFMAs on the wave's own rows, as in k_spconv_fwd3: 1/27 of the work)
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
      o2[tt] = r < a.n_out ? (unsigned)r * ds_row + (unsigned)q * 16u : X3_MISSING;
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
      if constexpr (MODE == 1) {
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

  // ---- epilogue (as k_spconv_fwd3; cout % 4 == 0 is a launch condition): lane (j, q) holds out[row j][16 jt + 4 q .. + 3]
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
          if (a.split > 1) {
            *(f32x4*)(dst + e0 + jt * 16) = v;
            continue;
          }
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
        }
      }
    }
  }
}

// ---- K4f: the same convolution with the rows gathered as FULL 128-BYTE LINES straight into LDS (round 6) -----------------------
//
// k_spconv_x3 gathers its rows in MFMA fragment shape: lane (i, q) loads 16 bytes of row i, so the 64 lanes of one buffer load
// touch 16 rows x 4 separate 16-byte pieces.  The texture path prices an instruction by the pieces ADJACENT lanes can be merged
// into (profiles/r06_gather_forms.txt, L2-resident table): that form costs 46 cycles per wave instruction, eight adjacent lanes
// reading one 128-byte line 20 -- and on misses the 64-byte-piece forms reach half the bandwidth of full lines (3.8 vs 7.6 TB/s).
// The kernel is bound by that path (texture path busy 60 %, matrix pipe 41 %, profiles/r05_pmc_conv_x3_c64.md), so here
//   * a 32-channel group of 32 rows is fetched by four buffer_load_dwordx4 ... lds, each = 8 rows x one whole 128-byte line (lane
//     (r, c) = (lane >> 3, lane & 7) reads chunk c ^ s of row r), landing as a row-major 1-KiB block in LDS; the MFMA fragments
//     (lane (i, q): chunks q and 4 + q of row i) are read back with ds_read_b128.  The XOR s = (slot & 3) + 4 (slot >> 3 & 1) on the
//     chunk makes those reads bank-conflict-free without padding (the texture path merges the 8 lanes of a line in any order:
//     20.3 cycles swizzled, 20.2 plain), so the block is exactly 4 KiB per wave and four workgroups still fit a CU at 4 column tiles;
//   * one block per wave, single-buffered: the fragments of step n are in registers before the lines of step n + 1 are requested
//     into the same block, and they have the whole split + MFMA phase to land -- the distance k_spconv_x3's register gathers have;
//   * the 14-KiB neighbour table leaves LDS: a lane keeps the byte offsets of ITS four rows for the offset k being fetched (4
//     VGPRs) and the raw map entries of the wave's next offset (4 VGPRs, one 16-byte load per offset: slot 4 r + j of the wave is
//     row r of block j, so a lane's four entries are adjacent in the map); the prologue only derives the tiles' occupancy masks.
// Same channel -> k mapping, same packed weights, same summation order as k_spconv_x3: bit-identical results
// (tests/test_hip_ops.py::test_spconv_x3_full_line_gathers_are_bit_identical).  Dense maps (same-level, strided) with
// c0 % 32 == 0; the 8-wide transposed and compact forms and the 48 / 80 / 112-channel inputs stay on k_spconv_x3.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
// X3F_ABLATE (profiling builds only, profiles/build_x3_variants.sh; results are wrong by construction): 1 = no line gathers,
// 2 = no operand split (X3_ABLATE=2 shares the switch), 3 = no MFMAs (X3_ABLATE=3), 4 = no weight staging, 5 = no per-step barrier,
// 6 = no fragment reads from LDS, 7 = no offset pipeline (map loads + ds_bpermute)
#ifndef X3F_ABLATE
#define X3F_ABLATE 0
#endif
template <int NTW, bool DS, int MODE>
__global__ __launch_bounds__(64 * X3_WPB, (NTW <= 2 && !DS) ? 5 : X3_WAVES) void k_spconv_x3f(SpconvArgs a, unsigned a_bytes, unsigned w_bytes, unsigned flags,
                                                                        unsigned nbr_bytes) {
  constexpr int T = 2, R = 32;
  constexpr int PL = MODE == 1 ? 1 : 3;
  constexpr int WB = NTW * PL * 64;                  // f32x4 per weight stage buffer
  __shared__ f32x4 s_mem[2 * WB + X3_WPB * 256];     // [2][WB] weight stage | per wave 4 blocks x 1 KiB of gathered lines
  f32x4* const s_wb = s_mem;
  unsigned* const s_u = (unsigned*)(s_mem + WB);     // (the union of the waves' masks: aliases stage buffer 1, written after the 2nd barrier)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 15, q = lane >> 4;
  const unsigned bid = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row_base = ((int64_t)bid * X3_WPB + wave) * R;
  const int jt0 = blockIdx.y * NTW;
  const unsigned row_bytes = (unsigned)a.c0 * 4u;
  const bool m24 = (flags & 1u) != 0u;

  // ---- prologue: per-tile occupancy masks.  Dense map: its sign bits (lanes 0 .. 31 offsets 0 .. 13, lanes 32 .. 63 offsets 14 .. 27).
  // 8-wide transposed map: entry j of a fine row of parity class cls is offset k(cls, j) (as in k_spconv_x3's prologue)
  unsigned m[T];
  if (a.t8) {
    const int rr = lane & 31;
    const bool rv = row_base + rr < a.n_out;
    const int64_t slot = rv ? row_base + rr : a.n_out - 1;
    int e8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e8[j] = a.nbr[(int64_t)j * a.n_out + slot];
    unsigned cls = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) cls |= e8[j] >= 0 ? (unsigned)e8[j] >> 28 : 0u;
    unsigned mk = 0;
    const unsigned kx = (cls & 1u) ? 0u : 1u, ky = (cls & 2u) ? 0u : 3u, kz = (cls & 4u) ? 0u : 9u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = rv && (((unsigned)j & ~cls) == 0u) && e8[j] >= 0;
      const unsigned k = kx + ((j & 1) ? 2u : 0u) * (cls & 1u) + ky + ((j & 2) ? 6u : 0u) * ((cls >> 1) & 1u) + kz +
                         ((j & 4) ? 18u : 0u) * ((cls >> 2) & 1u);
      mk |= ok ? 1u << k : 0u;
    }
    mk = x3_row_or16(mk);
#pragma unroll
    for (int tt = 0; tt < T; ++tt) m[tt] = (unsigned)__builtin_amdgcn_readlane((int)mk, tt * 16);
  } else {
    constexpr int NL = X3_MAXK / 2;
    const int rr = lane & 31, kh = lane >> 5;
    const bool rv = row_base + rr < a.n_out;
    const int64_t slot = rv ? row_base + rr : a.n_out - 1;
    unsigned ml = 0;
#pragma unroll
    for (int kk = 0; kk < NL; ++kk) {
      const int k = kk + NL * kh;
      const int kc = k < a.K ? k : a.K - 1;
      const int v = a.nbr[(int64_t)kc * a.n_out + slot];
      ml |= ((rv && k < a.K && v >= 0) ? 1u : 0u) << kk;
    }
    ml = x3_row_or16(ml);
#pragma unroll
    for (int tt = 0; tt < T; ++tt)
      m[tt] = (unsigned)__builtin_amdgcn_readlane((int)ml, tt * 16) | ((unsigned)__builtin_amdgcn_readlane((int)ml, 32 + tt * 16) << NL);
  }
  unsigned kmask = 0xFFFFFFFFu;
  if (a.split > 1) {
    const int k0 = (int)blockIdx.z * a.K / a.split, k1 = ((int)blockIdx.z + 1) * a.K / a.split;
    kmask = (k1 >= 32 ? 0xFFFFFFFFu : (1u << k1) - 1u) & ~((1u << k0) - 1u);
  }
  unsigned Uw = 0;  // offsets this wave computes
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    m[tt] = __builtin_amdgcn_readfirstlane(m[tt]) & kmask;
    Uw |= m[tt];
  }
  if (lane == 0) s_u[wave] = Uw;
  __syncthreads();
  unsigned U = 0;
#pragma unroll
  for (int w = 0; w < X3_WPB; ++w) U |= s_u[w];
  U = (unsigned)__builtin_amdgcn_readfirstlane((int)U);

  f32x4 acc[T][NTW];
#pragma unroll
  for (int tt = 0; tt < T; ++tt)
#pragma unroll
    for (int jt = 0; jt < NTW; ++jt) acc[tt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (U) {  // workgroup-uniform
    const int S0 = a.c0 >> 4, S = (a.c0 + a.c1) >> 4, G = S >> 1;  // (c0 % 32 == 0 and c1 in {0, c0}: whole groups, none straddles the sources)
    const unsigned x3_bytes = (unsigned)a.K * (unsigned)G * (unsigned)a.NT * 3072u;
    const unsigned long long pw_ = (unsigned long long)a.wp + w_bytes + (MODE == 1 ? x3_bytes : 0u);
    const unsigned wx_bytes = MODE == 1 ? x3_bytes / 3u : x3_bytes;
    const u32x4_t dw_ = {(unsigned)pw_, (unsigned)(pw_ >> 32) & 0xFFFFu, wx_bytes, 0x00020000u};
    const unsigned slice = (unsigned)a.NT * (PL * 1024u);  // weight bytes per (k, g)
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned wb_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) f32x4*)&s_mem[0];
    const unsigned rows_lds = wb_lds + 2u * WB * 16u + (unsigned)wave * 4096u;
    constexpr int NPW = (PL * NTW + X3_WPB - 1) / X3_WPB;  // weight pieces per wave
#define X3F_STAGE_W(KK, GG, BUF)                                                                             \
  {                                                                                                          \
    const unsigned so_ = ((unsigned)(KK) * (unsigned)G + (unsigned)(GG)) * slice + (unsigned)jt0 * (PL * 1024u); \
    _Pragma("unroll") for (int jj = 0; jj < NPW; ++jj) {                                                     \
      const int j_ = jj * X3_WPB + wave;                                                                     \
      if ((jj + 1) * X3_WPB <= PL * NTW || j_ < PL * NTW) {                                                  \
        const unsigned lds_ = wb_lds + (unsigned)(BUF) * (WB * 16u) + (unsigned)j_ * 1024u;                  \
        const unsigned sj_ = so_ + (unsigned)j_ * 1024u;                                                     \
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"             \
                     ::"v"(lane16), "s"(lds_), "s"(dw_), "s"(sj_) : "memory", "m0");                         \
      }                                                                                                      \
    }                                                                                                        \
  }
    // gather role: lane (r, c) = (lane >> 3, lane & 7); block j, row r <-> slot 4 r + j of the wave; chunk c ^ (j + 4 (r >> 1 & 1))
    const unsigned gr = (unsigned)lane >> 3, gc = (unsigned)lane & 7u;
    const unsigned x16 = (gc ^ (4u * ((gr >> 1) & 1u))) * 16u;
    // map role: lane (r, c) owns slot 4 r + (c & 3) of the wave (every slot twice): ONE 4-byte load per offset, contiguous inside the
    // lane quads; the entry is turned into the byte offset of the row (all ones: no neighbour / a slot behind the last row / on an
    // 8-wide map a row of another parity class), and four quad broadcasts (DPP quad_perm: plain vector-ALU moves, no LDS crossbar, no
    // wait) hand lane (r, c) the offsets of slots 4 r .. 4 r + 3
    const unsigned mslot = 4u * gr + (gc & 3u);
    const unsigned tail1 = (row_base + (int64_t)mslot >= a.n_out) ? 0xFFFFFFFFu : 0u;
    const __amdgpu_buffer_rsrc_t rn_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.nbr, 0, (int)nbr_bytes, 0x00020000);
    const unsigned nb_off = ((unsigned)row_base + mslot) * 4u;
    const bool t8 = a.t8 != 0;
    // 8-wide transposed map: offset k belongs to ONE parity class (bit a set <-> component a of k is not the centre) and to
    // entry j8 = (component == +1) per axis; a row's entries carry its class in bits 28 .. 30
#define X3F_K8(KK, J8, CLS)                                                       \
  const unsigned kx_ = (unsigned)(KK) % 3u, ky_ = ((unsigned)(KK) / 3u) % 3u, kz_ = (unsigned)(KK) / 9u; \
  const unsigned CLS = (kx_ != 1u ? 1u : 0u) | (ky_ != 1u ? 2u : 0u) | (kz_ != 1u ? 4u : 0u);             \
  const unsigned J8 = (kx_ == 2u ? 1u : 0u) | (ky_ == 2u ? 2u : 0u) | (kz_ == 2u ? 4u : 0u);
#define X3F_LOADN(KK, E)                                                                                     \
  {                                                                                                          \
    X3F_K8(KK, j8_, cls_)                                                                                    \
    (void)cls_;                                                                                              \
    E = (int)__builtin_amdgcn_raw_buffer_load_b32(rn_, (int)nb_off, (int)((t8 ? j8_ : (unsigned)(KK)) * (unsigned)a.n_out * 4u), 0); \
  }
    // raw entry of offset KK -> byte offset of the row, or all ones
#define X3F_CONV1(KK, E, O)                                                                                  \
  {                                                                                                          \
    X3F_K8(KK, j8_, cls_)                                                                                    \
    (void)j8_;                                                                                               \
    const unsigned rw_ = t8 ? (unsigned)(E) & PP_ROW_MASK : (unsigned)(E);                                  \
    const unsigned bad_ = (unsigned)((E) >> 31) | tail1 | ((t8 && (((unsigned)(E) >> 28) & 7u) != cls_) ? 0xFFFFFFFFu : 0u); \
    O = (m24 ? __umul24(rw_, row_bytes) : rw_ * row_bytes) | bad_;                                           \
  }
    // the four line loads of group GG (32 channels = 128 bytes per row) of the rows OFF points at
#define X3F_GATHER(GG, OFF)                                                                                        \
  {                                                                                                                \
    const int sl_ = 2 * (GG);                                                                                      \
    const float* src_ = sl_ < S0 ? a.in0 + sl_ * 16 : a.in1 + (sl_ - S0) * 16;                                     \
    const unsigned long long ps_ = (unsigned long long)src_;                                                       \
    const u32x4_t dr_ = {(unsigned)ps_, (unsigned)(ps_ >> 32) & 0xFFFFu, a_bytes, 0x00020000u};                    \
    if (X3F_ABLATE != 1) _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
      const unsigned lds_ = rows_lds + (unsigned)j * 1024u;                                                        \
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"                     \
                   ::"v"(OFF[j]), "s"(lds_), "s"(dr_), "s"(0u) : "memory", "m0");                                  \
    }                                                                                                              \
  }
#define X3F_ADV(KI, GI, UR, OK)           \
  {                                       \
    if (GI + 1 < G) {                     \
      ++GI;                               \
    } else if (UR) {                      \
      KI = __builtin_ctz(UR);             \
      UR &= UR - 1u;                      \
      GI = 0;                             \
    } else {                              \
      OK = 0;                             \
    }                                     \
  }
#define X3F_SIX(ACC, PLN, B0, B1, B2)  \
  ACC = x3_mfma(B2, PLN.p0, ACC);      \
  ACC = x3_mfma(B0, PLN.p2, ACC);      \
  ACC = x3_mfma(B1, PLN.p1, ACC);      \
  ACC = x3_mfma(B1, PLN.p0, ACC);      \
  ACC = x3_mfma(B0, PLN.p1, ACC);      \
  ACC = x3_mfma(B0, PLN.p0, ACC);
    // fragment role: lane (i, q); row i of tile t = slot 16 t + i = block i & 3, row 4 t + (i >> 2); chunk C sits at C ^ s
    const unsigned fs = ((unsigned)i & 3u) + 4u * (((unsigned)i >> 3) & 1u);
    const f32x4* const rows = s_mem + 2 * WB + wave * 256;
    const int fr0 = (i & 3) * 64 + (i >> 2) * 8 + (int)(((unsigned)q) ^ fs);        // f32x4 index of chunk q (tile 0)
    const int fr1 = (i & 3) * 64 + (i >> 2) * 8 + (int)((4u + (unsigned)q) ^ fs);   // chunk 4 + q

    // the wave's own offset sequence: kd = the offset whose row offsets are in offc; en = raw map entries (slot = lane & 31) of the
    // wave's next offset kn = the lowest bit of Un, requested when kd was taken up (a whole step or more before they are used)
#define X3F_QB(O, J, CTRL) (((unsigned)__builtin_amdgcn_update_dpp(0, (int)(O), CTRL, 0xf, 0xf, true) | x16) ^ ((unsigned)(J) << 4))
#define X3F_PERM(O, OFF)            /* (all ones stay >= 2^32 - 64 under the chunk bits: out of range) */ \
  {                                                                                                       \
    OFF[0] = X3F_QB(O, 0, 0x00);                                                                          \
    OFF[1] = X3F_QB(O, 1, 0x55);                                                                          \
    OFF[2] = X3F_QB(O, 2, 0xAA);                                                                          \
    OFF[3] = X3F_QB(O, 3, 0xFF);                                                                          \
  }
    unsigned Un = Uw;
    int kd = -1;
    unsigned offc[4] = {X3_MISSING, X3_MISSING, X3_MISSING, X3_MISSING};
    int en = -1;
    if (Un) {
      const int ka = __builtin_ctz(Un);
      Un &= Un - 1u;
      int ea;
      X3F_LOADN(ka, ea);
      if (Un) X3F_LOADN(__builtin_ctz(Un), en);
      unsigned oa;
      X3F_CONV1(ka, ea, oa);
      X3F_PERM(oa, offc);
      kd = ka;
    }
    int k0 = __builtin_ctz(U), g0 = 0, buf = 0;
    unsigned Ur = U & (U - 1u);
    int k1 = k0, g1 = g0, ok1 = 1;
    X3F_ADV(k1, g1, Ur, ok1);
    X3F_STAGE_W(k0, g0, 0);
    if ((Uw >> k0) & 1u) X3F_GATHER(g0, offc);   // (k0 is the lowest offset of the union: if the wave has it, it is kd)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (;;) {
      const unsigned act0 = (m[0] >> k0) & 1u, act1 = (m[1] >> k0) & 1u;
      f32x4 AA[T][2];
      if (X3F_ABLATE == 6) {
        AA[0][0] = AA[0][1] = AA[1][0] = AA[1][1] = (f32x4){1.f, 2.f, 3.f, (float)lane};
      } else {
        if (act0) {
          AA[0][0] = rows[fr0];
          AA[0][1] = rows[fr1];
        }
        if (act1) {
          AA[1][0] = rows[fr0 + 32];
          AA[1][1] = rows[fr1 + 32];
        }
      }
      // the next step's rows belong to another offset: its entries (requested a step or more ago: whatever the compiler waits for
      // here landed before the last barrier; behind the staging below it would wait for the weight pieces just requested) become
      // byte offsets and are broadcast inside the lane quads; the entries of the wave's offset after that are requested
      const bool turn = X3F_ABLATE != 7 && ok1 && ((Uw >> k1) & 1u) && k1 != kd;
      if (turn) {
        unsigned on;
        X3F_CONV1(k1, en, on);
        X3F_PERM(on, offc);
        kd = k1;
        Un &= Un - 1u;
        if (Un) X3F_LOADN(__builtin_ctz(Un), en);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ok1 && X3F_ABLATE != 4) X3F_STAGE_W(k1, g1, buf ^ 1);
      // the first tile is split while the second tile's fragments are still on their way; once they are in registers the next
      // step's lines may overwrite the block (the wait is free by then: LDS reads return in order)
      X3Planes P0, P1;
      if (act0) P0 = MODE == 1 ? x3_round(AA[0][0], AA[0][1]) : x3_split(AA[0][0], AA[0][1]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ok1 && ((Uw >> k1) & 1u)) X3F_GATHER(g1, offc);
      if (act0 | act1) {
        if (act1) P1 = MODE == 1 ? x3_round(AA[1][0], AA[1][1]) : x3_split(AA[1][0], AA[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4* wb = s_wb + buf * WB;
        bf16x8_t Bf[2][PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) Bf[0][p] = __builtin_bit_cast(bf16x8_t, wb[p * 64 + lane]);
#ifndef X3F_SETPRIO
#define X3F_SETPRIO 0  // A/B builds: 1 = the wave raises its issue priority for the MFMA phase of a step (s_setprio 3 ... 0)
#endif
        if (X3F_SETPRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int jt = 0; jt < NTW; ++jt) {
          if (jt + 1 < NTW) {
#pragma unroll
            for (int p = 0; p < PL; ++p) Bf[(jt + 1) & 1][p] = __builtin_bit_cast(bf16x8_t, wb[(jt + 1) * (PL * 64) + p * 64 + lane]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (MODE == 1) {
            const bf16x8_t B0 = Bf[jt & 1][0];
            if (act0) acc[0][jt] = x3_mfma(B0, P0.p0, acc[0][jt]);
            if (act1) acc[1][jt] = x3_mfma(B0, P1.p0, acc[1][jt]);
          } else {
            const bf16x8_t B0 = Bf[jt & 1][0], B1 = Bf[jt & 1][PL > 1 ? 1 : 0], B2 = Bf[jt & 1][PL > 2 ? 2 : 0];
            if (act0) {
              X3F_SIX(acc[0][jt], P0, B0, B1, B2)
            }
            if (act1) {
              X3F_SIX(acc[1][jt], P1, B0, B1, B2)
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (X3F_SETPRIO) __builtin_amdgcn_s_setprio(0);
      }
      if (!ok1) break;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (X3F_ABLATE != 5) __syncthreads();
      k0 = k1;
      g0 = g1;
      X3F_ADV(k1, g1, Ur, ok1);
      buf ^= 1;
    }
#undef X3F_SIX
#undef X3F_ADV
#undef X3F_GATHER
#undef X3F_CONV1
#undef X3F_LOADN
#undef X3F_K8
#undef X3F_PERM
#undef X3F_QB
#undef X3F_STAGE_W
  }

  // ---- fused 1x1 shortcut and epilogue: as k_spconv_x3
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
      o2[tt] = r < a.n_out ? (unsigned)r * ds_row + (unsigned)q * 16u : X3_MISSING;
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
      if constexpr (MODE == 1) {
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
          if (a.split > 1) {
            *(f32x4*)(dst + e0 + jt * 16) = v;
            continue;
          }
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
        }
      }
    }
  }
}

static bool x3f_ok(const SpconvArgs& a);
bool pp_spconv_x3_ok(const SpconvArgs& a, int64_t n_in, int ntw) {
  if (a.c0 % 16 != 0 || (a.c1 != 0 && a.c1 != a.c0)) return false;
  if ((a.cout & 3) != 0 || a.K > 27 || a.K < 2 || ntw < (x3f_ok(a) ? 1 : 2) || ntw > 6 || n_in <= 0) return false;  // (one column tile: k_spconv_x3f only)
  const double S = (a.c0 + a.c1) / 16, G = ((a.c0 + a.c1) / 16 + 1) / 2;
  return (double)n_in * a.c0 * 4.0 < 4294967000.0 && (double)a.K * S * a.NT * 1024.0 < 4294967000.0 &&
         (double)a.K * G * a.NT * 3072.0 < 4294967000.0 && (double)a.n_out * 32.0 < 4294967000.0;
}

// full-line gathers through LDS (k_spconv_x3f): dense maps over whole 32-channel groups.  PP_CONV_X3F=0 or
// pp_spconv_x3_full_lines(0): never (A/B runs, the bit-identity test)
static int g_x3f = -1;
static bool x3f_ok(const SpconvArgs& a) {
  if (g_x3f < 0) g_x3f = getenv("PP_CONV_X3F") ? (atoi(getenv("PP_CONV_X3F")) != 0) : 1;
  return g_x3f && (a.t8 == 0 || a.t8 == 1) && a.nbr && a.c0 % 32 == 0 && (a.c1 == 0 || a.c1 == a.c0) &&
         (double)a.K * (double)a.n_out * 4.0 < 4294967000.0;
}
bool pp_spconv_x3f_ok(const SpconvArgs& a) { return x3f_ok(a); }
extern "C" int pp_spconv_x3_full_lines(int32_t mode) {
  if (g_x3f < 0) g_x3f = getenv("PP_CONV_X3F") ? (atoi(getenv("PP_CONV_X3F")) != 0) : 1;
  const int was = g_x3f;
  if (mode == 0 || mode == 1) g_x3f = mode;
  return was;
}

int pp_spconv_x3_launch(const SpconvArgs& a, int64_t n_in, int ntw, unsigned groups, hipStream_t s) {
  const unsigned a_bytes = (unsigned)((uint64_t)n_in * a.c0 * 4u);
  const unsigned w_bytes = (unsigned)((uint64_t)a.K * ((a.c0 + a.c1) / 16) * a.NT * 1024u);
  const unsigned flags = (n_in < (int64_t(1) << 24) ? 1u : 0u);
  if (x3f_ok(a)) {
    const unsigned nbr_bytes = (unsigned)((uint64_t)(a.t8 ? 8 : a.K) * (uint64_t)a.n_out * 4u);
    dim3 gridf(pp_blocks(a.n_out, 32 * X3_WPB), groups, (unsigned)(a.split > 1 ? a.split : 1));
#define X3F_LAUNCH(N, DSV, MD) \
  hipLaunchKernelGGL((k_spconv_x3f<N, DSV, MD>), gridf, dim3(64 * X3_WPB), 0, s, a, a_bytes, w_bytes, flags, nbr_bytes)
#define X3F_CASE(N)                                       \
  case N:                                                 \
    if (a.bf16) {                                         \
      if (a.ds_in) X3F_LAUNCH(N, true, 1);                \
      else X3F_LAUNCH(N, false, 1);                       \
    } else {                                              \
      if (a.ds_in) X3F_LAUNCH(N, true, 0);                \
      else X3F_LAUNCH(N, false, 0);                       \
    }                                                     \
    break;
    switch (ntw) {
      X3F_CASE(1) X3F_CASE(2) X3F_CASE(3) X3F_CASE(4) X3F_CASE(5) X3F_CASE(6)
      default: pp_set_error("pp_spconv_x3: ntw %d out of range", ntw); return PP_ERR_INVALID;
    }
#undef X3F_CASE
#undef X3F_LAUNCH
    return PP_OK;
  }
  // (64 rows per wave -- T = 4 -- on the two-column-tile launches was built and measured in round 5 and lost: 32->32 at 5.4 M rows
  // 1630 us with 32 rows per wave, 1723 with 64; 64->32 2686 / 2953; 96->32 3798 / 4268, profiles/r05_ab_x3_two_tiles.txt: half the
  // staging and barriers per row do not pay for the third of the occupancy.  The instantiation was removed in round 6.)
  const int R = 32;
  dim3 grid(pp_blocks(a.n_out, R * X3_WPB), groups, (unsigned)(a.split > 1 ? a.split : 1));
#define X3_LAUNCH(N, DSV, MD, TT) \
  hipLaunchKernelGGL((k_spconv_x3<N, DSV, MD, TT>), grid, dim3(64 * X3_WPB), 0, s, a, a_bytes, w_bytes, flags)
#define X3_CASE(N)                                        \
  case N:                                                 \
    if (a.bf16) {                                         \
      if (a.ds_in) X3_LAUNCH(N, true, 1, 2);              \
      else X3_LAUNCH(N, false, 1, 2);                     \
    } else {                                              \
      if (a.ds_in) X3_LAUNCH(N, true, 0, 2);              \
      else X3_LAUNCH(N, false, 0, 2);                     \
    }                                                     \
    break;
  switch (ntw) {
    X3_CASE(2) X3_CASE(3) X3_CASE(4) X3_CASE(5) X3_CASE(6)
    default: pp_set_error("pp_spconv_x3: ntw %d out of range", ntw); return PP_ERR_INVALID;
  }
#undef X3_CASE
#undef X3_LAUNCH
  return PP_OK;
}
