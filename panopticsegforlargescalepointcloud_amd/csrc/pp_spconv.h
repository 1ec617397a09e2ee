// Shared between the dense-offset convolution kernels (pp_spconv.hip, pp_spconv2.hip).
#pragma once
#include "pp_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// 4 floats -> 4 bfloat16 (round to nearest even; two v_cvt_pk_bf16_f32), packed as the A / B operand of
// v_mfma_f32_16x16x16_bf16: lane (i, q) supplies k = 4q .. 4q+3
__device__ __forceinline__ s16x4 pp_bf16x4(f32x4 v) {
  const bf16x2 lo = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2);
  const bf16x2 hi = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2);
  const u32x2 p = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
  return __builtin_bit_cast(s16x4, p);
}

// 2 x 4 floats -> 8 bfloat16 = the A / B operand of v_mfma_f32_16x16x32_bf16 (lane (i, q) supplies k = 8q .. 8q+7)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t pp_bf16x8(f32x4 lo, f32x4 hi) {
  const s16x4 a = pp_bf16x4(lo), b = pp_bf16x4(hi);
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

struct SpconvArgs {
  const float* in0;
  const float* in1;
  const float* wp;
  const int32_t* nbr;
  const float* scale;
  const float* shift;
  const float* residual;
  const int32_t* row_order;  // optional processing order of the output rows (tile schedule); results do not depend on it
  float* out;
  int64_t n_out;
  int c0, c1, K, cout, NT, relu;
  int bf16;  // operands rounded to bfloat16 in registers, fp32 accumulation (v3 kernel only)
  // split-K (v3 kernel only): blockIdx.z owns the kernel offsets [z*K/split, (z+1)*K/split) and writes raw partial sums
  // to part[z][row][col]; k_spconv_split_reduce adds them in a fixed order and applies the epilogue
  int split;
  float* part;
  // fused 1x1 shortcut of a residual block (v3 kernel, same-level maps, unsplit): out += (ds_in @ ds_wp) * ds_scale + ds_shift
  // after the ReLU -- the block's "downsample" branch (1x1 convolution + BatchNorm of the block's input) computed by the
  // block's last convolution instead of a launch of its own that writes a tensor this kernel reads back
  const float* ds_in;
  const float* ds_wp;
  const float* ds_scale;
  const float* ds_shift;
  int ds_c;
  // 8-wide transposed map (pp_kernel_map_transpose8; v3 kernel only): nbr is [8][n_out] slot-major, every entry = coarse row |
  // parity class of the fine row << 28 -- the prologue expands (class, j) to the offset index k
  int t8;
  // compact same-level map (pp_map_compact; t8 == 2, v3 kernel only): nbr = the present entries (neighbour rows) grouped by chunks
  // of 32 output rows, cm_start [chunks + 1] their offsets, cm_tag[e] = offset index << 6 | output row & 63, cm_mask [n_out] the
  // rows' offset masks -- 4 + 6 x pairs bytes per row instead of 108
  const uint32_t* cm_mask;
  const int32_t* cm_start;
  const uint16_t* cm_tag;
};
#define PP_ROW_MASK 0x0FFFFFFFu  // row part of a T8 map entry (bits 28..30: parity class)

// XCD-aware block remap: the dispatcher places block b on XCD b % 8; give every XCD a CONTIGUOUS range of row
// blocks so the neighbour rows gathered by adjacent blocks hit that XCD's private L2 (guide T1; speed only).
__device__ inline unsigned pp_xcd_remap(unsigned b, unsigned n) {
  const unsigned q = n >> 3, r = n & 7u, x = b & 7u, j = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// pipelined variant (pp_spconv2.hip); mode16 only, K <= 28
// VALU-free main loop (buffer loads); needs the input row count for the descriptor
bool pp_spconv_fwd3_ok(const SpconvArgs& a, int64_t n_in);
int pp_spconv_fwd3_launch(const SpconvArgs& a, int64_t n_in, int ntw, unsigned groups, int T, int depth, hipStream_t s);
int pp_spconv_split_reduce_launch(const SpconvArgs& a, hipStream_t s);
// wide layers on the bf16 matrix pipe with exactly split fp32 operands (pp_spconv3.hip)
bool pp_spconv_x3_ok(const SpconvArgs& a, int64_t n_in, int ntw);
int pp_spconv_x3_launch(const SpconvArgs& a, int64_t n_in, int ntw, unsigned groups, hipStream_t s);
// does pp_spconv_x3_launch run this launch on k_spconv_x3f (full-line gathers through LDS)?  (a.nbr, a.t8, channels, sizes)
bool pp_spconv_x3f_ok(const SpconvArgs& a);

// pipelined weight gradient (pp_spconv_bww.hip); 32-bit buffer offsets over the input rows
bool pp_spconv_bww2_ok(int cin, int cout, int64_t n_in, const int32_t* nbr);
int pp_spconv_bww2_launch(const float* in, int cin, int64_t n_in, const float* dout, int cout, const int32_t* nbr,
                          int K, int64_t n_out, float* dw, int bf16, hipStream_t s);
