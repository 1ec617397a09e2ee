// K4/K5: sparse convolution on CDNA4 matrix cores (exact-f32 MFMA, v_mfma_f32_16x16x4_f32).
//
// Output-stationary implicit GEMM: a wave owns 32 output rows x (NTW*16) output channels and walks the
// kernel offsets k and input-channel steps; its A fragments are GATHERED straight from the input rows
// named by the kernel map (float4 per lane = 16 B, a full 64 B row segment per 4 lanes), its B fragments
// are read from weights pre-packed into MFMA fragment order (one coalesced 1 KiB load per wave).
// No atomics, no scatter: every output row is written exactly once, in a fixed summation order, with the
// folded BatchNorm / ReLU / residual epilogue fused (reference: MinkowskiConvolution -> MinkowskiBatchNorm
// -> MinkowskiReLU, torch_points3d/modules/MinkowskiEngine/api_modules.py:30-54,76-82,259-270; ME.cat of
// the skip connection api_modules.py:308 is the second source pointer).
//
// MFMA operand maps (guide 3, 16x16x4 f32): A lane l -> A[i=l&15][kk=l>>4]; B lane l -> B[kk=l>>4][j=l&15];
// D reg r of lane l -> D[row=4*(l>>4)+r][col=l&15].  With a float4 per lane, MFMA #t of a 16-channel
// step consumes channels {t, 4+t, 8+t, 12+t}: lane (i,q) feeds A[i][4q+t], lane (j,q) feeds W[4q+t][j].
#include <stdlib.h>
#include <string.h>

#include "pp_spconv.h"

// ---------------------------------------------------------------------------------------------
// weight packing
//   mode16 (cin % 16 == 0): packed[k][s][jt][lane][t] = W[k][16s + 4(lane>>4) + t][16jt + (lane&15)]
//   mode4  (otherwise, cin % 4 == 0): packed[k][s4][jt][lane] = W[k][4s4 + (lane>>4)][16jt + (lane&15)]
//   transpose_w: W[k][a][b] read as W[k][b][a] (cin/cout given for the PACKED orientation)
// ---------------------------------------------------------------------------------------------
static inline int pp_nt(int cout) { return (cout + 15) / 16; }

__host__ __device__ static inline size_t packed_floats(int K, int cin, int cout) {
  const int NT = (cout + 15) / 16;
  if (cin % 16 == 0) return (size_t)K * (cin / 16) * NT * 256;
  return (size_t)K * ((cin + 3) / 4) * NT * 64;
}
// mode16 weights carry a second section behind the fp32 one (pp_spconv3.hip): the same fragments split into three bfloat16
// planes (w == hi + mid + lo exactly), two 16-channel steps per group g (the second one zero when the step count is odd):
//   x3[k][g][jt][plane][lane][t8] = plane of packed[k][2g + (t8 >> 2)][jt][lane][t8 & 3]        (768 floats per (k, g, jt))
__host__ __device__ static inline size_t packed_x3_floats(int K, int cin, int cout) {
  if (cin % 16 != 0) return 0;
  const int NT = (cout + 15) / 16, G = (cin / 16 + 1) / 2;
  return (size_t)K * G * NT * 768;
}
// ... and a third one: the same fragments rounded to nearest-even bfloat16 (bfloat16 compute of the same kernel):
//   b16[k][g][jt][lane][t8] = bf16(packed[k][2g + (t8 >> 2)][jt][lane][t8 & 3])                 (256 floats per (k, g, jt))
__host__ __device__ static inline size_t packed_total_floats(int K, int cin, int cout) {
  const size_t x3 = packed_x3_floats(K, cin, cout);
  return packed_floats(K, cin, cout) + x3 + x3 / 3;
}
extern "C" size_t pp_packed_weight_floats(int32_t K, int32_t cin, int32_t cout) { return packed_total_floats(K, cin, cout); }

// element e of the packed tensor of one layer (shared by the single-layer and the batched kernel)
__device__ __forceinline__ float pack_weight_element(const float* __restrict__ w, int K, int cin, int cout, int transpose_w, int64_t e) {
  int NT = (cout + 15) / 16;
  int k, ci, co;
  if (cin % 16 == 0) {
    int t = (int)(e & 3);
    int lane = (int)((e >> 2) & 63);
    int64_t r = e >> 8;
    int jt = (int)(r % NT);
    r /= NT;
    int S = cin / 16;
    int s = (int)(r % S);
    k = (int)(r / S);
    ci = 16 * s + 4 * (lane >> 4) + t;
    co = 16 * jt + (lane & 15);
  } else {
    int lane = (int)(e & 63);
    int64_t r = e >> 6;
    int jt = (int)(r % NT);
    r /= NT;
    int S4 = (cin + 3) / 4;
    int s4 = (int)(r % S4);
    k = (int)(r / S4);
    ci = 4 * s4 + (lane >> 4);
    co = 16 * jt + (lane & 15);
  }
  float v = 0.f;
  if (ci < cin && co < cout) {
    // source layout: not transposed [K][cin][cout]; transposed (bit 0): the ME tensor is [K][cout][cin];
    // bit 1: offsets in reverse order (packed[k] = W_{K-1-k}: the mirrored map of a same-level convolution)
    const int64_t ks = (transpose_w & 2) ? K - 1 - k : k;
    v = (transpose_w & 1) ? w[(ks * cout + co) * cin + ci] : w[(ks * cin + ci) * cout + co];
  }
  return v;
}
// float slot f of the pre-split section: two bfloat16 (t8 = 2 (f & 3), + 1) of plane (f >> 8) % 3
__device__ __forceinline__ float pack_weight_x3_element(const float* __restrict__ w, int K, int cin, int cout, int transpose_w, int64_t f) {
  const int NT = (cout + 15) / 16, S = cin / 16, G = (S + 1) / 2;
  const int64_t nx3 = (int64_t)packed_x3_floats(K, cin, cout);
  const bool rne = f >= nx3;  // the bfloat16 section: one plane per (k, g, jt)
  if (rne) f -= nx3;
  const int tp = (int)(f & 3), lane = (int)((f >> 2) & 63), plane = rne ? 3 : (int)((f >> 8) % 3);
  int64_t r = rne ? f / 256 : f / 768;
  const int jt = (int)(r % NT);
  r /= NT;
  const int g = (int)(r % G), k = (int)(r / G);
  unsigned out = 0;
  for (int u = 0; u < 2; ++u) {
    const int t8 = 2 * tp + u, s = 2 * g + (t8 >> 2);
    float x = 0.f;
    if (s < S) x = pack_weight_element(w, K, cin, cout, transpose_w, ((((int64_t)k * S + s) * NT + jt) * 64 + lane) * 4 + (t8 & 3));
    const float h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u);
    const float rr = x - h;
    const float m = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rr) & 0xFFFF0000u);
    const float l = rr - m;
    unsigned bits = __builtin_bit_cast(unsigned, plane == 0 ? h : (plane == 1 ? m : l)) >> 16;
    if (plane == 3) {  // round to nearest even (finite inputs), as v_cvt_pk_bf16_f32 rounds the gathered rows
      const unsigned u32 = __builtin_bit_cast(unsigned, x);
      bits = (u32 + 0x7FFFu + ((u32 >> 16) & 1u)) >> 16;
    }
    out |= bits << (16 * u);
  }
  return __builtin_bit_cast(float, out);
}
__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ w, int K, int cin, int cout,
                                                     int transpose_w, float* __restrict__ packed, int64_t total) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t n32 = (int64_t)packed_floats(K, cin, cout);
  packed[e] = e < n32 ? pack_weight_element(w, K, cin, cout, transpose_w, e) : pack_weight_x3_element(w, K, cin, cout, transpose_w, e - n32);
}

extern "C" int pp_pack_weight(const float* weight, int32_t K, int32_t cin, int32_t cout, int32_t transpose_w,
                              float* packed, pp_stream_t stream) {
  PP_REQUIRE(weight && packed, "pp_pack_weight: null pointer");
  PP_REQUIRE(cin % 4 == 0, "pp_pack_weight: cin must be a multiple of 4");
  int64_t total = (int64_t)pp_packed_weight_floats(K, cin, cout);
  hipLaunchKernelGGL(k_pack_weight, dim3(pp_blocks(total, 256)), dim3(256), 0, pp_s(stream), weight, K, cin, cout,
                     transpose_w, packed, total);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// every layer of a model in ONE launch (training: all weights change with the optimizer step, i.e. ~190 packings of a few
// microseconds each per step otherwise).  desc[l] = {weight pointer, packed pointer, K, cin, cout, flags}, first_block[l] =
// number of 256-element blocks of the layers in front of l (first_block[n] = grid size).
__global__ __launch_bounds__(256) void k_pack_weights_batched(const int64_t* __restrict__ desc, const int64_t* __restrict__ first_block,
                                                              int n_desc) {
  const int64_t b = blockIdx.x;
  int lo = 0, hi = n_desc;  // last l with first_block[l] <= b
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first_block[mid] <= b) lo = mid; else hi = mid;
  }
  const int64_t* d = desc + 6 * (int64_t)lo;
  const int K = (int)d[2], cin = (int)d[3], cout = (int)d[4], flags = (int)d[5];
  const int64_t n32 = (int64_t)packed_floats(K, cin, cout), total = (int64_t)packed_total_floats(K, cin, cout);
  const int64_t e = (b - first_block[lo]) * 256 + threadIdx.x;
  if (e < total)
    ((float*)d[1])[e] = e < n32 ? pack_weight_element((const float*)d[0], K, cin, cout, flags, e)
                                : pack_weight_x3_element((const float*)d[0], K, cin, cout, flags, e - n32);
}
extern "C" int pp_pack_weights_batched(const int64_t* desc, const int64_t* first_block, int32_t n_desc, int64_t total_blocks,
                                       pp_stream_t stream) {
  if (n_desc == 0 || total_blocks == 0) return PP_OK;
  PP_REQUIRE(desc && first_block && n_desc > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "pp_pack_weights_batched: bad arguments");
  hipLaunchKernelGGL(k_pack_weights_batched, dim3((unsigned)total_blocks), dim3(256), 0, pp_s(stream), desc, first_block, n_desc);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// forward kernel.  block = 4 waves = 128 output rows; grid.y = output-channel tile groups.
// ---------------------------------------------------------------------------------------------
template <int NTW, bool MODE16>
__global__ __launch_bounds__(256) void k_spconv_fwd(SpconvArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const unsigned bid = pp_xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row_base = ((int64_t)bid * 4 + wave) * 32;
  if (row_base >= a.n_out) return;  // wave-uniform
  const int jt0 = blockIdx.y * NTW;
  const int64_t r0 = row_base + i, r1 = row_base + 16 + i;
  const bool v0 = r0 < a.n_out, v1 = r1 < a.n_out;

  f32x4 acc[2][NTW];
#pragma unroll
  for (int jt = 0; jt < NTW; ++jt) {
    acc[0][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int cin = a.c0 + a.c1;
  // neighbour indices are prefetched one offset ahead (breaks the index -> gather -> MFMA dependency chain)
  int a0 = -1, a1 = -1;
  if (a.nbr) {
    if (v0) a0 = a.nbr[r0];
    if (v1) a1 = a.nbr[r1];
  } else {
    if (v0) a0 = (int)r0;
    if (v1) a1 = (int)r1;
  }
  for (int k = 0; k < a.K; ++k) {
    int a0n = -1, a1n = -1;
    if (k + 1 < a.K) {
      if (v0) a0n = a.nbr[(int64_t)(k + 1) * a.n_out + r0];
      if (v1) a1n = a.nbr[(int64_t)(k + 1) * a.n_out + r1];
    }
    // per-16-row-tile skip: with Morton-ordered rows a tile is a compact surface patch, so whole offsets are empty
    const bool act0 = __ballot(a0 >= 0) != 0ull;
    const bool act1 = __ballot(a1 >= 0) != 0ull;
    if (act0 || act1) {
      if (MODE16) {
        const int S0 = a.c0 >> 4, S = cin >> 4;
        const float* wk = a.wp + (int64_t)k * S * a.NT * 256;
        for (int s = 0; s < S; ++s) {
          const float* src;
          int cs, ss;
          if (s < S0) {
            src = a.in0; cs = a.c0; ss = s;
          } else {
            src = a.in1; cs = a.c1; ss = s - S0;
          }
          f32x4 A0 = (f32x4){0.f, 0.f, 0.f, 0.f}, A1 = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (a0 >= 0) A0 = *(const f32x4*)(src + (int64_t)a0 * cs + ss * 16 + q * 4);
          if (a1 >= 0) A1 = *(const f32x4*)(src + (int64_t)a1 * cs + ss * 16 + q * 4);
          const float* ws = wk + ((int64_t)s * a.NT + jt0) * 256 + lane * 4;
          if (act0 && act1) {
#pragma unroll
            for (int jt = 0; jt < NTW; ++jt) {
              if (jt0 + jt < a.NT) {
                f32x4 B = *(const f32x4*)(ws + jt * 256);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                  acc[0][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[t], B[t], acc[0][jt], 0, 0, 0);
                  acc[1][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[t], B[t], acc[1][jt], 0, 0, 0);
                }
              }
            }
          } else if (act0) {
#pragma unroll
            for (int jt = 0; jt < NTW; ++jt) {
              if (jt0 + jt < a.NT) {
                f32x4 B = *(const f32x4*)(ws + jt * 256);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                  acc[0][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[t], B[t], acc[0][jt], 0, 0, 0);
              }
            }
          } else {
#pragma unroll
            for (int jt = 0; jt < NTW; ++jt) {
              if (jt0 + jt < a.NT) {
                f32x4 B = *(const f32x4*)(ws + jt * 256);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                  acc[1][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[t], B[t], acc[1][jt], 0, 0, 0);
              }
            }
          }
        }
      } else {
        const int S4 = (cin + 3) >> 2;
        const float* wk = a.wp + (int64_t)k * S4 * a.NT * 64;
        for (int s = 0; s < S4; ++s) {
          const int ch = 4 * s + q;
          float A0 = 0.f, A1 = 0.f;
          if (ch < cin) {
            const float* src = ch < a.c0 ? a.in0 : a.in1;
            const int cs = ch < a.c0 ? a.c0 : a.c1;
            const int cc = ch < a.c0 ? ch : ch - a.c0;
            if (a0 >= 0) A0 = src[(int64_t)a0 * cs + cc];
            if (a1 >= 0) A1 = src[(int64_t)a1 * cs + cc];
          }
          const float* ws = wk + ((int64_t)s * a.NT + jt0) * 64 + lane;
#pragma unroll
          for (int jt = 0; jt < NTW; ++jt) {
            if (jt0 + jt < a.NT) {
              float B = ws[jt * 64];
              acc[0][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, B, acc[0][jt], 0, 0, 0);
              acc[1][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, B, acc[1][jt], 0, 0, 0);
            }
          }
        }
      }
    }
    a0 = a0n;
    a1 = a1n;
  }

  // epilogue: lane (col = i, row group = q) holds rows 4q+r of each 16-row tile
#pragma unroll
  for (int jt = 0; jt < NTW; ++jt) {
    const int col = (jt0 + jt) * 16 + i;
    if (jt0 + jt < a.NT && col < a.cout) {
      const float sc = a.scale ? a.scale[col] : 1.f;
      const float sh = a.shift ? a.shift[col] : 0.f;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
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

template <bool MODE16>
static void launch_fwd(int ntw, dim3 grid, hipStream_t s, const SpconvArgs& a) {
  switch (ntw) {
    case 1: hipLaunchKernelGGL((k_spconv_fwd<1, MODE16>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_spconv_fwd<2, MODE16>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_spconv_fwd<3, MODE16>), grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_spconv_fwd<4, MODE16>), grid, dim3(256), 0, s, a); break;
    case 5: hipLaunchKernelGGL((k_spconv_fwd<5, MODE16>), grid, dim3(256), 0, s, a); break;
    case 6: hipLaunchKernelGGL((k_spconv_fwd<6, MODE16>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((k_spconv_fwd<7, MODE16>), grid, dim3(256), 0, s, a); break;
  }
}

// caller-provided scratch for the split-K path of small launches (pp_spconv_set_scratch)
static float* g_scratch = nullptr;
static size_t g_scratch_bytes = 0;
extern "C" int pp_spconv_set_scratch(void* scratch, size_t bytes) {
  g_scratch = (float*)scratch;
  g_scratch_bytes = scratch ? bytes : 0;
  return PP_OK;
}

// Which launches take the split-operand kernel k_spconv_x3 (pp_spconv3.hip), and with how many column tiles per wave: 0 = none
// (the fp32-MFMA kernel k_spconv_fwd3 runs the launch).  The rule: cin % 16 == 0, 2 <= K <= 27, >= PP_CONV_X3_MIN_NTW (default 1; one tile only where k_spconv_x3f serves the shape)
// sixteen-column tiles per wave and -- on launches with <= 2 column tiles -- >= 32 input channels.  PP_CONV_X3=0: never.
//   column tiles per wave: 4, or up to 6 where that saves a column group (80 / 96 / 160 / 192 output channels: every group
//   gathers AND splits the rows again) -- 96->96 transposed onto 5.4 M rows 2942 -> 2348 us, 96->96 at 2.4 M rows 5566 -> 4662,
//   192->80 at 0.15 M rows 842 -> 656, 80->80 384 -> 333, 160->160 928 -> 799 (140 - 152 VGPRs and 45 - 51 KiB of LDS: three
//   workgroups per CU instead of four); launches below 0.4 M rows stop at 5 (96->96 at 32 k rows 118 -> 135 us with 6), the
//   fused-shortcut form at 4 (its second accumulator set would leave two waves per SIMD).  PP_CONV_X3_MAX_NTW = 4 .. 6 forces
//   the bound (A/B runs, profiles/r05_ab_x3_ntw.txt).
//   two column tiles per wave (the 32-channel layers): 10 - 12 % faster than the fp32-MFMA kernel layer by layer (32->32 at 5.4 M
//   rows 1793 -> 1630 us, 64->32 2991 -> 2686, 96->32 4254 -> 3798) and 1.1 ms per step (118.8 -> 117.7, three alternating pairs,
//   profiles/r05_ab_x3_two_tiles.txt); one column tile (the 16-channel layers) stays on the fp32 MFMAs: texture-path bound
//   (... with >= 32 input channels: a 16-channel input is ONE half-empty group per offset -- 16->32 at 5.1 M rows 837 us on the
//   fp32 MFMAs, 1162 on the split kernel; one column tile, the 16-channel outputs: 16->16 1018 / 1601 us, 64->16 and 32->16 even)
//   round 6: with the rows gathered as full lines through LDS (k_spconv_x3f; whole 32-channel groups on dense / 8-wide maps) ONE
//   column tile wins as well -- 64->16 at 10.3 M rows 3434 -> 3002 us, 32->16 1722 -> 1565, 64->16 at 5.4 M rows 2190 -> 1765
//   (profiles/r06_x3f_ntw1.txt) -- so PP_CONV_X3_MIN_NTW defaults to 1; pp_spconv_x3_ok admits one tile only on that kernel
static int x3_column_tiles(const SpconvArgs& a, int64_t n_in, bool shortcut) {
  static const int env_x3 = getenv("PP_CONV_X3") ? atoi(getenv("PP_CONV_X3")) : 1;
  static const int env_x3_ntw = getenv("PP_CONV_X3_MIN_NTW") ? atoi(getenv("PP_CONV_X3_MIN_NTW")) : 1;
  static const int env_x3_max = getenv("PP_CONV_X3_MAX_NTW") ? atoi(getenv("PP_CONV_X3_MAX_NTW")) : 0;
  if (!env_x3 || (a.c0 + a.c1) % 16 != 0) return 0;
  const int mx = shortcut ? 4 : (env_x3_max >= 4 && env_x3_max <= 6 ? env_x3_max : (a.n_out >= 400000 ? 6 : 5));
  int g4 = (a.NT + 3) / 4;
  if ((a.NT + mx - 1) / mx < g4) g4 = (a.NT + mx - 1) / mx;
  const int n4 = (a.NT + g4 - 1) / g4;
  const bool thin_in = n4 <= 2 && (a.c0 + a.c1) < 32;
  return (n4 >= env_x3_ntw && !thin_in && pp_spconv_x3_ok(a, n_in, n4)) ? n4 : 0;
}

/* Which kernel pp_spconv_fwd / _shortcut / _t8 run a launch of this shape on: 1 = k_spconv_x3 (split-operand, bf16 matrix pipe),
 * 2 = k_spconv_x3f (the same arithmetic, rows gathered as full lines through LDS), 0 = k_spconv_fwd3 / k_spconv_fwd (fp32 MFMAs).  The same function the dispatch calls -- for profilers that attribute launch times
 * to a kernel family (ops.LaunchProfiler) without mirroring the rule. */
extern "C" int pp_spconv_kernel_family(int32_t c0, int32_t c1, int64_t n_in, int32_t K, int64_t n_out, int32_t cout,
                                       int32_t shortcut) {
  if (c0 <= 0 || c1 < 0 || n_out <= 0) return 0;
  SpconvArgs a;
  memset(&a, 0, sizeof(a));
  a.n_out = n_out; a.c0 = c0; a.c1 = c1; a.K = K; a.cout = cout; a.NT = pp_nt(cout);
  a.nbr = (const int32_t*)(uintptr_t)64;  // (a map is given: K >= 2 launches always have one; the full-line form of the split kernel asks)
  const bool mode16 = ((c0 + c1) % 16 == 0) && c0 % 16 == 0;
  if (!mode16 || K > 28 || !pp_spconv_fwd3_ok(a, n_in)) return 0;
  if (!x3_column_tiles(a, n_in, shortcut != 0)) return 0;
  return pp_spconv_x3f_ok(a) ? 2 : 1;  // (on a dense or 8-wide map; the compact form is never handed to such a launch)
}

// rows_per_wave (0 | 32 | 64), pipeline (0 | 1 | 3) and split_k (0 | 1 | 2 | 4 | 8) select the variant of the pipelined
// kernel explicitly; 0 = the per-shape choice below.  The parity tests drive every variant through pp_spconv_fwd_ex.
static int spconv_fwd_impl(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in,
                           const float* packed_weight, const int32_t* nbr, int32_t K, int64_t n_out, int32_t cout,
                           const float* scale, const float* shift, int32_t relu, const float* residual,
                           const int32_t* row_order, float* out, int bf16, int rows_per_wave, int pipeline, int split_k,
                           pp_stream_t stream, const float* ds_in = nullptr, int32_t ds_c = 0,
                           const float* ds_packed = nullptr, const float* ds_scale = nullptr,
                           const float* ds_shift = nullptr, int t8 = 0, const uint32_t* cm_mask = nullptr,
                           const int32_t* cm_start = nullptr, const uint16_t* cm_tag = nullptr) {
  PP_REQUIRE(in0 && packed_weight && out, "pp_spconv_fwd: null pointer");
  PP_REQUIRE(c0 > 0 && c1 >= 0 && (c1 == 0 || in1), "pp_spconv_fwd: bad channel split");
  PP_REQUIRE(nbr || K == 1, "pp_spconv_fwd: a kernel map is required unless K == 1");
  PP_REQUIRE((c0 + c1) % 4 == 0, "pp_spconv_fwd: cin must be a multiple of 4");
  PP_REQUIRE(rows_per_wave == 0 || rows_per_wave == 32 || rows_per_wave == 64, "pp_spconv_fwd_ex: rows_per_wave in {0, 32, 64}");
  PP_REQUIRE(pipeline == 0 || pipeline == 1 || pipeline == 3 || pipeline == 5 || pipeline == 6, "pp_spconv_fwd_ex: pipeline in {0, 1, 3, 5, 6}");
  PP_REQUIRE(split_k == 0 || split_k == 1 || split_k == 2 || split_k == 4 || split_k == 8, "pp_spconv_fwd_ex: split_k in {0, 1, 2, 4, 8}");
  const bool mode16 = ((c0 + c1) % 16 == 0);
  if (mode16) PP_REQUIRE(c0 % 16 == 0, "pp_spconv_fwd: with cin % 16 == 0 both sources must be multiples of 16");
  if (n_out == 0) return PP_OK;
  SpconvArgs a;
  a.in0 = in0; a.in1 = in1; a.wp = packed_weight; a.nbr = nbr; a.scale = scale; a.shift = shift;
  a.residual = residual; a.row_order = row_order; a.out = out; a.n_out = n_out; a.c0 = c0; a.c1 = c1; a.K = K; a.cout = cout;
  a.NT = pp_nt(cout); a.relu = relu; a.bf16 = bf16; a.split = 1; a.part = nullptr;
  a.ds_in = nullptr; a.ds_wp = nullptr; a.ds_scale = nullptr; a.ds_shift = nullptr; a.ds_c = 0;
  a.t8 = t8;
  a.cm_mask = cm_mask; a.cm_start = cm_start; a.cm_tag = cm_tag;
  if (t8 == 1) PP_REQUIRE(row_order && nbr && K == 27 && !ds_in, "pp_spconv_fwd_t8: needs the 8-wide map, its slot order and K = 27");
  if (t8 == 2)
    PP_REQUIRE(!row_order && nbr && cm_mask && cm_start && cm_tag && K == 27 && n_in == n_out,
               "pp_spconv_fwd_cmap: needs the compact form of a same-level map (pp_map_compact), K = 27 and no slot order");
  const bool c4 = c0 == 4 && c1 == 0;  // the input layer has its own form of the pipelined kernel
  // column tiles per wave: 4, or up to 6 where that saves a column group on a LARGE launch (80 / 96 / 160 / 192 output
  // channels: every group gathers the input rows again) -- 96->96 transposed onto 5.4 M rows 3814 -> 3400 us, 160->160 onto
  // 0.67 M rows 1395 -> 1191 us; launches below ~0.4 M rows lose (fewer waves: 80->80 at 149 k rows 487 -> 527 us), 128
  // channels are two groups of 4 either way (profiles/r03_ab_ntw.log).  PP_MAX_NTW = 4 .. 6 forces the bound (A/B runs).
  static const int env_ntw = getenv("PP_MAX_NTW") ? atoi(getenv("PP_MAX_NTW")) : 0;
  int max_ntw = (mode16 || c4) ? 4 : 7;
  const int want_ntw = env_ntw >= 4 && env_ntw <= 6 ? env_ntw : (n_out >= 400000 ? 6 : 4);
  if (mode16 && (a.NT + want_ntw - 1) / want_ntw < (a.NT + 3) / 4) max_ntw = want_ntw;
  int groups = (a.NT + max_ntw - 1) / max_ntw;
  int ntw = (a.NT + groups - 1) / groups;
  groups = (a.NT + ntw - 1) / ntw;
  const bool v3_ok = (mode16 || c4) && K <= 28 && pp_spconv_fwd3_ok(a, n_in);
  if (v3_ok) {
    // rows per wave: 64 on launches with <= 2 column tiles per wave and >= 2 M rows -- narrow layers are bound by per-step
    // work, which 64 rows halve per row -- 32 on wider ones (the extra accumulators cost occupancy: 48->48 660 vs 690 us)
    // and on smaller ones (halving the number of waves costs more: one rank's share at 8 GPUs 34.1 vs 34.6 ms)
    const int T = rows_per_wave ? rows_per_wave / 16 : (ntw <= 2 && n_out >= 2000000 ? 4 : 2);
    // loop form: 3 (one step of loads in flight, load side advanced early) on <= 2 column tiles; 5 = the depth-3 register ring
    // (two steps in flight, inline-asm loads with hand-counted waits) on same-level / strided launches with 3 - 4 column tiles
    // per wave: 48->48 1.90 -> 1.79 ms, 64->64 1.02 -> 0.97, 128->48 4.97 -> 4.73, 80->80 0.48 -> 0.42; the <= 32-channel
    // layers LOSE 3 - 7 % with it (they are bound by the texture path's gather rate, not by latency) and so do the transposed
    // launches (profiles/r04_ab_ring.txt).  PP_CONV_RING=0 / 1: never / wherever it is instantiated (A/B runs).
    static const int env_ring = getenv("PP_CONV_RING") ? atoi(getenv("PP_CONV_RING")) : -1;
    int depth = pipeline ? pipeline : (ntw <= 2 ? 3 : 1);
    if (!pipeline && !c4 && ntw <= 4) {
      if (env_ring == 1 || (env_ring < 0 && ntw >= 3 && !row_order)) depth = 5;
    }
    // small launches (fewer waves than SIMD slots) are bound by the latency of one wave's walk over the 27 offsets:
    // split the offsets over up to 8 waves and add the partials in a fixed order (deterministic)
    const int64_t waves = ((n_out + 16 * T - 1) / (16 * T)) * groups;
    int sk = 1;
    if (split_k > 1)
      sk = split_k;
    else if (split_k == 0 && K >= 8 && g_scratch && waves < 1536) {
      sk = 2;
      while (sk < 8 && waves * sk < 2048) sk *= 2;
    }
    if (sk > 1) {
      if (g_scratch && (size_t)sk * (size_t)n_out * (size_t)cout * sizeof(float) <= g_scratch_bytes && K >= sk) {
        a.split = sk;
        a.part = g_scratch;
      } else if (split_k > 1) {
        pp_set_error("pp_spconv_fwd_ex: split_k %d needs K >= split_k and a scratch of %zu bytes (pp_spconv_set_scratch)", sk,
                     (size_t)sk * (size_t)n_out * (size_t)cout * sizeof(float));
        return PP_ERR_WORKSPACE;
      }
    }
    if (ds_in) {
      // the fused shortcut rides on the unsplit pipelined kernel over a same-level map; anything else is the caller's
      // two launches (PP_UNSUPPORTED: nothing was launched)
      if (a.split > 1 || row_order || n_in != n_out || ds_c % 16 != 0 || ds_c <= 0 ||
          (double)n_out * ds_c * 4.0 >= 4294967000.0 || rows_per_wave || pipeline)
        return PP_UNSUPPORTED;
      a.ds_in = ds_in; a.ds_wp = ds_packed; a.ds_scale = ds_scale; a.ds_shift = ds_shift; a.ds_c = ds_c;
    }
    // wide layers: the bf16 matrix pipe with exactly split fp32 operands (pp_spconv3.hip) unless a variant was asked for
    // (x3_column_tiles above: the rule, also exported as pp_spconv_kernel_family)
    int rc = PP_UNSUPPORTED;
    if (!pipeline && !rows_per_wave) {
      const int n4 = x3_column_tiles(a, n_in, ds_in != nullptr);
      if (n4) rc = pp_spconv_x3_launch(a, n_in, n4, (unsigned)((a.NT + n4 - 1) / n4), pp_s(stream));
    }
    if (rc == PP_UNSUPPORTED) rc = pp_spconv_fwd3_launch(a, n_in, ntw, (unsigned)groups, T, depth, pp_s(stream));
    if (rc != PP_OK) return rc;
    if (a.split > 1) return pp_spconv_split_reduce_launch(a, pp_s(stream));
  } else {
    if (ds_in) return PP_UNSUPPORTED;
    if (t8) {
      pp_set_error("pp_spconv_fwd_t8 / _cmap: needs cin %% 16 == 0 (or the 4-channel input layer) and inputs < 4 GiB per source (the pipelined kernel)");
      return PP_ERR_INVALID;
    }
    // first-version kernel: Cin % 16 != 0 other than the 4-channel input layer, and inputs of 4 GiB or more per source
    if (bf16) {
      pp_set_error("pp_spconv_fwd_bf16: needs cin %% 16 == 0, K <= 28, inputs < 4 GiB (got cin %d, K %d)", c0 + c1, K);
      return PP_ERR_INVALID;
    }
    if (row_order) {  // only the pipelined kernel understands slot-ordered maps
      pp_set_error("pp_spconv_fwd: a slot-ordered map needs cin %% 16 == 0 and inputs < 4 GiB per source");
      return PP_ERR_INVALID;
    }
    if (rows_per_wave || pipeline || split_k > 1) {
      pp_set_error("pp_spconv_fwd_ex: this shape runs on the first-version kernel, which has no variants");
      return PP_ERR_INVALID;
    }
    dim3 grid(pp_blocks(n_out, 128), (unsigned)groups);
    if (mode16)
      launch_fwd<true>(ntw, grid, pp_s(stream), a);
    else
      launch_fwd<false>(ntw, grid, pp_s(stream), a);
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_spconv_fwd(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in,
                             const float* packed_weight, const int32_t* nbr, int32_t K, int64_t n_out, int32_t cout,
                             const float* scale, const float* shift, int32_t relu, const float* residual,
                             const int32_t* row_order, float* out, pp_stream_t stream) {
  return spconv_fwd_impl(in0, c0, in1, c1, n_in, packed_weight, nbr, K, n_out, cout, scale, shift, relu, residual,
                         row_order, out, 0, 0, 0, 0, stream);
}
/* pp_spconv_fwd with the 1x1 shortcut of a residual block fused in:
 *   out[o] = relu?( conv(o) * scale + shift ) + residual[o] + ( ds_in[o] . W_ds ) * ds_scale + ds_shift
 * (ds_in [n_out, ds_c] fp32, ds_packed = pp_pack_weight of the [1, ds_c, cout] kernel).  Served on same-level maps by the
 * unsplit pipelined kernel; otherwise PP_UNSUPPORTED is returned and NOTHING has been launched (the caller runs the
 * shortcut as its own 1x1 convolution and passes it as `residual`). */
extern "C" int pp_spconv_fwd_shortcut(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in,
                                      const float* packed_weight, const int32_t* nbr, int32_t K, int64_t n_out, int32_t cout,
                                      const float* scale, const float* shift, int32_t relu, const float* residual,
                                      float* out, int32_t bf16, const float* ds_in, int32_t ds_c, const float* ds_packed,
                                      const float* ds_scale, const float* ds_shift, pp_stream_t stream) {
  PP_REQUIRE(ds_in && ds_packed, "pp_spconv_fwd_shortcut: null shortcut input / weights");
  return spconv_fwd_impl(in0, c0, in1, c1, n_in, packed_weight, nbr, K, n_out, cout, scale, shift, relu, residual, nullptr,
                         out, bf16 ? 1 : 0, 0, 0, 0, stream, ds_in, ds_c, ds_packed, ds_scale, ds_shift);
}
extern "C" int pp_spconv_fwd_bf16(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in,
                                  const float* packed_weight, const int32_t* nbr, int32_t K, int64_t n_out,
                                  int32_t cout, const float* scale, const float* shift, int32_t relu,
                                  const float* residual, const int32_t* row_order, float* out, pp_stream_t stream) {
  return spconv_fwd_impl(in0, c0, in1, c1, n_in, packed_weight, nbr, K, n_out, cout, scale, shift, relu, residual,
                         row_order, out, 1, 0, 0, 0, stream);
}
/* Transposed stride-2 convolution on the 8-wide map of pp_kernel_map_transpose8: nbr8 int32 [8][n_out] slot-major (after
 * pp_map_permute with K = 8), row_order [n_out] the slot order.  Same results, bit for bit, as pp_spconv_fwd on the dense
 * 27-wide map of pp_kernel_map_transpose in the same slot order. */
extern "C" int pp_spconv_fwd_t8(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in,
                                const float* packed_weight, const int32_t* nbr8, int64_t n_out, int32_t cout,
                                const float* scale, const float* shift, int32_t relu, const float* residual,
                                const int32_t* row_order, float* out, int32_t bf16, pp_stream_t stream) {
  return spconv_fwd_impl(in0, c0, in1, c1, n_in, packed_weight, nbr8, 27, n_out, cout, scale, shift, relu, residual,
                         row_order, out, bf16 ? 1 : 0, 0, 0, 0, stream, nullptr, 0, nullptr, nullptr, nullptr, 1);
}
/* pp_spconv_fwd / pp_spconv_fwd_shortcut on the COMPACT form of a same-level map (pp_map_compact_count / _write): the kernel's
 * prologue reads 4 + 6 x pairs bytes per row instead of the dense map's 108.  Same results, bit for bit.  ds_in NULL: no fused
 * shortcut.  PP_UNSUPPORTED as for pp_spconv_fwd_shortcut. */
extern "C" int pp_spconv_fwd_cmap(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in,
                                  const float* packed_weight, const uint32_t* cm_mask, const int32_t* cm_start,
                                  const uint32_t* cm_entries, const uint16_t* cm_tags, int64_t n_out, int32_t cout,
                                  const float* scale, const float* shift, int32_t relu, const float* residual, float* out,
                                  int32_t bf16, const float* ds_in, int32_t ds_c, const float* ds_packed, const float* ds_scale,
                                  const float* ds_shift, pp_stream_t stream) {
  PP_REQUIRE(!ds_in || ds_packed, "pp_spconv_fwd_cmap: null shortcut weights");
  return spconv_fwd_impl(in0, c0, in1, c1, n_in, packed_weight, (const int32_t*)cm_entries, 27, n_out, cout, scale, shift, relu,
                         residual, nullptr, out, bf16 ? 1 : 0, 0, 0, 0, stream, ds_in, ds_c, ds_packed, ds_scale, ds_shift, 2,
                         cm_mask, cm_start, cm_tags);
}
extern "C" int pp_spconv_fwd_ex(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in,
                                const float* packed_weight, const int32_t* nbr, int32_t K, int64_t n_out, int32_t cout,
                                const float* scale, const float* shift, int32_t relu, const float* residual,
                                const int32_t* row_order, float* out, int32_t bf16, int32_t rows_per_wave,
                                int32_t pipeline, int32_t split_k, pp_stream_t stream) {
  return spconv_fwd_impl(in0, c0, in1, c1, n_in, packed_weight, nbr, K, n_out, cout, scale, shift, relu, residual,
                         row_order, out, bf16 ? 1 : 0, rows_per_wave, pipeline, split_k, stream);
}

// ---------------------------------------------------------------------------------------------
// K5 weight gradient: dW[k] = sum_o in[nbr[k][o]]^T dout[o]
// MFMA with the reduction (rows) as the K dimension: A[i=ci][kk=row] , B[kk=row][j=co].
// grid = (row chunks, K offsets, ci tiles); each wave owns ROWS_PER_WAVE rows, one 16-wide ci tile and all
// co tiles (<= 12), then adds its partial tile into dW with float atomics (few per wave).
// ---------------------------------------------------------------------------------------------
#define BWW_ROWS_PER_WAVE 512
template <int NTO>
__global__ __launch_bounds__(256) void k_spconv_bwd_weight(const float* __restrict__ in, int cin,
                                                           const float* __restrict__ dout, int cout,
                                                           const int32_t* __restrict__ nbr, int K, int64_t n_out,
                                                           float* __restrict__ dw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int k = blockIdx.y;
  const int ci = blockIdx.z * 16 + i;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * BWW_ROWS_PER_WAVE;
  if (row0 >= n_out) return;
  const int64_t row_end = row0 + BWW_ROWS_PER_WAVE < n_out ? row0 + BWW_ROWS_PER_WAVE : n_out;
  f32x4 acc[NTO];
#pragma unroll
  for (int jt = 0; jt < NTO; ++jt) acc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int64_t r = row0; r < row_end; r += 4) {
    const int64_t row = r + q;
    float A = 0.f;
    int src = -1;
    if (row < row_end) src = nbr ? nbr[(int64_t)k * n_out + row] : (int)row;
    if (__ballot(src >= 0) == 0ull) continue;
    if (src >= 0 && ci < cin) A = in[(int64_t)src * cin + ci];
#pragma unroll
    for (int jt = 0; jt < NTO; ++jt) {
      const int co = jt * 16 + i;
      float B = 0.f;
      if (src >= 0 && co < cout) B = dout[row * cout + co];
      acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, acc[jt], 0, 0, 0);
    }
  }
  // D[row = ci_local = 4q + r][col = co_local = i]
#pragma unroll
  for (int jt = 0; jt < NTO; ++jt) {
    const int co = jt * 16 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cil = blockIdx.z * 16 + 4 * q + r;
      if (cil < cin && co < cout && acc[jt][r] != 0.f)
        atomicAdd(&dw[((int64_t)k * cin + cil) * cout + co], acc[jt][r]);
    }
  }
}

static int spconv_bwd_weight_impl(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout,
                                  const int32_t* nbr, int32_t K, int64_t n_out, float* dw, int bf16,
                                  pp_stream_t stream) {
  PP_REQUIRE(in && dout && dw, "pp_spconv_bwd_weight: null pointer");
  PP_REQUIRE(nbr || K == 1, "pp_spconv_bwd_weight: a kernel map is required unless K == 1");
  PP_REQUIRE(cout <= 192, "pp_spconv_bwd_weight: cout > 192 unsupported");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)K * cin * cout, s));
  if (n_out == 0) return PP_OK;
  if (pp_spconv_bww2_ok(cin, cout, n_in, nbr))
    return pp_spconv_bww2_launch(in, cin, n_in, dout, cout, nbr, K, n_out, dw, bf16, s);
  if (bf16) {
    pp_set_error("pp_spconv_bwd_weight_bf16: needs a kernel map, cout <= 192 and inputs < 4 GiB");
    return PP_ERR_INVALID;
  }
  dim3 grid(pp_blocks(n_out, 4 * BWW_ROWS_PER_WAVE), (unsigned)K, (unsigned)((cin + 15) / 16));
  int nto = pp_nt(cout);
#define BWW_CASE(N) \
  case N: hipLaunchKernelGGL((k_spconv_bwd_weight<N>), grid, dim3(256), 0, s, in, cin, dout, cout, nbr, K, n_out, dw); break;
  switch (nto) {
    BWW_CASE(1) BWW_CASE(2) BWW_CASE(3) BWW_CASE(4) BWW_CASE(5) BWW_CASE(6) BWW_CASE(7) BWW_CASE(8) BWW_CASE(9)
    BWW_CASE(10) BWW_CASE(11)
    default: hipLaunchKernelGGL((k_spconv_bwd_weight<12>), grid, dim3(256), 0, s, in, cin, dout, cout, nbr, K, n_out, dw);
  }
#undef BWW_CASE
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_spconv_bwd_weight(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout,
                                    const int32_t* nbr, int32_t K, int64_t n_out, float* dw, pp_stream_t stream) {
  return spconv_bwd_weight_impl(in, cin, n_in, dout, cout, nbr, K, n_out, dw, 0, stream);
}
extern "C" int pp_spconv_bwd_weight_bf16(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout,
                                         const int32_t* nbr, int32_t K, int64_t n_out, float* dw, pp_stream_t stream) {
  return spconv_bwd_weight_impl(in, cin, n_in, dout, cout, nbr, K, n_out, dw, 1, stream);
}
