// K5 weight gradient, pipelined variant:  dW[k] = sum_o in[nbr[k][o]]^T dout[o]
// replaces: the kernel-weight gradient of ME.MinkowskiConvolution / ConvolutionTranspose backward
// (torch_points3d/modules/MinkowskiEngine/api_modules.py:40,53; training path base_model / PointGroup3heads.py:120-173).
//
// MFMA 16x16x4 with the reduction (output rows) as the K dimension: A[i = ci][kk = row], B[kk = row][j = co].
// One wave owns ONE kernel offset k, MT 16-wide ci tiles and all NTO co tiles over a chunk of rows:
//  * the 4 waves of a block walk the SAME rows for 4 different k, so the dout rows they load are shared in L1;
//  * 16 rows per iteration: 4 x (MT + NTO) buffer loads are issued back to back before the first MFMA, and the
//    neighbour indices of the next iteration are prefetched, so there is no load -> load -> MFMA chain per 4 rows;
//  * missing neighbours, rows past the chunk and channels past cin/cout read hardware zeros through out-of-range
//    buffer offsets (no branches between the loads and the MFMAs); 4-row sub-steps without any neighbour skip
//    their MFMAs, whole iterations without any neighbour skip their loads as well.
// The partial tiles are added into dW with float atomics (4 * MT * NTO per lane per chunk).
#include "pp_spconv.h"

typedef unsigned int u32;

template <int MT, int NTO, bool BF16>
__global__ __launch_bounds__(256) void k_spconv_bww2(const float* __restrict__ in, int cin, u32 in_bytes,
                                                     const float* __restrict__ dout, int cout,
                                                     const int32_t* __restrict__ nbr, int K, int64_t n_out,
                                                     int rows_per_block, float* __restrict__ dw) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int k = blockIdx.y * 4 + wave;
  if (k >= K) return;
  const int ci0 = blockIdx.z * (16 * MT);
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t row_end = row0 + rows_per_block < n_out ? row0 + rows_per_block : n_out;
  const int nrows = (int)(row_end - row0);

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(dout + row0 * cout), 0,
                                                                      nrows * cout * 4, 0x00020000);
  const u32 cin4 = (u32)cin * 4u, cout4 = (u32)cout * 4u;
  u32 cio[MT], cofs[NTO];
  bool civ[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int ci = ci0 + mt * 16 + i;
    civ[mt] = ci < cin;
    cio[mt] = (u32)ci * 4u;
  }
#pragma unroll
  for (int jt = 0; jt < NTO; ++jt) {
    const int co = jt * 16 + i;
    cofs[jt] = co < cout ? (u32)co * 4u : 0x80000000u;  // chunk bytes < 2^31: stays out of range after the row offset
  }

  f32x4 acc[MT][NTO];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int jt = 0; jt < NTO; ++jt) acc[mt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // neighbour indices through a buffer descriptor over this wave's slice of the map: the prefetch of the next
  // iteration is unconditional (rows past the chunk read 0 and are turned into -1 when they become current)
  const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)(nbr + (int64_t)k * n_out + row0), 0,
                                                                      nrows * 4, 0x00020000);
  int s[4], ns[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int rl = 4 * q + m;
    s[m] = __builtin_amdgcn_raw_buffer_load_b32(rn, rl * 4, 0, 0);
    s[m] = rl < nrows ? s[m] : -1;
  }
  for (int r = 0; r < nrows; r += 16) {
#pragma unroll
    for (int m = 0; m < 4; ++m) ns[m] = __builtin_amdgcn_raw_buffer_load_b32(rn, (r + 16 + 4 * q + m) * 4, 0, 0);
    if (__ballot((s[0] & s[1] & s[2] & s[3]) >= 0) != 0ull) {
      float A[4][MT], B[4][NTO];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const u32 o = (s[m] < 0 || !civ[mt]) ? 0xFFFFFFFFu : (u32)s[m] * cin4 + cio[mt];
          A[m][mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (int)o, 0, 0));
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const u32 ro = (u32)(r + 4 * q + m) * cout4;
#pragma unroll
        for (int jt = 0; jt < NTO; ++jt)
          B[m][jt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (int)(ro + cofs[jt]), 0, 0));
      }
      asm volatile("" ::: "memory");  // keep every load above the sub-step branches (hipcc would sink them)
      if constexpr (BF16) {
        // the 4 sub-steps of a lane are k = 4q .. 4q+3 of one v_mfma_f32_16x16x16_bf16: one MFMA instead of four
        s16x4 ah[MT], bh[NTO];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ah[mt] = pp_bf16x4((f32x4){A[0][mt], A[1][mt], A[2][mt], A[3][mt]});
#pragma unroll
        for (int jt = 0; jt < NTO; ++jt) bh[jt] = pp_bf16x4((f32x4){B[0][jt], B[1][jt], B[2][jt], B[3][jt]});
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int jt = 0; jt < NTO; ++jt)
            acc[mt][jt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[mt], bh[jt], acc[mt][jt], 0, 0, 0);
      } else
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (__ballot(s[m] >= 0) != 0ull) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int jt = 0; jt < NTO; ++jt)
              acc[mt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m][mt], B[m][jt], acc[mt][jt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) s[m] = (r + 16 + 4 * q + m) < nrows ? ns[m] : -1;
  }
  // D[row = ci_local = 4q + e][col = co_local = i]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int jt = 0; jt < NTO; ++jt) {
      const int co = jt * 16 + i;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ci = ci0 + mt * 16 + 4 * q + e;
        if (ci < cin && co < cout && acc[mt][jt][e] != 0.f)
          atomicAdd(&dw[((int64_t)k * cin + ci) * cout + co], acc[mt][jt][e]);
      }
    }
}

bool pp_spconv_bww2_ok(int cin, int cout, int64_t n_in, const int32_t* nbr) {
  if (n_in <= 0 || cout > 192 || !nbr) return false;  // K == 1 without a map stays on the first kernel
  return (double)n_in * cin * 4.0 < 4294967040.0;  // 32-bit buffer offsets over the input rows
}

template <int MT, int NTO>
static void bww2_go(dim3 grid, hipStream_t s, const float* in, int cin, u32 in_bytes, const float* dout, int cout,
                    const int32_t* nbr, int K, int64_t n_out, int rpb, float* dw, int bf16) {
  if (bf16)
    hipLaunchKernelGGL((k_spconv_bww2<MT, NTO, true>), grid, dim3(256), 0, s, in, cin, in_bytes, dout, cout, nbr, K,
                       n_out, rpb, dw);
  else
    hipLaunchKernelGGL((k_spconv_bww2<MT, NTO, false>), grid, dim3(256), 0, s, in, cin, in_bytes, dout, cout, nbr, K,
                       n_out, rpb, dw);
}

int pp_spconv_bww2_launch(const float* in, int cin, int64_t n_in, const float* dout, int cout, const int32_t* nbr,
                          int K, int64_t n_out, float* dw, int bf16, hipStream_t s) {
  const int nto = (cout + 15) / 16, ntiles = (cin + 15) / 16;
  int mt = nto <= 2 ? 4 : (nto <= 6 ? 2 : 1);
  while (mt > 1 && ntiles % mt != 0) mt >>= 1;
  const unsigned gy = (unsigned)((K + 3) / 4), gz = (unsigned)(ntiles / mt);
  int rpb = 512;
  while (rpb > 128 && ((n_out + rpb - 1) / rpb) * gy * gz < 1024) rpb >>= 1;
  dim3 grid((unsigned)((n_out + rpb - 1) / rpb), gy, gz);
  const u32 in_bytes = (u32)((uint64_t)n_in * cin * 4u);
#define BWW2(M, N) \
  bww2_go<M, N>(grid, s, in, cin, in_bytes, dout, cout, nbr, K, n_out, rpb, dw, bf16); break;
  switch (mt * 16 + nto) {
    case 4 * 16 + 1: BWW2(4, 1)
    case 4 * 16 + 2: BWW2(4, 2)
    case 2 * 16 + 1: BWW2(2, 1)
    case 2 * 16 + 2: BWW2(2, 2)
    case 2 * 16 + 3: BWW2(2, 3)
    case 2 * 16 + 4: BWW2(2, 4)
    case 2 * 16 + 5: BWW2(2, 5)
    case 2 * 16 + 6: BWW2(2, 6)
    case 1 * 16 + 1: BWW2(1, 1)
    case 1 * 16 + 2: BWW2(1, 2)
    case 1 * 16 + 3: BWW2(1, 3)
    case 1 * 16 + 4: BWW2(1, 4)
    case 1 * 16 + 5: BWW2(1, 5)
    case 1 * 16 + 6: BWW2(1, 6)
    case 1 * 16 + 7: BWW2(1, 7)
    case 1 * 16 + 8: BWW2(1, 8)
    case 1 * 16 + 9: BWW2(1, 9)
    case 1 * 16 + 10: BWW2(1, 10)
    case 1 * 16 + 11: BWW2(1, 11)
    default: BWW2(1, 12)
  }
#undef BWW2
  PP_LAUNCH_CHECK();
  return PP_OK;
}
