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


// -------------------------------------------------------------------------------------------------------------------
// Pair-major weight gradient.  On the model's maps k_spconv_bww2 runs only 1.2 - 1.7 x faster than on a dense map with
// 27 neighbours per row although the maps hold 5.6 .. 16 (profiles/r03_ab_bww3.log): its time goes into the (offset,
// 16-row) groups it has to enter -- every group with at least one neighbour costs the full loads -- and with the
// neighbours of one offset scattered over the rows, about twice as many groups are entered as the pairs would fill.
// dW[k] is a sum over PAIRS, and unlike the forward pass nothing is accumulated per output row, so the pairs of every
// offset can be compacted first:
//   pp_wgrad_pairs_build : per offset k the list of (output row, input row) of its pairs, in row order (tile counts ->
//                          exclusive scan -> ordered write; a map is compacted once and serves every layer that uses it)
//   k_spconv_bww4        : a wave walks a chunk of one offset's list; every 16-pair step is full (but the tail), both
//                          operands are gathered (dout rows by the stored output row: a slot-ordered map needs no
//                          re-ordered copy of dout any more, the row order is folded into the list).
// -------------------------------------------------------------------------------------------------------------------
#define WP_TILE 1024   // rows of one offset per counted tile: 256 threads x 4 consecutive rows

__global__ __launch_bounds__(256) void k_wpairs_count(const int32_t* __restrict__ nbr, int64_t n_out, int tiles_per_k,
                                                      int32_t* __restrict__ tile_count) {
  __shared__ int wc[4];
  const int k = blockIdx.x / tiles_per_k, t = blockIdx.x - k * tiles_per_k;
  const int32_t* row = nbr + (int64_t)k * n_out;
  const int64_t r0 = (int64_t)t * WP_TILE + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) c += (r0 + e < n_out && row[r0 + e] >= 0) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ __launch_bounds__(256) void k_wpairs_write(const int32_t* __restrict__ nbr, const int32_t* __restrict__ order,
                                                      int64_t n_out, int tiles_per_k,
                                                      const int32_t* __restrict__ tile_start, int2* __restrict__ pairs) {
  __shared__ int wc[4];
  const int k = blockIdx.x / tiles_per_k, t = blockIdx.x - k * tiles_per_k;
  const int32_t* row = nbr + (int64_t)k * n_out;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)t * WP_TILE + threadIdx.x * 4;
  int src[4], before = 0, total = 0;   // pairs of this wave in earlier lanes / in the whole wave (row order = lane, then e)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    src[e] = r0 + e < n_out ? row[r0 + e] : -1;
    const unsigned long long b = __ballot(src[e] >= 0);
    before += __popcll(b & ((1ull << lane) - 1ull));
    total += __popcll(b);
  }
  if (lane == 0) wc[wave] = total;
  __syncthreads();
  int off = tile_start[blockIdx.x] + before;
  for (int w = 0; w < wave; ++w) off += wc[w];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (src[e] >= 0) pairs[off++] = make_int2(order ? order[r0 + e] : (int)(r0 + e), src[e]);
}

extern "C" size_t pp_wgrad_pairs_workspace(int32_t K, int64_t n_out) {
  const int64_t tiles = (int64_t)K * ((n_out + WP_TILE - 1) / WP_TILE);
  return pp_align((size_t)tiles * 4) + pp_scan_workspace(tiles) + 256;
}

extern "C" int pp_wgrad_pairs_build(const int32_t* nbr, int32_t K, int64_t n_out, const int32_t* row_order,
                                    int32_t* pairs, int32_t* tile_start, void* ws, size_t ws_bytes, pp_stream_t stream) {
  PP_REQUIRE(tile_start && (n_out == 0 || (nbr && pairs && ws)), "pp_wgrad_pairs_build: null pointer");
  PP_REQUIRE(K >= 1 && n_out >= 0 && (int64_t)K * n_out < (int64_t(1) << 31), "pp_wgrad_pairs_build: K * n_out must be < 2^31");
  PP_REQUIRE(ws_bytes >= pp_wgrad_pairs_workspace(K, n_out), "pp_wgrad_pairs_build: workspace too small");
  hipStream_t s = pp_s(stream);
  const int tiles_per_k = (int)((n_out + WP_TILE - 1) / WP_TILE);
  const int64_t tiles = (int64_t)K * tiles_per_k;
  if (tiles == 0) {
    PP_HIP(hipMemsetAsync(tile_start, 0, sizeof(int32_t), s));
    return PP_OK;
  }
  PPArena ar(ws, ws_bytes);
  int32_t* cnt = ar.take<int32_t>((size_t)tiles);
  hipLaunchKernelGGL(k_wpairs_count, dim3((unsigned)tiles), dim3(256), 0, s, nbr, n_out, tiles_per_k, cnt);
  PP_LAUNCH_CHECK();
  int rc = pp_exclusive_scan_i32(cnt, tile_start, tiles, tile_start + tiles, ar.cur(), ar.left(), s);
  if (rc != PP_OK) return rc;
  hipLaunchKernelGGL(k_wpairs_write, dim3((unsigned)tiles), dim3(256), 0, s, nbr, row_order, n_out, tiles_per_k, tile_start,
                     (int2*)pairs);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// DET: the block's tile sum is STORED as a partial (part[k][z][block x][tile][256]) instead of added to dw with float atomics;
// k_wgrad_reduce then adds the partials of an offset in block order -- the same bits run after run
template <int MT, int NTO, bool BF16, int WPB, bool DET = false>
__global__ __launch_bounds__(WPB * 64) void k_spconv_bww4(const float* __restrict__ in, int cin, u32 in_bytes,
                                                     const float* __restrict__ dout, int cout, u32 dout_bytes,
                                                     const int2* __restrict__ pairs, const int32_t* __restrict__ tile_start,
                                                     int tiles_per_k, int chunk, float* __restrict__ dw) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  // the WPB waves of a block: consecutive chunks of `chunk` pairs of ONE offset; their partial tiles are added in LDS
  // and leave the block as one set of atomics (device-scope float atomics are the expensive part of short chunks)
  __shared__ float red[WPB][MT * NTO][256];
  const int k = blockIdx.y;
  const int k0 = tile_start[k * tiles_per_k], cnt = tile_start[(k + 1) * tiles_per_k] - k0;
  if ((int64_t)blockIdx.x * WPB * chunk >= cnt) return;  // the whole block is past the end of the list
  const int64_t p0 = ((int64_t)blockIdx.x * WPB + wave) * chunk;
  const int ks = k0 + (int)(p0 < cnt ? p0 : cnt);
  const int np = p0 >= cnt ? 0 : (cnt - (int)p0 < chunk ? cnt - (int)p0 : chunk);
  const int ci0 = blockIdx.z * (16 * MT);

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)dout, 0, (int)dout_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(pairs + ks), 0, np * 8, 0x00020000);
  const u32 cin4 = (u32)cin * 4u, cout4 = (u32)cout * 4u;
  u32 cio[MT], cofs[NTO];
  bool civ[MT], cov[NTO];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int ci = ci0 + mt * 16 + i;
    civ[mt] = ci < cin;
    cio[mt] = (u32)ci * 4u;
  }
#pragma unroll
  for (int jt = 0; jt < NTO; ++jt) {
    const int co = jt * 16 + i;
    cov[jt] = co < cout;
    cofs[jt] = (u32)co * 4u;
  }
  f32x4 acc[MT][NTO];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int jt = 0; jt < NTO; ++jt) acc[mt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef int v2i __attribute__((ext_vector_type(2)));
  // software pipeline, two steps deep: the operands of step r+16 are in flight during the MFMAs of step r, the pairs of
  // step r+32 behind them.  Pairs beyond the chunk become (-1, -1) -> out-of-range offsets -> hardware zeros, no traffic.
#define BWW4_PAIRS(R, P)                                                                                              \
  _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                                     \
    P[m] = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(rp, ((R) + 4 * q + m) * 8, 0, 0));            \
  }
#define BWW4_MASK(R, P) \
  _Pragma("unroll") for (int m = 0; m < 4; ++m) if ((R) + 4 * q + m >= np) P[m] = (v2i){-1, -1};
#define BWW4_OPERANDS(P, AX, BX)                                                                                      \
  _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                                     \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                               \
      const u32 o = (P[m].y < 0 || !civ[mt]) ? 0xFFFFFFFFu : (u32)P[m].y * cin4 + cio[mt];                            \
      AX[m][mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (int)o, 0, 0));                  \
    }                                                                                                                 \
    _Pragma("unroll") for (int jt = 0; jt < NTO; ++jt) {                                                              \
      const u32 o = (P[m].x < 0 || !cov[jt]) ? 0xFFFFFFFFu : (u32)P[m].x * cout4 + cofs[jt];                          \
      BX[m][jt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (int)o, 0, 0));                  \
    }                                                                                                                 \
  }
  v2i pr[4];
  float A[4][MT], B[4][NTO], An[4][MT], Bn[4][NTO];
  BWW4_PAIRS(0, pr)
  BWW4_MASK(0, pr)
  BWW4_OPERANDS(pr, A, B)
  BWW4_PAIRS(16, pr)
  for (int r = 0; r < np; r += 16) {
    BWW4_MASK(r + 16, pr)
    BWW4_OPERANDS(pr, An, Bn)
    BWW4_PAIRS(r + 32, pr)
    if constexpr (BF16) {
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
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int jt = 0; jt < NTO; ++jt)
            acc[mt][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m][mt], B[m][jt], acc[mt][jt], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) A[m][mt] = An[m][mt];
#pragma unroll
      for (int jt = 0; jt < NTO; ++jt) B[m][jt] = Bn[m][jt];
    }
  }
#undef BWW4_PAIRS
#undef BWW4_MASK
#undef BWW4_OPERANDS
  // block reduction of the waves' tiles through LDS, then one atomic per entry.  D[row = ci_local = 4q + e][col = co_local = i]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int jt = 0; jt < NTO; ++jt)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][mt * NTO + jt][e * 64 + lane] = acc[mt][jt][e];
  __syncthreads();
  for (int x = threadIdx.x; x < MT * NTO * 256; x += WPB * 64) {
    const int tile = x >> 8, e = (x >> 6) & 3, ln = x & 63;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) v += red[w][tile][e * 64 + ln];
    if constexpr (DET) {  // dw = the partial buffer here: [k][z][x][MT * NTO tiles][256]
      const int64_t slot = ((int64_t)k * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x;
      dw[slot * (MT * NTO * 256) + x] = v;
    } else {
      const int mt = tile / NTO, jt = tile - mt * NTO;
      const int ci = ci0 + mt * 16 + 4 * (ln >> 4) + e, co = jt * 16 + (ln & 15);
      if (ci < cin && co < cout && v != 0.f) atomicAdd(&dw[((int64_t)k * cin + ci) * cout + co], v);
    }
  }
}

// dw[k][ci][co] = sum over the blocks x = 0 .. nb_k - 1 of offset k, in that order, of their partial tiles (blocks past the end
// of the offset's pair list wrote nothing and are not read)
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, const int32_t* __restrict__ tile_start,
                                                      int tiles_per_k, int per_block, int gx, int gz, int mt, int nto, int cin,
                                                      int cout, float* __restrict__ dw) {
  const int k = blockIdx.y, z = blockIdx.z, tile = blockIdx.x;
  const int cnt = tile_start[(k + 1) * tiles_per_k] - tile_start[k * tiles_per_k];
  const int nb = (cnt + per_block - 1) / per_block;
  const int e = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int m = tile / nto, jt = tile - m * nto;
  const int ci = z * (16 * mt) + m * 16 + 4 * (ln >> 4) + e, co = jt * 16 + (ln & 15);
  const int tiles = mt * nto;
  const float* p = part + (((int64_t)k * gz + z) * gx) * ((int64_t)tiles * 256) + (int64_t)tile * 256 + threadIdx.x;
  float v = 0.f;
  for (int x = 0; x < nb; ++x) v += p[(int64_t)x * tiles * 256];
  if (ci < cin && co < cout) dw[((int64_t)k * cin + ci) * cout + co] = v;
}

#ifndef BWW4_WPB
#define BWW4_WPB 4   // waves per block (their tiles meet in LDS: WPB * MT * NTO KiB)
#endif
template <int MT, int NTO>
static void bww4_go(dim3 grid, hipStream_t s, const float* in, int cin, u32 in_bytes, const float* dout, int cout,
                    u32 dout_bytes, const int2* pairs, const int32_t* tile_start, int tiles_per_k, int chunk, float* dw,
                    int bf16, bool det = false) {
  if (det) {
    if (bf16)
      hipLaunchKernelGGL((k_spconv_bww4<MT, NTO, true, BWW4_WPB, true>), grid, dim3(BWW4_WPB * 64), 0, s, in, cin, in_bytes, dout, cout,
                         dout_bytes, pairs, tile_start, tiles_per_k, chunk, dw);
    else
      hipLaunchKernelGGL((k_spconv_bww4<MT, NTO, false, BWW4_WPB, true>), grid, dim3(BWW4_WPB * 64), 0, s, in, cin, in_bytes, dout, cout,
                         dout_bytes, pairs, tile_start, tiles_per_k, chunk, dw);
    return;
  }
  if (bf16)
    hipLaunchKernelGGL((k_spconv_bww4<MT, NTO, true, BWW4_WPB>), grid, dim3(BWW4_WPB * 64), 0, s, in, cin, in_bytes, dout, cout,
                       dout_bytes, pairs, tile_start, tiles_per_k, chunk, dw);
  else
    hipLaunchKernelGGL((k_spconv_bww4<MT, NTO, false, BWW4_WPB>), grid, dim3(BWW4_WPB * 64), 0, s, in, cin, in_bytes, dout, cout,
                       dout_bytes, pairs, tile_start, tiles_per_k, chunk, dw);
}

// launch plan of the pair-major weight gradient: ci tiles per wave (mt), pairs per wave (chunk), grid
struct Bww4Plan {
  int mt, nto, chunk, tiles_per_k;
  unsigned gx, gz;
};
static size_t bww4_det_bytes(const Bww4Plan& p, int K) {
  return (size_t)p.gx * (size_t)K * p.gz * (size_t)(p.mt * p.nto) * 256u * sizeof(float) + 256;
}
// det: the plan of the deterministic form -- one partial tile set per block, so the workspace grows with the number of blocks; from
// PP_WGRAD_DET_MAX_MB (default 1024) on, a wave walks more pairs (fewer, longer blocks: 512 pairs per wave cost a few per cent)
// so that the partials always fit: the ordered reduction serves every size, no fall-back to float atomics (round 6)
static Bww4Plan bww4_plan(int cin, int cout, int K, int64_t map_rows, bool det = false) {
  Bww4Plan p;
  p.tiles_per_k = (int)((map_rows + WP_TILE - 1) / WP_TILE);
  p.nto = (cout + 15) / 16;
  const int ntiles = (cin + 15) / 16;
  p.mt = p.nto <= 2 ? 4 : (p.nto <= 6 ? 2 : 1);
  while (p.mt > 1 && ntiles % p.mt != 0) p.mt >>= 1;
  p.gz = (unsigned)(ntiles / p.mt);
  // an offset holds at most map_rows pairs; blocks past the end of their offset's list leave at once
  static const int chunk_env = [] { const char* e = getenv("PP_WGRAD_CHUNK"); return e ? atoi(e) : 0; }();
  p.chunk = 256;  // pairs per wave: 128 / 256 / 384 measured within 5 % of each other, 512 and 64 slower
  if (((map_rows + 4 * p.chunk - 1) / (4 * p.chunk)) * (int64_t)K * p.gz < 2048) p.chunk = 128;
  if (chunk_env > 0) p.chunk = chunk_env;
  p.gx = (unsigned)((map_rows + BWW4_WPB * p.chunk - 1) / (BWW4_WPB * p.chunk));
  if (det) {
    static const size_t cap = [] { const char* e = getenv("PP_WGRAD_DET_MAX_MB"); return (size_t)(e && atoi(e) > 0 ? atoi(e) : 1024) << 20; }();
    while (bww4_det_bytes(p, K) > cap && p.gx > 1) {
      const size_t over = (bww4_det_bytes(p, K) + cap - 1) / cap;  // grow by the factor that is missing, in steps of 128 pairs
      p.chunk = (int)(((int64_t)p.chunk * (int64_t)over + 127) / 128 * 128);
      p.gx = (unsigned)((map_rows + BWW4_WPB * p.chunk - 1) / (BWW4_WPB * p.chunk));
    }
  }
  return p;
}

static int bww4_launch(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout, int64_t n_out,
                       const int32_t* pairs, const int32_t* tile_start, int32_t K, const Bww4Plan& pl, float* target,
                       int32_t bf16, bool det, hipStream_t s) {
  const int mt = pl.mt, nto = pl.nto, tiles_per_k = pl.tiles_per_k, chunk = pl.chunk;
  dim3 grid(pl.gx, (unsigned)K, pl.gz);
  const u32 in_bytes = (u32)((uint64_t)n_in * cin * 4u), dout_bytes = (u32)((uint64_t)n_out * cout * 4u);
  float* dw = target;
#define BWW4(M, N) \
  bww4_go<M, N>(grid, s, in, cin, in_bytes, dout, cout, dout_bytes, (const int2*)pairs, tile_start, tiles_per_k, chunk, dw, bf16, det); break;
  switch (mt * 16 + nto) {
    case 4 * 16 + 1: BWW4(4, 1)
    case 4 * 16 + 2: BWW4(4, 2)
    case 2 * 16 + 1: BWW4(2, 1)
    case 2 * 16 + 2: BWW4(2, 2)
    case 2 * 16 + 3: BWW4(2, 3)
    case 2 * 16 + 4: BWW4(2, 4)
    case 2 * 16 + 5: BWW4(2, 5)
    case 2 * 16 + 6: BWW4(2, 6)
    case 1 * 16 + 1: BWW4(1, 1)
    case 1 * 16 + 2: BWW4(1, 2)
    case 1 * 16 + 3: BWW4(1, 3)
    case 1 * 16 + 4: BWW4(1, 4)
    case 1 * 16 + 5: BWW4(1, 5)
    case 1 * 16 + 6: BWW4(1, 6)
    case 1 * 16 + 7: BWW4(1, 7)
    case 1 * 16 + 8: BWW4(1, 8)
    case 1 * 16 + 9: BWW4(1, 9)
    case 1 * 16 + 10: BWW4(1, 10)
    case 1 * 16 + 11: BWW4(1, 11)
    default: BWW4(1, 12)
  }
#undef BWW4
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_spconv_bwd_weight_pairs(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout,
                                          int64_t n_out, const int32_t* pairs, const int32_t* tile_start, int32_t K,
                                          int64_t map_rows, float* dw, int32_t bf16, pp_stream_t stream) {
  PP_REQUIRE(dw && tile_start, "pp_spconv_bwd_weight_pairs: null pointer");
  PP_REQUIRE(cin >= 1 && cout >= 1 && cout <= 192, "pp_spconv_bwd_weight_pairs: cout must be in [1,192]");
  PP_REQUIRE((double)n_in * cin * 4.0 < 4294967040.0 && (double)n_out * cout * 4.0 < 4294967040.0,
             "pp_spconv_bwd_weight_pairs: in and dout must be < 4 GiB each (32-bit buffer offsets)");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)K * cin * cout, s));
  if (map_rows == 0 || n_in == 0 || n_out == 0) return PP_OK;
  PP_REQUIRE(in && dout && pairs, "pp_spconv_bwd_weight_pairs: null pointer");
  return bww4_launch(in, cin, n_in, dout, cout, n_out, pairs, tile_start, K, bww4_plan(cin, cout, K, map_rows), dw, bf16, false, s);
}

// ---- deterministic form: block partials in a caller-provided workspace + an ordered reduction (no float atomics)
extern "C" size_t pp_spconv_bwd_weight_pairs_det_workspace(int32_t cin, int32_t cout, int32_t K, int64_t map_rows) {
  if (cin < 1 || cout < 1 || K < 1 || map_rows <= 0) return 256;
  return bww4_det_bytes(bww4_plan(cin, cout, K, map_rows, true), K);
}

extern "C" int pp_spconv_bwd_weight_pairs_det(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout,
                                              int64_t n_out, const int32_t* pairs, const int32_t* tile_start, int32_t K,
                                              int64_t map_rows, float* dw, int32_t bf16, void* ws, size_t ws_bytes,
                                              pp_stream_t stream) {
  PP_REQUIRE(dw && tile_start, "pp_spconv_bwd_weight_pairs_det: null pointer");
  PP_REQUIRE(cin >= 1 && cout >= 1 && cout <= 192, "pp_spconv_bwd_weight_pairs_det: cout must be in [1,192]");
  PP_REQUIRE((double)n_in * cin * 4.0 < 4294967040.0 && (double)n_out * cout * 4.0 < 4294967040.0,
             "pp_spconv_bwd_weight_pairs_det: in and dout must be < 4 GiB each (32-bit buffer offsets)");
  hipStream_t s = pp_s(stream);
  if (map_rows == 0 || n_in == 0 || n_out == 0) {
    PP_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)K * cin * cout, s));
    return PP_OK;
  }
  PP_REQUIRE(in && dout && pairs && ws, "pp_spconv_bwd_weight_pairs_det: null pointer");
  PP_REQUIRE(ws_bytes >= pp_spconv_bwd_weight_pairs_det_workspace(cin, cout, K, map_rows),
             "pp_spconv_bwd_weight_pairs_det: workspace too small");
  const Bww4Plan pl = bww4_plan(cin, cout, K, map_rows, true);
  int rc = bww4_launch(in, cin, n_in, dout, cout, n_out, pairs, tile_start, K, pl, (float*)ws, bf16, true, s);
  if (rc != PP_OK) return rc;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)(pl.mt * pl.nto), (unsigned)K, pl.gz), dim3(256), 0, s, (const float*)ws,
                     tile_start, pl.tiles_per_k, BWW4_WPB * pl.chunk, (int)pl.gx, (int)pl.gz, pl.mt, pl.nto, cin, cout, dw);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
