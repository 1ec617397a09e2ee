// K3b/K4b: block-compacted rulebook + sparse convolution on it.
//
// Why: on surface-like voxel data only 6-13 of the 27 offsets of a 3x3x3 kernel are occupied per output voxel
// (SURVEY.md App. F), so a dense offset loop spends 50-77 % of its MFMA issue slots on zero rows.  The rulebook
// regroups the kernel map per block of RB_ROWS consecutive (Morton-ordered) output rows and per offset k into
// COMPACT lists of (input row, local output row) pairs, padded to 16: the convolution then runs one 16-row MFMA
// tile per 16 ACTIVE pairs (~85 % useful work) instead of one per 16 output rows.
//
// Convolution kernel (one wave = one 128-row block x NTW*16 output channels, waves fully independent: no barriers,
// no atomics, fixed summation order => deterministic):
//   accumulators   LDS [128][NTW*16 (+4 pad)] fp32, private to the wave
//   A (gathered)   global_load_lds_dwordx4: lane L fetches quarter L&3 of input row L>>2, so every 4-lane quad reads
//                  one full 64 B row segment (coalesced; 16 line requests per instruction instead of 64) straight into
//                  a 2 x 1 KiB wave-private staging ring; the DMA of step s+1 overlaps the MFMAs of step s
//   A fragments    ds_read_b128 from the staging tile in MFMA layout (lane (i,q) -> row i, floats 4q..4q+3)
//   B fragments    packed weights (pp_pack_weight mode16), one coalesced 1 KiB load per (k, s, jt), L1/L2 resident
//   MFMA           v_mfma_f32_16x16x4_f32, NTW independent accumulator chains
//   scatter        D rows go to the LDS accumulator rows named by the rulebook (plain read-add-write: within a tile
//                  every output row appears once, and the LDS block belongs to this wave alone)
//   epilogue       folded BN / ReLU / residual, float4 coalesced stores (same contract as pp_spconv_fwd)
#include <stdlib.h>

#include "pp_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RB_ROWS 64
#define RB_K 27

// ---------------------------------------------------------------------------------------------
// rulebook build: one workgroup (4 waves) per block; wave w handles offsets k = w, w+4, ...
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rb_count(const int32_t* __restrict__ nbr, int64_t n_out, int64_t nblk,
                                                  int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  const int64_t row0 = b * RB_ROWS;
  for (int k = wave; k < RB_K; k += 4) {
    int c = 0;
#pragma unroll
    for (int h = 0; h < RB_ROWS / 64; ++h) {
      int64_t r = row0 + h * 64 + lane;
      bool act = r < n_out && nbr[(int64_t)k * n_out + r] >= 0;
      c += __popcll(__ballot(act));
    }
    if (lane == 0) cnt[b * (RB_K + 1) + k] = (c + 15) & ~15;
  }
  if (threadIdx.x == 0) cnt[b * (RB_K + 1) + RB_K] = 0;
}

__global__ __launch_bounds__(256) void k_rb_fill(const int32_t* __restrict__ nbr, int64_t n_out, int64_t nblk,
                                                 const int32_t* __restrict__ off, int32_t* __restrict__ rb_in,
                                                 int32_t* __restrict__ rb_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  const int64_t row0 = b * RB_ROWS;
  for (int k = wave; k < RB_K; k += 4) {
    const int beg = off[b * (RB_K + 1) + k], end = off[b * (RB_K + 1) + k + 1];
    int w = beg;
#pragma unroll
    for (int h = 0; h < RB_ROWS / 64; ++h) {
      int64_t r = row0 + h * 64 + lane;
      int v = r < n_out ? nbr[(int64_t)k * n_out + r] : -1;
      bool act = v >= 0;
      unsigned long long m = __ballot(act);
      if (act) {
        int p = w + __popcll(m & ((1ull << lane) - 1ull));
        rb_in[p] = v;
        rb_out[p] = h * 64 + lane;
      }
      w += __popcll(m);
    }
    for (int p = w + lane; p < end; p += 64) {  // padding: a valid input row, output row -1 (result discarded)
      rb_in[p] = 0;
      rb_out[p] = -1;
    }
  }
}

extern "C" int64_t pp_rulebook_blocks(int64_t n_out) { return (n_out + RB_ROWS - 1) / RB_ROWS; }
extern "C" size_t pp_rulebook_workspace(int64_t n_out) {
  return pp_align((size_t)(pp_rulebook_blocks(n_out) * (RB_K + 1) + 1) * 4) + pp_scan_workspace(pp_rulebook_blocks(n_out) * (RB_K + 1) + 1) + 1024;
}

// step 1: offsets.  rb_off int32 [nblk*28 + 1]; total[0] = number of entries (device)
extern "C" int pp_rulebook_offsets(const int32_t* nbr, int64_t n_out, int32_t* rb_off, int32_t* total, void* workspace,
                                   size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(nbr && rb_off && total, "pp_rulebook_offsets: null pointer");
  if (workspace_bytes < pp_rulebook_workspace(n_out)) return PP_ERR_WORKSPACE;
  hipStream_t s = pp_s(stream);
  const int64_t nblk = pp_rulebook_blocks(n_out);
  if (nblk == 0) {
    PP_HIP(hipMemsetAsync(total, 0, sizeof(int32_t), s));
    PP_HIP(hipMemsetAsync(rb_off, 0, sizeof(int32_t), s));
    return PP_OK;
  }
  PPArena ar(workspace, workspace_bytes);
  const int64_t m = nblk * (RB_K + 1);
  int32_t* cnt = ar.take<int32_t>((size_t)m + 1);
  PP_HIP(hipMemsetAsync(cnt + m, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(k_rb_count, dim3((unsigned)nblk), dim3(256), 0, s, nbr, n_out, nblk, cnt);
  PP_LAUNCH_CHECK();
  int rc = pp_exclusive_scan_i32(cnt, rb_off, m + 1, nullptr, ar.cur(), ar.left(), s);
  if (rc) return rc;
  PP_HIP(hipMemcpyAsync(total, rb_off + m, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  return PP_OK;
}
// step 2: entries.  rb_in / rb_out int32 [total]
extern "C" int pp_rulebook_fill(const int32_t* nbr, int64_t n_out, const int32_t* rb_off, int32_t* rb_in,
                                int32_t* rb_out, pp_stream_t stream) {
  PP_REQUIRE(nbr && rb_off && rb_in && rb_out, "pp_rulebook_fill: null pointer");
  const int64_t nblk = pp_rulebook_blocks(n_out);
  if (nblk == 0) return PP_OK;
  hipLaunchKernelGGL(k_rb_fill, dim3((unsigned)nblk), dim3(256), 0, pp_s(stream), nbr, n_out, nblk, rb_off, rb_in,
                     rb_out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// convolution on the rulebook
// ---------------------------------------------------------------------------------------------
struct RbArgs {
  const float* in0;
  const float* in1;
  const float* wp;
  const int32_t* rb_off;
  const int32_t* rb_in;
  const int32_t* rb_out;
  const float* scale;
  const float* shift;
  const float* residual;
  float* out;
  int64_t n_out, nblk;
  int c0, c1, cout, NT, relu;
};

__device__ inline unsigned rb_xcd_remap(unsigned b, unsigned n) {
  const unsigned q = n >> 3, r = n & 7u, x = b & 7u, j = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// Work unit = TB tiles x CH channel steps of ONE offset k: all TB*CH A-tile DMAs and the CH*NTW B fragments are
// issued back to back, ONE s_waitcnt covers them, then TB*CH*NTW*4 MFMAs run.  (TB,CH) = (4,1) for Cin = 16,
// (2,2) for Cin = 32, (1,4) otherwise (chunks of 4 channel steps).  The entries of the next unit are prefetched
// before the wait, so the dependent chain entry -> DMA address is off the critical path.
template <int NTW, int TB, int CH>
__global__ __launch_bounds__(256) void k_spconv_rb(RbArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LD = NTW * 16 + 4;                                 // padded accumulator row stride (floats)
  constexpr int ACC_FLOATS = (RB_ROWS + 1) * LD;                   // +1: dummy row that swallows padding pairs
  constexpr int STAGE_FLOATS = TB * CH * 256;                      // TB*CH x 1 KiB A tiles
  constexpr int WAVE_FLOATS = ACC_FLOATS + STAGE_FLOATS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int64_t bid = (int64_t)rb_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (bid >= a.nblk) return;  // wave-uniform; waves never synchronise with each other
  float* acc_lds = (float*)smem + (size_t)wave * WAVE_FLOATS;
  float* stage = acc_lds + ACC_FLOATS;
  const int jt0 = blockIdx.y * NTW;
  const int cin = a.c0 + a.c1;
  const int S0 = a.c0 >> 4, S = cin >> 4;

  for (int t = lane; t < ACC_FLOATS / 4; t += 64) ((f32x4*)acc_lds)[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int32_t* off = a.rb_off + bid * (RB_K + 1);
  const int dma_row = lane >> 2, dma_q = lane & 3;
  const int blk_beg = __builtin_amdgcn_readfirstlane(off[0]);
  const int blk_end = __builtin_amdgcn_readfirstlane(off[RB_K]);
  // entries of the first unit
  int e_in[TB], e_out[TB];
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) {
    const int p = blk_beg + tb * 16 + i;
    e_in[tb] = p < blk_end ? a.rb_in[p] : 0;
    e_out[tb] = p < blk_end ? a.rb_out[p] : -1;
  }
  int k = 0;
  int kend = __builtin_amdgcn_readfirstlane(off[1]);
  for (int t = blk_beg; t < blk_end;) {
    while (t >= kend) {  // advance to the offset owning tile t (wave-uniform)
      ++k;
      kend = __builtin_amdgcn_readfirstlane(off[k + 1]);
    }
    // tiles of this unit: up to TB, all inside offset k
    int ntile = (kend - t) >> 4;
    if (ntile > TB) ntile = TB;
    const float* wk = a.wp + (int64_t)k * S * a.NT * 256;
    int row_dma[TB], lo_cur[TB];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      row_dma[tb] = __shfl(e_in[tb], dma_row);
      lo_cur[tb] = e_out[tb];
    }
    // prefetch the next unit's entries (consumed after this unit's MFMAs)
    const int tn = t + ntile * 16;
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      const int p = tn + tb * 16 + i;
      e_in[tb] = p < blk_end ? a.rb_in[p] : 0;
      e_out[tb] = p < blk_end ? a.rb_out[p] : -1;
    }
    f32x4 acc[TB][NTW];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int jt = 0; jt < NTW; ++jt) acc[tb][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int s0 = 0; s0 < S; s0 += CH) {
      const int ch = (S - s0) < CH ? (S - s0) : CH;
      // issue: A tiles by LDS-DMA, B fragments to registers
#pragma unroll
      for (int sl = 0; sl < CH; ++sl) {
        if (sl < ch) {
          const int sg = s0 + sl;
#pragma unroll
          for (int tb = 0; tb < TB; ++tb) {
            if (tb < ntile) {
              const float* g = sg < S0 ? a.in0 + (int64_t)row_dma[tb] * a.c0 + sg * 16 + dma_q * 4
                                       : a.in1 + (int64_t)row_dma[tb] * a.c1 + (sg - S0) * 16 + dma_q * 4;
              __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(stage + (tb * CH + sl) * 256), 16, 0, 0);
            }
          }
        }
      }
      f32x4 B[CH][NTW];
#pragma unroll
      for (int sl = 0; sl < CH; ++sl) {
#pragma unroll
        for (int jt = 0; jt < NTW; ++jt) {
          B[sl][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (sl < ch && jt0 + jt < a.NT)
            B[sl][jt] = *(const f32x4*)(wk + ((int64_t)(s0 + sl) * a.NT + jt0 + jt) * 256 + lane * 4);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMAs (and B, and the prefetched entries) have landed
#pragma unroll
      for (int sl = 0; sl < CH; ++sl) {
        if (sl < ch) {
#pragma unroll
          for (int tb = 0; tb < TB; ++tb) {
            if (tb < ntile) {
              const f32x4 A = *(const f32x4*)(stage + (tb * CH + sl) * 256 + i * 16 + q * 4);
#pragma unroll
              for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int jt = 0; jt < NTW; ++jt)
                  acc[tb][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt], B[sl][jt][tt], acc[tb][jt], 0, 0, 0);
            }
          }
        }
      }
      asm volatile("" ::: "memory");  // the staging tiles are re-filled by the next chunk's DMAs
    }
    // scatter the TB x (16 x NTW*16) tiles into the wave's LDS accumulator rows (padding pairs -> dummy row)
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      if (tb < ntile) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int lo = __shfl(lo_cur[tb], 4 * q + r);
          lo = lo < 0 ? RB_ROWS : lo;
#pragma unroll
          for (int jt = 0; jt < NTW; ++jt) acc_lds[lo * LD + jt * 16 + i] += acc[tb][jt][r];
        }
      }
    }
    t = tn;
  }

  // epilogue: LDS rows -> global, float4 per lane
  const int64_t row_base = bid * RB_ROWS;
  constexpr int CQ = NTW * 4;  // float4 chunks per row
  for (int c = lane; c < RB_ROWS * CQ; c += 64) {
    const int row = c / CQ, cq = c % CQ;
    const int64_t grow = row_base + row;
    const int col = jt0 * 16 + cq * 4;
    if (grow < a.n_out && col < a.cout) {
      f32x4 v = *(const f32x4*)(acc_lds + row * LD + cq * 4);
      float e[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (col + u < a.cout) {
          float x = e[u];
          if (a.scale) x *= a.scale[col + u];
          if (a.shift) x += a.shift[col + u];
          if (a.relu) x = fmaxf(x, 0.f);
          e[u] = x;
        }
      }
      if (col + 3 < a.cout) {
        if (a.residual) {
          f32x4 rr = *(const f32x4*)(a.residual + grow * a.cout + col);
          e[0] += rr[0]; e[1] += rr[1]; e[2] += rr[2]; e[3] += rr[3];
        }
        *(f32x4*)(a.out + grow * a.cout + col) = (f32x4){e[0], e[1], e[2], e[3]};
      } else {
        for (int u = 0; u < 4 && col + u < a.cout; ++u)
          a.out[grow * a.cout + col + u] = e[u] + (a.residual ? a.residual[grow * a.cout + col + u] : 0.f);
      }
    }
  }
}

template <int NTW, int TB, int CH>
static int rb_launch(const RbArgs& a, dim3 grid, hipStream_t s) {
  constexpr size_t lds = 4 * ((RB_ROWS + 1) * (NTW * 16 + 4) + TB * CH * 256) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    PP_HIP(hipFuncSetAttribute((const void*)k_spconv_rb<NTW, TB, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL((k_spconv_rb<NTW, TB, CH>), grid, dim3(256), lds, s, a);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// v3: counted-vmcnt DMA ring.  One wave = one 64-row block x ONE 16-channel output tile.  A "unit" is CH channel
// steps of one 16-pair tile: its CH A-tiles AND its CH B-fragments arrive by global_load_lds (2*CH DMAs, always --
// short chunks re-issue a valid address so the count is constant).  Units are issued NBUF-1 ahead into an LDS ring and
// retired with s_waitcnt vmcnt((NBUF-1)*2*CH): no other vector-memory instruction exists in the loop (the rulebook
// entries come through scalar loads), so hipcc never inserts a draining vmcnt(0) (guide: "mixing load kinds").
// ---------------------------------------------------------------------------------------------
template <int CH, int NBUF>
__global__ __launch_bounds__(256) void k_spconv_rb3(const float* __restrict__ in0, const float* __restrict__ in1,
                                                    const float* __restrict__ wp, const int32_t* __restrict__ rb_off,
                                                    const int32_t* __restrict__ rb_in, const int32_t* __restrict__ rb_out,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ residual, float* __restrict__ out,
                                                    int64_t n_out, int64_t nblk, int c0, int c1, int cout, int NT, int relu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LD = 20;                              // 16 columns + 4 pad
  constexpr int ACC_FLOATS = (RB_ROWS + 1) * LD;      // + dummy row for padding pairs
  constexpr int SLOT_FLOATS = 2 * CH * 256;           // CH A tiles then CH B fragments, 1 KiB each
  constexpr int WAVE_FLOATS = ACC_FLOATS + NBUF * SLOT_FLOATS;
  constexpr int D = NBUF - 1;                         // units in flight ahead of the one being computed
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int64_t bid = (int64_t)rb_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (bid >= nblk) return;
  float* acc_lds = (float*)smem + (size_t)wave * WAVE_FLOATS;
  float* ring = acc_lds + ACC_FLOATS;
  const int jt = blockIdx.y;
  const int cin = c0 + c1;
  const int S0 = c0 >> 4, S = cin >> 4;
  const int NCH = (S + CH - 1) / CH;                  // chunks (units) per tile

  for (int t = lane; t < ACC_FLOATS / 4; t += 64) ((f32x4*)acc_lds)[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int32_t* off = rb_off + bid * (RB_K + 1);
  const int blk_beg = __builtin_amdgcn_readfirstlane(off[0]);
  const int blk_end = __builtin_amdgcn_readfirstlane(off[RB_K]);
  const int ntiles = (blk_end - blk_beg) >> 4;
  const int U = ntiles * NCH;
  const int dma_row = lane >> 2, dma_q = lane & 3;

  // issue cursor (tile, chunk, offset k) and compute cursor
  int it = 0, ic = 0, ik = 0, ikend = __builtin_amdgcn_readfirstlane(off[1]);
  int ct = 0, cc = 0, ck = 0, ckend = ikend;
  int row_dma = 0;

  auto issue = [&](int slot) {
    if (ic == 0) {  // new tile: its 16 input rows (scalar loads), pick this lane's row
      const int pos = blk_beg + it * 16;
      while (pos >= ikend) {
        ++ik;
        ikend = __builtin_amdgcn_readfirstlane(off[ik + 1]);
      }
      const int32_t* pin = rb_in + pos;
      int r = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int v = __builtin_amdgcn_readfirstlane(pin[j]);
        r = (dma_row == j) ? v : r;
      }
      row_dma = r;
    }
    float* sl = ring + slot * SLOT_FLOATS;
    const float* wk = wp + ((int64_t)ik * S * NT + jt) * 256 + lane * 4;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      int sg = ic * CH + c;
      sg = sg < S ? sg : S - 1;  // short last chunk: re-issue a valid step (keeps the DMA count per unit constant)
      const float* g = sg < S0 ? in0 + (int64_t)row_dma * c0 + sg * 16 + dma_q * 4
                               : in1 + (int64_t)row_dma * c1 + (sg - S0) * 16 + dma_q * 4;
      __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(sl + c * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)(wk + (int64_t)sg * NT * 256), (lds_void*)(sl + (CH + c) * 256), 16, 0, 0);
    }
    if (++ic == NCH) {
      ic = 0;
      ++it;
    }
  };

  int issued = 0;
  for (; issued < D && issued < U; ++issued) issue(issued % NBUF);

  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int u = 0; u < U; ++u) {
    if (issued < U) {
      issue(issued % NBUF);
      ++issued;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * 2 * CH) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail: fewer than D units behind us
    }
    const float* sl = ring + (u % NBUF) * SLOT_FLOATS;
    const int nch = (S - cc * CH) < CH ? (S - cc * CH) : CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (c < nch) {
        const f32x4 A = *(const f32x4*)(sl + c * 256 + i * 16 + q * 4);
        const f32x4 B = *(const f32x4*)(sl + (CH + c) * 256 + lane * 4);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tt], B[tt], acc, 0, 0, 0);
      }
    }
    asm volatile("" ::: "memory");
    if (++cc == NCH) {  // tile complete: scatter its 16 x 16 result into the LDS accumulator rows
      const int32_t* pout = rb_out + blk_beg + ct * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int lo = 0;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int v = __builtin_amdgcn_readfirstlane(pout[4 * qq + r]);
          lo = (q == qq) ? v : lo;
        }
        lo = lo < 0 ? RB_ROWS : lo;
        acc_lds[lo * LD + i] += acc[r];
      }
      acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      cc = 0;
      ++ct;
    }
  }
  (void)ck; (void)ckend;

  // epilogue: 64 rows x 16 columns, float4 per lane
  const int64_t row_base = bid * RB_ROWS;
  for (int c = lane; c < RB_ROWS * 4; c += 64) {
    const int row = c >> 2, cq = c & 3;
    const int64_t grow = row_base + row;
    const int col = jt * 16 + cq * 4;
    if (grow < n_out && col < cout) {
      f32x4 v = *(const f32x4*)(acc_lds + row * LD + cq * 4);
      float e[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        if (col + uu < cout) {
          float x = e[uu];
          if (scale) x *= scale[col + uu];
          if (shift) x += shift[col + uu];
          if (relu) x = fmaxf(x, 0.f);
          e[uu] = x;
        }
      }
      if (col + 3 < cout) {
        if (residual) {
          f32x4 rr = *(const f32x4*)(residual + grow * cout + col);
          e[0] += rr[0]; e[1] += rr[1]; e[2] += rr[2]; e[3] += rr[3];
        }
        *(f32x4*)(out + grow * cout + col) = (f32x4){e[0], e[1], e[2], e[3]};
      } else {
        for (int uu = 0; uu < 4 && col + uu < cout; ++uu)
          out[grow * cout + col + uu] = e[uu] + (residual ? residual[grow * cout + col + uu] : 0.f);
      }
    }
  }
}

template <int CH, int NBUF>
static int rb3_launch(const RbArgs& a, hipStream_t s) {
  constexpr size_t lds = 4 * ((RB_ROWS + 1) * 20 + NBUF * 2 * CH * 256) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    PP_HIP(hipFuncSetAttribute((const void*)k_spconv_rb3<CH, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  dim3 grid((unsigned)((a.nblk + 3) / 4), (unsigned)a.NT);
  hipLaunchKernelGGL((k_spconv_rb3<CH, NBUF>), grid, dim3(256), lds, s, a.in0, a.in1, a.wp, a.rb_off, a.rb_in, a.rb_out,
                     a.scale, a.shift, a.residual, a.out, a.n_out, a.nblk, a.c0, a.c1, a.cout, a.NT, a.relu);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_spconv_fwd_rb(const float* in0, int32_t c0, const float* in1, int32_t c1, const float* packed_weight,
                                const int32_t* rb_off, const int32_t* rb_in, const int32_t* rb_out, int64_t n_out,
                                int32_t cout, const float* scale, const float* shift, int32_t relu,
                                const float* residual, float* out, pp_stream_t stream) {
  PP_REQUIRE(in0 && packed_weight && rb_off && rb_in && rb_out && out, "pp_spconv_fwd_rb: null pointer");
  PP_REQUIRE(c0 >= 0 && c1 >= 0 && (c0 + c1) > 0 && (c1 == 0 || in1), "pp_spconv_fwd_rb: bad channel split");
  PP_REQUIRE(c0 % 16 == 0 && c1 % 16 == 0, "pp_spconv_fwd_rb: input channels must be multiples of 16 (use pp_spconv_fwd)");
  PP_REQUIRE(cout % 4 == 0, "pp_spconv_fwd_rb: cout must be a multiple of 4");
  if (n_out == 0) return PP_OK;
  RbArgs a;
  a.in0 = in0; a.in1 = in1; a.wp = packed_weight; a.rb_off = rb_off; a.rb_in = rb_in; a.rb_out = rb_out;
  a.scale = scale; a.shift = shift; a.residual = residual; a.out = out; a.n_out = n_out;
  a.nblk = pp_rulebook_blocks(n_out); a.c0 = c0; a.c1 = c1; a.cout = cout; a.NT = (cout + 15) / 16; a.relu = relu;
  static const int rb_ver = getenv("PP_RB_VER") ? atoi(getenv("PP_RB_VER")) : 2;  // tuning knobs (A/B runs)
  if (rb_ver == 3) {
    const int S3 = (c0 + c1) / 16;
    hipStream_t s3 = pp_s(stream);
    if (S3 == 1) return rb3_launch<1, 6>(a, s3);
    return rb3_launch<2, 4>(a, s3);
  }
  // one 16-channel output tile per wave by default: 9 KiB of LDS per wave -> 16 waves per CU; measured 20-70 %
  // faster than two tiles per wave (8-12 waves per CU) even though A tiles are gathered once per column tile
  static const int force_ntw = getenv("PP_RB_NTW") ? atoi(getenv("PP_RB_NTW")) : 1;
  static const int small_stage = getenv("PP_RB_SMALL") ? atoi(getenv("PP_RB_SMALL")) : 0;
  const int ntw = force_ntw == 1 ? 1 : (a.NT >= 2 ? 2 : 1);
  const int groups = (a.NT + ntw - 1) / ntw;
  dim3 grid((unsigned)((a.nblk + 3) / 4), (unsigned)groups);
  hipStream_t s = pp_s(stream);
  const int S = (c0 + c1) / 16;
  if (ntw == 2) {
    if (S == 1) return rb_launch<2, 4, 1>(a, grid, s);
    if (S == 2) return rb_launch<2, 2, 2>(a, grid, s);
    return rb_launch<2, 1, 4>(a, grid, s);
  }
  if (small_stage) {
    if (S == 1) return rb_launch<1, 2, 1>(a, grid, s);
    return rb_launch<1, 1, 2>(a, grid, s);
  }
  if (S == 1) return rb_launch<1, 2, 1>(a, grid, s);  // Cin = 16: 2 KiB staging, 20 waves per CU
  if (S == 2) return rb_launch<1, 2, 2>(a, grid, s);
  return rb_launch<1, 1, 4>(a, grid, s);
}

