// K3d: tile scheduling at map-build time.
//
// The dense-offset convolution (pp_spconv2.hip) executes a kernel offset for a 16-row MFMA tile as soon as ONE of the
// tile's rows has that neighbour; with rows in plain block / Z-order only ~0.3 of the executed tile rows are real pairs on
// the fine levels (profiles/r01_w_layer_table.md).  Here every kernel map gets its own SLOT ORDER: inside windows of
// PP_MAP_WINDOW consecutive rows, output rows are sorted by neighbour mask so that the 16 rows of a tile
// -- and the 32 / 64 rows of a wave -- want the same offsets (measured on the bench scene: executed tile rows per pair
// 3.2 -> 1.8 on same-level maps, 2.4 -> 1.05 on transposed stride-2 maps, 4.6 -> 2.0 on strided maps).
//   * same-level maps: the slot order IS the physical row order of the level (pp_level_permute renumbers the level), so
//     the convolution needs no indirection at all;
//   * cross-level maps (strided / transposed): the map is stored slot-major (coalesced reads), `order[slot]` names the
//     physical output row (the convolution scatters 64-byte row segments inside a window -- L2 merges them).
// Windows are consecutive rows of the block / Z-order, so neighbours stay a few thousand rows apart (L2-resident).
// Mask bits are compared rarest first, PER WINDOW (the offset fewest rows of the window have is the most significant bit of
// the key; the first version used a fixed class order: corners, edges, faces, centre): rows that differ only in the
// window's common offsets end up next to each other.  The sort is a hand-written stable radix sort, one workgroup per
// window, rows in registers: ordered by (remapped mask, row in window) -- a total order, hence deterministic.
// Reference: none -- MinkowskiEngine keeps kernel maps as unordered (in, out) pair lists per offset; results of the
// convolution do not depend on the row order (every output row is still written exactly once, same summation order).
#include <algorithm>

#include "pp_common.h"

#define MO_MASK_BITS 27
static int g_window = 8192;      // rows per window = keys per workgroup (8 B of LDS each); pp_map_set_window

// ---- neighbour mask of a dense map ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_map_mask(const int32_t* __restrict__ nbr, int K, int64_t n, uint32_t* __restrict__ mask) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  uint32_t m = 0;
  for (int k = 0; k < K; ++k) m |= (nbr[(int64_t)k * n + o] >= 0 ? 1u : 0u) << k;
  mask[o] = m;
}
extern "C" int pp_map_mask(const int32_t* nbr, int32_t K, int64_t n_out, uint32_t* mask, pp_stream_t stream) {
  PP_REQUIRE((nbr && mask) || n_out == 0, "pp_map_mask: null pointer");
  PP_REQUIRE(K >= 1 && K <= 27, "pp_map_mask: K must be in [1,27]");
  if (n_out == 0) return PP_OK;
  hipLaunchKernelGGL(k_map_mask, dim3(pp_blocks(n_out, 256)), dim3(256), 0, pp_s(stream), nbr, K, n_out, mask);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- window sort ---------------------------------------------------------------------------------------------------------
// one workgroup per window of W rows, 8 rows per thread in registers: stable LSD radix sort of the remapped 27-bit masks,
// two bits per pass.  A pass: every thread counts its 8 digits in four packed 16-bit counters (one 64-bit word), the
// workgroup scans that word (wave scan by shuffles, the 16 wave totals through LDS), the rows move to
// base[digit] + rows with that digit in front of them through an LDS exchange.  Stable, so equal masks stay in row
// order: the result is sorted by (mask, row) -- a total order, hence deterministic.  ~170 vector instructions per row in
// all, against ~730 for the bitonic network this replaces (91 compare-exchange stages on 64-bit keys; 300 -> 110 us for
// the 10 M rows of the bench scene's finest level).
template <int W>
__global__ __launch_bounds__(W / 8) void k_window_sort(const uint32_t* __restrict__ mask, int64_t n, int32_t* __restrict__ order) {
  constexpr int NT = W / 8, NW = NT / 64;
  __shared__ uint32_t kl[W];
  __shared__ unsigned short il[W];
  __shared__ unsigned long long wtot[NW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t base = (int64_t)blockIdx.x * W;
  const int cnt = (int)((n - base) < W ? (n - base) : W);
  uint32_t key[8];
  int idx[8];
  // Significance of the 27 offsets in the sort key, PER WINDOW: the offset most rows of this window have is the least
  // significant bit, the rarest the most significant (ties: lower offset index lower).  A fixed order (centre < faces <
  // edges < corners) cannot know whether the window holds a floor, a wall or a pole; by frequency the rows that differ in
  // the window's common offsets stay neighbours: executed tile rows per useful pair 2.02 -> 1.90 on the bench scene
  // (profiles/tile_reuse_tradeoff.py).
  __shared__ int fcnt[MO_MASK_BITS];
  __shared__ int fpos[MO_MASK_BITS];
  if (t < MO_MASK_BITS) fcnt[t] = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = t * 8 + r;
    key[r] = e < cnt ? (mask[base + e] & 0x7FFFFFFu) : 0u;  // raw masks first (padding rows do not count)
    idx[r] = e;
  }
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < MO_MASK_BITS; ++k) {
    int c = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) c += __popcll(__ballot((key[r] >> k) & 1u));
    if (lane == 0 && c) atomicAdd(&fcnt[k], c);
  }
  __syncthreads();
  if (t < MO_MASK_BITS) {
    const int f = fcnt[t];
    int p = 0;
    for (int j = 0; j < MO_MASK_BITS; ++j) {
      const int fj = fcnt[j];
      p += (fj > f || (fj == f && j < t)) ? 1 : 0;
    }
    fpos[t] = p;
  }
  __syncthreads();
  // offsets no row of the window has take the highest positions of the key and are zero in every key: the radix passes stop
  // at the number of offsets that occur (an 8-wide transposed map's key has 11 bits, a window on a plane ~9 offsets)
  int nbits = 0;
  for (int k = 0; k < MO_MASK_BITS; ++k) nbits += fcnt[k] > 0 ? 1 : 0;
  nbits = __builtin_amdgcn_readfirstlane(nbits);
  {
    uint32_t out[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) out[r] = 0u;
#pragma unroll 1
    for (int k = 0; k < MO_MASK_BITS; ++k) {
      const int p = __builtin_amdgcn_readfirstlane(fpos[k]);
#pragma unroll
      for (int r = 0; r < 8; ++r) out[r] |= ((key[r] >> k) & 1u) << p;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) key[r] = (t * 8 + r) < cnt ? out[r] : 0x7FFFFFFu;  // padding: the largest key, behind its equals
  }
#pragma unroll 1
  for (int bit = 0; bit < nbits; bit += 2) {
    unsigned long long c = 0;
    int lr[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int sh = (int)((key[r] >> bit) & 3u) * 16;
      lr[r] = (int)((c >> sh) & 0xFFFFull);
      c += 1ull << sh;
    }
    unsigned long long incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long v = (unsigned long long)__shfl_up((long long)incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned long long woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const unsigned long long v = wtot[w];
      if (w < wave) woff += v;
      total += v;
    }
    // start of every digit's range: rows with smaller digits
    const unsigned long long t0 = total & 0xFFFFull, t1 = (total >> 16) & 0xFFFFull, t2 = (total >> 32) & 0xFFFFull;
    const unsigned long long pb = incl - c + woff + ((t0 << 16) | ((t0 + t1) << 32) | ((t0 + t1 + t2) << 48));
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int sh = (int)((key[r] >> bit) & 3u) * 16;
      const int pos = (int)((pb >> sh) & 0xFFFFull) + lr[r];
      kl[pos] = key[r];
      il[pos] = (unsigned short)idx[r];
    }
    __syncthreads();
    const uint4 k0 = *(const uint4*)&kl[t * 8], k1 = *(const uint4*)&kl[t * 8 + 4];
    const uint4 i4 = *(const uint4*)&il[t * 8];
    key[0] = k0.x; key[1] = k0.y; key[2] = k0.z; key[3] = k0.w;
    key[4] = k1.x; key[5] = k1.y; key[6] = k1.z; key[7] = k1.w;
    idx[0] = i4.x & 0xFFFF; idx[1] = i4.x >> 16; idx[2] = i4.y & 0xFFFF; idx[3] = i4.y >> 16;
    idx[4] = i4.z & 0xFFFF; idx[5] = i4.z >> 16; idx[6] = i4.w & 0xFFFF; idx[7] = i4.w >> 16;
    __syncthreads();  // everything is back in registers before the next pass refills the exchange buffers
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int e = t * 8 + r;
    if (e < cnt) order[base + e] = (int32_t)(base + idx[r]);
  }
}


// ---- window sort, 16384 / 32768 rows per window -------------------------------------------------------------------------
// The same stable LSD radix sort with W / 1024 rows per thread.  Keys and row indices no longer fit the LDS side by side
// (6 bytes x 32768), and a workgroup that takes most of a CU's LDS waits for the convolution workgroups it runs beside to
// drain: the exchange goes through ONE 16-bit buffer (64 KiB at 32768 rows) in up to three rounds per pass -- the row index,
// the high half of the key, and the low half while later passes still need it (from bit 16 on they do not).
template <int W, int NT>
__global__ __launch_bounds__(NT) void k_window_sort_big(const uint32_t* __restrict__ mask, int64_t n, int32_t* __restrict__ order) {
  constexpr int RPT = W / NT, NW = NT / 64;
  static_assert(RPT % 8 == 0 && W <= 32768, "rows per thread in groups of eight; 15-bit row index");
  __shared__ __attribute__((aligned(16))) unsigned short xl[W];
  __shared__ unsigned long long wtot[NW];
  __shared__ int fcnt[MO_MASK_BITS];
  __shared__ int fpos[MO_MASK_BITS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t base = (int64_t)blockIdx.x * W;
  const int cnt = (int)((n - base) < W ? (n - base) : W);
  // per row: the key and one word holding the row index (low half) and, during a pass, the row's new position (high half).
  // 32768 rows run as 512 threads x 64 rows: a 1024-thread workgroup has 128 registers per thread, 32 rows x (key, index /
  // position) + the 64-bit packed counters do not fit them (121 spilled registers), a 512-thread one has 256
  uint32_t key[RPT], ip[RPT];
  if (t < MO_MASK_BITS) fcnt[t] = 0;
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int e = t * RPT + r;
    const uint32_t m = mask[base + (e < cnt ? e : 0)] & 0x7FFFFFFu;  // branch-free: a select, not a guarded load
    key[r] = e < cnt ? m : 0u;
  }
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < MO_MASK_BITS; ++k) {
    int c = 0;
#pragma unroll
    for (int r = 0; r < RPT; ++r) c += __popcll(__ballot((key[r] >> k) & 1u));
    if (lane == 0 && c) atomicAdd(&fcnt[k], c);
  }
  __syncthreads();
  if (t < MO_MASK_BITS) {  // rarest offset of the window = most significant key bit (k_window_sort)
    const int f = fcnt[t];
    int p = 0;
    for (int j = 0; j < MO_MASK_BITS; ++j) {
      const int fj = fcnt[j];
      p += (fj > f || (fj == f && j < t)) ? 1 : 0;
    }
    fpos[t] = p;
  }
  __syncthreads();
  int nbits = 0;
  for (int k = 0; k < MO_MASK_BITS; ++k) nbits += fcnt[k] > 0 ? 1 : 0;
  nbits = __builtin_amdgcn_readfirstlane(nbits);
#pragma unroll
  for (int r = 0; r < RPT; ++r) ip[r] = 0u;
#pragma unroll 1
  for (int k = 0; k < MO_MASK_BITS; ++k) {
    const int p = __builtin_amdgcn_readfirstlane(fpos[k]);
#pragma unroll
    for (int r = 0; r < RPT; ++r) ip[r] |= ((key[r] >> k) & 1u) << p;
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    key[r] = (t * RPT + r) < cnt ? ip[r] : 0x7FFFFFFu;  // padding: the largest key, behind its equals
    ip[r] = (uint32_t)(t * RPT + r);
  }

  // one exchange round: every row's 16-bit value goes to its new position, the thread reads back its RPT consecutive ones
#define MOB_EXCHANGE(PUT, GET)                                                              \
  {                                                                                         \
    _Pragma("unroll") for (int r = 0; r < RPT; ++r) {                                       \
      asm volatile("" : "+v"(ip[r])); /* the address is re-derived per round, not kept (a register per row) */ \
      xl[ip[r] >> 16] = (unsigned short)(PUT);                                              \
      if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0); /* keeps the 32 addresses from being formed at once */ \
    }                                                                                       \
    __syncthreads();                                                                        \
    _Pragma("unroll") for (int v = 0; v < RPT / 8; ++v) {                                   \
      __builtin_amdgcn_sched_barrier(0);                                                    \
      const uint4 x_ = *(const uint4*)&xl[t * RPT + 8 * v];                                 \
      const uint32_t w_[4] = {x_.x, x_.y, x_.z, x_.w};                                      \
      _Pragma("unroll") for (int h = 0; h < 8; ++h) {                                       \
        const int r = 8 * v + h;                                                            \
        const uint32_t g_ = (w_[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu;                       \
        GET;                                                                                \
      }                                                                                     \
    }                                                                                       \
    __syncthreads();                                                                        \
  }
#pragma unroll 1
  for (int bit = 0; bit < nbits; bit += 2) {
    unsigned long long c = 0;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int sh = (int)((key[r] >> bit) & 3u) * 16;
      ip[r] = (ip[r] & 0xFFFFu) | ((uint32_t)((c >> sh) & 0xFFFFull) << 16);
      c += 1ull << sh;
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (hipcc would form all 32 shifted ones first: 64 registers)
    }
    unsigned long long incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long v = (unsigned long long)__shfl_up((long long)incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned long long woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const unsigned long long v = wtot[w];
      if (w < wave) woff += v;
      total += v;
    }
    const unsigned long long t0 = total & 0xFFFFull, t1 = (total >> 16) & 0xFFFFull, t2 = (total >> 32) & 0xFFFFull;
    const unsigned long long pb = incl - c + woff + ((t0 << 16) | ((t0 + t1) << 32) | ((t0 + t1 + t2) << 48));
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      asm volatile("" : "+v"(key[r]));  // the digit is re-derived, not kept from the counting loop (a register per row)
      const int sh = (int)((key[r] >> bit) & 3u) * 16;
      ip[r] += (uint32_t)((pb >> sh) & 0xFFFFull) << 16;  // position < W <= 2^15: no carry out of the word
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // the new row indices land in the low halves only after every old one has been sent: a second word per row would be
    // the third array, so they wait in the exchange buffer's readback (g_) and replace the low half in place -- the high
    // half (position) is still needed by the key rounds
    MOB_EXCHANGE(ip[r], ip[r] = (ip[r] & 0xFFFF0000u) | g_);
    const bool lo = bit + 2 < 16 && bit + 2 < nbits;  // a later pass still reads the low half
    if (nbits > 16) {
      if (lo) {
        // the received high half replaces the one just sent; the low half of the word is still the OLD row's until its round
        MOB_EXCHANGE(key[r] >> 16, key[r] = (g_ << 16) | (key[r] & 0xFFFFu));
        MOB_EXCHANGE(key[r], key[r] = (key[r] & 0xFFFF0000u) | g_);
      } else {
        MOB_EXCHANGE(key[r] >> 16, key[r] = g_ << 16);
      }
    } else if (lo) {
      MOB_EXCHANGE(key[r], key[r] = g_);
    }
  }
#undef MOB_EXCHANGE
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int e = t * RPT + r;
    if (e < cnt) order[base + e] = (int32_t)(base + (ip[r] & 0xFFFFu));
  }
}

// 4096 is NOT served (round 6).  k_window_sort<4096> -- the one instantiation with 512 threads = 8 waves per workgroup, 72 VGPRs and
// 24.3 KiB of LDS -- returns orders whose rows leave their window (16-bit row indices overwritten by keys: a scatter position
// beyond the window) in 20 - 34 of 40 launches when k_kernel_map_bi or k_map_permute_big runs on ANOTHER stream at the same time, and
// never alone, never beside the other sorts, the coarsening kernels or a fill, while 1024 / 2048 / 8192 / 16384 / 32768 pass 40 of 40
// under the same load (profiles/sort_race_probe.py -> profiles/r06_sort_race_probe.txt; rocgdb put the bench's memory faults of the
// window sweeps of rounds 4 and 6 into the k_map_permute_win<4096> launch that consumed such an order).  The ISA was read barrier by
// barrier without finding a missing one; until the cause is known the size is refused instead of being left as a trap.
static bool map_window_ok(int w) { return w >= 1024 && w <= 32768 && (w & (w - 1)) == 0 && w != 4096; }
extern "C" int32_t pp_map_window(void) { return g_window; }
extern "C" int pp_map_set_window(int32_t window) {
  PP_REQUIRE(map_window_ok(window), "pp_map_set_window: 1024, 2048, 8192, 16384 or 32768");
  g_window = window;
  return PP_OK;
}

static int map_order_launch(const uint32_t* mask, int64_t n, int window, int32_t* order, hipStream_t s) {
  switch (window) {
    case 1024: hipLaunchKernelGGL(k_window_sort<1024>, dim3(pp_blocks(n, 1024)), dim3(128), 0, s, mask, n, order); break;
    case 2048: hipLaunchKernelGGL(k_window_sort<2048>, dim3(pp_blocks(n, 2048)), dim3(256), 0, s, mask, n, order); break;
    case 8192: hipLaunchKernelGGL(k_window_sort<8192>, dim3(pp_blocks(n, 8192)), dim3(1024), 0, s, mask, n, order); break;
    case 16384: hipLaunchKernelGGL((k_window_sort_big<16384, 1024>), dim3(pp_blocks(n, 16384)), dim3(1024), 0, s, mask, n, order); break;
    default: hipLaunchKernelGGL((k_window_sort_big<32768, 512>), dim3(pp_blocks(n, 32768)), dim3(512), 0, s, mask, n, order); break;
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_map_order(const uint32_t* mask, int64_t n, int32_t* order, pp_stream_t stream) {
  PP_REQUIRE((mask && order) || n == 0, "pp_map_order: null pointer");
  PP_REQUIRE(n < (1ll << 31), "pp_map_order: more than 2^31 rows");
  if (n == 0) return PP_OK;
  return map_order_launch(mask, n, g_window, order, pp_s(stream));
}
// the same with an explicit window (1024 .. 32768 rows, a power of two): the levels' same-level maps take larger windows than
// the cross-level ones, whose convolutions scatter their output rows inside a window
extern "C" int pp_map_order_window(const uint32_t* mask, int64_t n, int32_t window, int32_t* order, pp_stream_t stream) {
  PP_REQUIRE((mask && order) || n == 0, "pp_map_order_window: null pointer");
  PP_REQUIRE(n < (1ll << 31), "pp_map_order_window: more than 2^31 rows");
  PP_REQUIRE(map_window_ok(window), "pp_map_order_window: window must be 1024, 2048, 8192, 16384 or 32768");
  if (n == 0) return PP_OK;
  return map_order_launch(mask, n, window, order, pp_s(stream));
}

// ---- compact form of a same-level map -----------------------------------------------------------------------------------------
// The convolution's prologue needs, per wave, the (offset, row) -> neighbour row table of its 32 or 64 output rows.  The dense map
// stores 27 entries per row, of which a surface-like level has 5 - 10: 108 bytes per row where 4 (mask) + 6 per present entry
// would do, and the map is the largest stream of the <= 32-channel layers (108 of 236 bytes per row at 16 channels).  Compact form:
// entries grouped by chunks of 32 output rows (offset-major inside a chunk), start[chunk] their offsets, tag[e] = offset << 6 |
// output row & 63, mask[row] = the row's offsets.
__global__ __launch_bounds__(256) void k_cmap_mask_count(const int32_t* __restrict__ nbr, int K, int64_t n, uint32_t* __restrict__ mask,
                                                         int32_t* __restrict__ cnt) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t m = 0;
  if (o < n)
    for (int k = 0; k < K; ++k) m |= (nbr[(int64_t)k * n + o] >= 0 ? 1u : 0u) << k;
  if (o < n) mask[o] = m;
  int c = __popc(m);
  for (int d = 16; d > 0; d >>= 1) c += __shfl_xor(c, d);  // over the 32 lanes of a chunk
  if ((threadIdx.x & 31) == 0 && o < n) cnt[o >> 5] = c;
}
__global__ __launch_bounds__(256) void k_cmap_write(const int32_t* __restrict__ nbr, int K, int64_t n, const int32_t* __restrict__ start,
                                                    uint32_t* __restrict__ ent, uint16_t* __restrict__ tag) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // lane = row; a wave covers two chunks
  const int lane = threadIdx.x & 63, half = lane >> 5;
  const int64_t oc = o < n ? o : n - 1;
  int run = start[oc >> 5];
  for (int k = 0; k < K; ++k) {
    const int v = o < n ? nbr[(int64_t)k * n + o] : -1;
    const unsigned long long b = __ballot(v >= 0);
    const uint32_t hb = half ? (uint32_t)(b >> 32) : (uint32_t)b;
    if (v >= 0) {
      const int pos = run + __popc(hb & ((1u << (lane & 31)) - 1u));
      ent[pos] = (uint32_t)v;
      tag[pos] = (uint16_t)((k << 6) | (int)(o & 63));
    }
    run += __popc(hb);
  }
}
extern "C" size_t pp_map_compact_workspace(int64_t n_out) {
  const int64_t chunks = (std::max<int64_t>(n_out, 1) + 31) / 32;
  return pp_align((size_t)chunks * 4) + pp_scan_workspace(chunks) + 256;
}
// step 1: mask [n_out], start [chunks + 1] (chunks = ceil(n_out / 32); start[chunks] = the map's pairs); the caller reads
// start[chunks] and allocates entries (uint32) / tags (uint16) for step 2
extern "C" int pp_map_compact_count(const int32_t* nbr, int32_t K, int64_t n_out, uint32_t* mask, int32_t* start, void* workspace,
                                    size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(nbr && mask && start && n_out > 0, "pp_map_compact_count: null pointer / empty map");
  PP_REQUIRE(K >= 1 && K <= 27, "pp_map_compact_count: K must be in [1,27]");
  PP_REQUIRE((double)K * (double)n_out < 2147483000.0, "pp_map_compact_count: more than 2^31 map entries");
  if (workspace_bytes < pp_map_compact_workspace(n_out)) return PP_ERR_WORKSPACE;
  const int64_t chunks = (n_out + 31) / 32;
  hipStream_t s = pp_s(stream);
  PPArena ar(workspace, workspace_bytes);
  int32_t* cnt = ar.take<int32_t>((size_t)chunks);
  PP_REQUIRE(cnt, "pp_map_compact_count: workspace");
  hipLaunchKernelGGL(k_cmap_mask_count, dim3(pp_blocks(n_out, 256)), dim3(256), 0, s, nbr, K, n_out, mask, cnt);
  PP_LAUNCH_CHECK();
  return pp_exclusive_scan_i32(cnt, start, chunks, start + chunks, ar.cur(), ar.left(), s);
}
extern "C" int pp_map_compact_write(const int32_t* nbr, int32_t K, int64_t n_out, const int32_t* start, uint32_t* entries,
                                    uint16_t* tags, pp_stream_t stream) {
  PP_REQUIRE(nbr && start && entries && tags && n_out > 0, "pp_map_compact_write: null pointer / empty map");
  PP_REQUIRE(K >= 1 && K <= 27, "pp_map_compact_write: K must be in [1,27]");
  hipLaunchKernelGGL(k_cmap_write, dim3(pp_blocks(n_out, 256)), dim3(256), 0, pp_s(stream), nbr, K, n_out, start, entries, tags);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- applying an order -----------------------------------------------------------------------------------------------------
// out[k][s] = T(nbr[k][order[s]]),  T(v) = v < 0 ? -1 : (translate ? translate[v] : v).
// `order` is window-local (pp_map_order), so a workgroup stages the window's slices in LDS with coalesced loads, gathers
// from LDS and writes coalesced: the 27 scattered 4-byte global reads per row of the naive form (14.7 ms per bench step)
// become LDS reads.  window = 0 selects the naive form (any order / no order).
template <int W>
__global__ __launch_bounds__(1024) void k_map_permute_win(const int32_t* __restrict__ nbr, int K, int64_t n,
                                                          const int32_t* __restrict__ order, const int32_t* __restrict__ translate,
                                                          int64_t n_tr, int32_t* __restrict__ out) {
  // one workgroup per window, all K offsets: the window's order (16-bit local) and, with `translate`, the window's own
  // slice of it stay in LDS -- most neighbours of a window's rows live in the window itself (rows are in block order), so
  // the translation is an LDS read for them and a global gather only across window borders; the K slices of the map
  // stream through registers (next slice's loads in flight while the current one is permuted)
  constexpr int NT = 1024, PER = (W + NT - 1) / NT;
  __shared__ int32_t cur[W];
  __shared__ int32_t tr[W];
  __shared__ unsigned short ord[W];
  const int64_t base = (int64_t)blockIdx.x * W;
  const int cnt = (int)((n - base) < W ? (n - base) : W);
  for (int j = threadIdx.x; j < cnt; j += NT) {
    ord[j] = (unsigned short)(order[base + j] - (int32_t)base);
    // (`translate` has n_tr entries -- the rows of the level the map's VALUES name, which a cross-level map's n output rows can
    // outnumber: the preload stops at its end; values are < n_tr by construction)
    if (translate) tr[j] = base + j < n_tr ? translate[base + j] : 0;
  }
  int32_t stage[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int j = threadIdx.x + NT * u;
    stage[u] = j < cnt ? nbr[base + j] : -1;
  }
  for (int k = 0; k < K; ++k) {
    if (k) __syncthreads();  // the previous slice has been read
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int j = threadIdx.x + NT * u;
      if (j < cnt) cur[j] = stage[u];
    }
    __syncthreads();
    if (k + 1 < K) {
      const int32_t* src = nbr + (int64_t)(k + 1) * n + base;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int j = threadIdx.x + NT * u;
        stage[u] = j < cnt ? src[j] : -1;
      }
    }
    int32_t* dst = out + (int64_t)k * n + base;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int j = threadIdx.x + NT * u;
      if (j < cnt) {
        int32_t v = cur[ord[j]];
        if (translate && v >= 0) {
          const int64_t l = (int64_t)v - base;
          v = (l >= 0 && l < cnt && v < n_tr) ? tr[l] : translate[v];
        }
        dst[j] = v;
      }
    }
  }
}
// 16384 / 32768 rows per window: the window's slice of one offset (64 / 128 KiB) does not fit the LDS beside the convolution's
// workgroups, so it passes through a 32 KiB buffer in quarters of 8192 source rows; every thread keeps the (window-local)
// source rows of its W / 1024 slots in registers and picks its values out of the quarter that holds them.  Translation is a
// global gather (the window's slice of `translate` would be another 128 KiB).
template <int W>
__global__ __launch_bounds__(1024) void k_map_permute_big(const int32_t* __restrict__ nbr, int K, int64_t n,
                                                          const int32_t* __restrict__ order, const int32_t* __restrict__ translate,
                                                          int32_t* __restrict__ out) {
  constexpr int NT = 1024, PER = W / NT, QW = 8192, SPT = QW / NT;
  __shared__ int32_t cur[QW];
  const int t = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * W;
  const int cnt = (int)((n - base) < W ? (n - base) : W);
  const int nq = (cnt + QW - 1) / QW;
  uint32_t ord2[PER / 2];  // two 16-bit window-local source rows per word
#pragma unroll
  for (int u = 0; u < PER / 2; ++u) ord2[u] = 0u;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int j = t + NT * u;
    // slots past the end: 0xFFFF, a row of the last quarter that the stores below skip
    ord2[u >> 1] |= (j < cnt ? (uint32_t)(order[base + j] - (int32_t)base) : 0xFFFFu) << ((u & 1) * 16);
  }
  int32_t stage[SPT];
#pragma unroll
  for (int v = 0; v < SPT; ++v) {
    const int e = t + NT * v;
    stage[v] = e < cnt ? nbr[base + e] : -1;
  }
  for (int k = 0; k < K; ++k) {
    int32_t val[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) val[u] = -1;
    for (int q = 0; q < nq; ++q) {
      __syncthreads();  // the previous quarter has been read
#pragma unroll
      for (int v = 0; v < SPT; ++v) cur[t + NT * v] = stage[v];
      __syncthreads();
      // next quarter's (or next offset's first quarter's) loads in flight while this one is picked apart
      const int qn = q + 1 < nq ? q + 1 : 0, kn = q + 1 < nq ? k : k + 1;
      if (kn < K) {
        const int32_t* src = nbr + (int64_t)kn * n + base;
#pragma unroll
        for (int v = 0; v < SPT; ++v) {
          const int e = qn * QW + t + NT * v;
          stage[v] = e < cnt ? src[e] : -1;
        }
      }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const uint32_t o = (ord2[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu;
        if ((o >> 13) == (uint32_t)q) val[u] = cur[o & (QW - 1)];
      }
    }
    int32_t* dst = out + (int64_t)k * n + base;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int j = t + NT * u;
      if (j < cnt) {
        int32_t v = val[u];
        if (translate && v >= 0) v = translate[v];
        dst[j] = v;
      }
    }
  }
}
__global__ __launch_bounds__(256) void k_map_permute(const int32_t* __restrict__ nbr, int K, int64_t n,
                                                     const int32_t* __restrict__ order, const int32_t* __restrict__ translate,
                                                     int32_t* __restrict__ out) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int64_t o = order ? (int64_t)order[s] : s;
  for (int k = 0; k < K; ++k) {
    int32_t v = nbr[(int64_t)k * n + o];
    if (translate && v >= 0) v = translate[v];
    out[(int64_t)k * n + s] = v;
  }
}
// window: the window size `order` was built with (pp_map_window() at that time), or 0 for an arbitrary / NULL order
extern "C" int pp_map_permute(const int32_t* nbr, int32_t K, int64_t n_out, const int32_t* order, const int32_t* translate,
                              int64_t translate_rows, int32_t window, int32_t* out, pp_stream_t stream) {
  PP_REQUIRE(!translate || translate_rows > 0, "pp_map_permute: translate needs its number of rows");
  PP_REQUIRE((nbr && out) || n_out == 0, "pp_map_permute: null pointer");
  PP_REQUIRE(K >= 1 && K <= 27, "pp_map_permute: K must be in [1,27]");
  PP_REQUIRE(nbr != out, "pp_map_permute: in-place permutation is not supported");
  if (n_out == 0) return PP_OK;
  hipStream_t s = pp_s(stream);
  if (order && window > 0) {
    PP_REQUIRE(map_window_ok(window), "pp_map_permute: bad window");
    const dim3 grid(pp_blocks(n_out, window));
    switch (window) {
      case 16384: hipLaunchKernelGGL(k_map_permute_big<16384>, grid, dim3(1024), 0, s, nbr, K, n_out, order, translate, out); break;
      case 32768: hipLaunchKernelGGL(k_map_permute_big<32768>, grid, dim3(1024), 0, s, nbr, K, n_out, order, translate, out); break;
      case 1024: hipLaunchKernelGGL(k_map_permute_win<1024>, grid, dim3(1024), 0, s, nbr, K, n_out, order, translate, translate_rows, out); break;
      case 2048: hipLaunchKernelGGL(k_map_permute_win<2048>, grid, dim3(1024), 0, s, nbr, K, n_out, order, translate, translate_rows, out); break;
      default: hipLaunchKernelGGL(k_map_permute_win<8192>, grid, dim3(1024), 0, s, nbr, K, n_out, order, translate, translate_rows, out); break;
    }
  } else
    hipLaunchKernelGGL(k_map_permute, dim3(pp_blocks(n_out, 256)), dim3(256), 0, s, nbr, K, n_out, order, translate, out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// coords_out[s] = coords[order[s]],  inverse[order[s]] = s,  mask_out[s] = mask[order[s]] (mask optional)
__global__ __launch_bounds__(256) void k_level_permute(const int4* __restrict__ coords, int64_t n, const int32_t* __restrict__ order,
                                                       int4* __restrict__ coords_out, int32_t* __restrict__ inverse) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int32_t o = order[s];
  coords_out[s] = coords[o];
  inverse[o] = (int32_t)s;
}
extern "C" int pp_level_permute(const int32_t* coords, int64_t n, const int32_t* order, int32_t* coords_out, int32_t* inverse,
                                pp_stream_t stream) {
  PP_REQUIRE((coords && order && coords_out && inverse) || n == 0, "pp_level_permute: null pointer");
  if (n == 0) return PP_OK;
  hipLaunchKernelGGL(k_level_permute, dim3(pp_blocks(n, 256)), dim3(256), 0, pp_s(stream), (const int4*)coords, n, order,
                     (int4*)coords_out, inverse);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// caller <-> internal row permutation of the input level in one pass: perm_out[s] = perm32[order[s]] (either may be NULL =
// identity), inv_out[perm_out[s]] = s.  Replaces three elementwise torch launches over all input rows.
__global__ __launch_bounds__(256) void k_compose_perm(const int32_t* __restrict__ perm32, const int32_t* __restrict__ order, int64_t n,
                                                      int64_t* __restrict__ perm_out, int64_t* __restrict__ inv_out) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int64_t o = order ? (int64_t)order[s] : s;
  const int64_t p = perm32 ? (int64_t)perm32[o] : o;
  perm_out[s] = p;
  inv_out[p] = s;
}
extern "C" int pp_compose_perm(const int32_t* perm32, const int32_t* order, int64_t n, int64_t* perm_out, int64_t* inv_out,
                               pp_stream_t stream) {
  PP_REQUIRE((perm_out && inv_out) || n == 0, "pp_compose_perm: null output");
  if (n == 0) return PP_OK;
  hipLaunchKernelGGL(k_compose_perm, dim3(pp_blocks(n, 256)), dim3(256), 0, pp_s(stream), perm32, order, n, perm_out, inv_out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
