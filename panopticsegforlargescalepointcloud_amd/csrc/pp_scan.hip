// Library plumbing: error reporting, version, the device-wide exclusive scan and radix sort every compaction / ordering of
// the library goes through, and the HBM triad used by bench.py to confirm the roofline denominator.
#include <stdarg.h>

#include "pp_common.h"

static thread_local char g_err[512] = "";

void pp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pp_last_error(void) { return g_err; }
extern "C" const char* pp_version(void) { return "panoptic_hip 1 gfx950"; }

__global__ __launch_bounds__(256) void k_triad(float4* __restrict__ a, const float4* __restrict__ b,
                                               const float4* __restrict__ c, float s, int64_t n4) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 x = b[i], y = c[i];
    a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
  }
}
extern "C" int pp_triad(float* a, const float* b, const float* c, float s, int64_t n, pp_stream_t stream) {
  PP_REQUIRE(n % 4 == 0, "pp_triad: n must be a multiple of 4");
  int64_t n4 = n / 4;
  unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 256 * 64);
  if (blocks == 0) return PP_OK;
  hipLaunchKernelGGL(k_triad, dim3(blocks), dim3(256), 0, pp_s(stream), (float4*)a, (const float4*)b,
                     (const float4*)c, s, n4);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// Device-wide exclusive prefix sum (int32): reduce-then-scan over tiles of 4096 elements.
//   k_scan_sums   one workgroup per tile: the tile's sum
//   (the tile sums are scanned by the same three steps, recursively: one level reaches 16 M elements, two 2^31)
//   k_scan_tiles  one workgroup per tile: 16 elements per thread in registers, serial prefix in the thread, shuffle scan of
//                 the thread totals in the wave, the 4 wave totals through LDS, plus the tile's offset
// In place (in == out) is fine: a tile is read completely before it is written and the sums are taken first.
// ---------------------------------------------------------------------------------------------
#define SC_NT 256
#define SC_IPT 16
#define SC_TILE (SC_NT * SC_IPT)

__device__ __forceinline__ int sc_block_exclusive(int v, int* total) {  // exclusive scan of one value per thread over the workgroup
  __shared__ int wsum[SC_NT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o);
    if (lane >= o) incl += u;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SC_NT / 64; ++w) {
    const int u = wsum[w];
    if (w < wave) woff += u;
    tot += u;
  }
  *total = tot;
  return incl - v + woff;
}
__global__ __launch_bounds__(SC_NT) void k_scan_sums(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ sums) {
  const int64_t base = (int64_t)blockIdx.x * SC_TILE;
  int v = 0;
#pragma unroll
  for (int r = 0; r < SC_IPT; ++r) {
    const int64_t e = base + (int64_t)r * SC_NT + threadIdx.x;  // strided: coalesced, the order inside the sum is irrelevant
    v += e < n ? in[e] : 0;
  }
  int tot;
  sc_block_exclusive(v, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(SC_NT) void k_scan_tiles(const int32_t* in, int32_t* out, int64_t n,
                                                      const int32_t* __restrict__ offs, int32_t* __restrict__ total) {
  const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_IPT;
  int v[SC_IPT];
  int sum = 0;
#pragma unroll
  for (int r = 0; r < SC_IPT; ++r) {
    v[r] = base + r < n ? in[base + r] : 0;
    sum += v[r];
  }
  int tot;
  int run = sc_block_exclusive(sum, &tot) + (offs ? offs[blockIdx.x] : 0);
#pragma unroll
  for (int r = 0; r < SC_IPT; ++r) {
    if (base + r < n) {
      out[base + r] = run;
      if (total && base + r == n - 1) total[0] = run + v[r];
    }
    run += v[r];
  }
}

size_t pp_scan_workspace(int64_t n) {
  size_t bytes = 256;
  for (int64_t m = std::max<int64_t>(n, 1); m > SC_TILE;) {
    m = (m + SC_TILE - 1) / SC_TILE;
    bytes += pp_align(2 * (size_t)m * 4);  // tile sums + their scan
  }
  return bytes;
}

int pp_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total, void* ws, size_t ws_bytes,
                          hipStream_t stream) {
  if (n <= 0) {
    if (total) PP_HIP(hipMemsetAsync(total, 0, sizeof(int32_t), stream));
    return PP_OK;
  }
  if (ws_bytes < pp_scan_workspace(n)) return PP_ERR_WORKSPACE;
  const int64_t nb = (n + SC_TILE - 1) / SC_TILE;
  const int32_t* offs = nullptr;
  if (nb > 1) {
    int32_t* sums = (int32_t*)ws;
    int32_t* scanned = sums + nb;
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(SC_NT), 0, stream, in, n, sums);
    PP_LAUNCH_CHECK();
    const size_t used = pp_align((size_t)2 * nb * 4);
    int rc = pp_exclusive_scan_i32(sums, scanned, nb, nullptr, (char*)ws + used, ws_bytes - used, stream);
    if (rc) return rc;
    offs = scanned;
  }
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)nb), dim3(SC_NT), 0, stream, in, out, n, offs, total);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// Device-wide stable LSD radix sort of (key, int32 value) pairs on the key bits [0, end_bit), 8 bits per pass.  A pass:
//   k_rs_hist     one workgroup per tile of 2048 pairs: digit counts (LDS atomics) -> hist[digit][tile]
//   exclusive scan of hist (digit-major), i.e. the first output position of every (digit, tile)
//   k_rs_scatter  one workgroup per tile, 8 pairs per thread: the tile is sorted by the digit with four stable 2-bit
//                 splits (packed 16-bit counters in one 64-bit word, scanned by shuffles + LDS -- the scheme of the map
//                 window sort, pp_maporder.hip), on 32-bit words (digit | position in the tile); keys and values wait in
//                 LDS and are written out in sorted order, so every run of equal digits is one contiguous store
// Passes alternate between the output and a temporary pair of arrays so that the last one writes keys_out / vals_out.
// ---------------------------------------------------------------------------------------------
#define RS_NT 256
#define RS_IPT 8
#define RS_TILE (RS_NT * RS_IPT)

template <typename K>
__global__ __launch_bounds__(RS_NT) void k_rs_hist(const K* __restrict__ keys, int64_t n, int shift, uint32_t dmask, int64_t nb,
                                                   int32_t* __restrict__ hist) {
  __shared__ int h[256];
  if (threadIdx.x < 256) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_IPT; ++r) {
    const int64_t e = base + (int64_t)r * RS_NT + threadIdx.x;
    if (e < n) atomicAdd(&h[(int)((uint32_t)(keys[e] >> shift) & dmask)], 1);
  }
  __syncthreads();
  if (threadIdx.x < 256) hist[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

template <typename K>
__global__ __launch_bounds__(RS_NT) void k_rs_scatter(const K* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                                                      K* __restrict__ keys_out, int32_t* __restrict__ vals_out, int64_t n,
                                                      int shift, uint32_t dmask, int64_t nb,
                                                      const int32_t* __restrict__ goff) {
  __shared__ K ks[RS_TILE];
  __shared__ int32_t vs[RS_TILE];
  __shared__ uint32_t ex[RS_TILE];
  __shared__ unsigned long long wtot[RS_NT / 64];
  __shared__ int first[256], gbase[256];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  const int cnt = (int)((n - base) < RS_TILE ? (n - base) : RS_TILE);
  if (t < 256) gbase[t] = goff[(int64_t)t * nb + blockIdx.x];
  uint32_t pk[RS_IPT];  // digit << 12 | position in the tile
#pragma unroll
  for (int r = 0; r < RS_IPT; ++r) {
    const int e = t * RS_IPT + r;
    K key = 0;
    int32_t val = 0;
    if (e < cnt) {
      key = keys_in[base + e];
      val = vals_in[base + e];
    }
    ks[e] = key;
    vs[e] = val;
    // rows past the end carry the largest digit: the sort is stable, so they end up behind every real row
    pk[r] = ((e < cnt ? (uint32_t)(key >> shift) & dmask : 255u) << 12) | (uint32_t)e;
  }
#pragma unroll 1
  for (int bit = 12; bit < 20; bit += 2) {
    unsigned long long c = 0;
    int lr[RS_IPT];
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
      const int sh = (int)((pk[r] >> bit) & 3u) * 16;
      lr[r] = (int)((c >> sh) & 0xFFFFull);
      c += 1ull << sh;
    }
    unsigned long long incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long u = (unsigned long long)__shfl_up((long long)incl, o);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned long long woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < RS_NT / 64; ++w) {
      const unsigned long long u = wtot[w];
      if (w < wave) woff += u;
      total += u;
    }
    const unsigned long long t0 = total & 0xFFFFull, t1 = (total >> 16) & 0xFFFFull, t2 = (total >> 32) & 0xFFFFull;
    const unsigned long long pb = incl - c + woff + ((t0 << 16) | ((t0 + t1) << 32) | ((t0 + t1 + t2) << 48));
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
      const int sh = (int)((pk[r] >> bit) & 3u) * 16;
      ex[(int)((pb >> sh) & 0xFFFFull) + lr[r]] = pk[r];
    }
    __syncthreads();
    const uint4 a = *(const uint4*)&ex[t * RS_IPT], b = *(const uint4*)&ex[t * RS_IPT + 4];
    pk[0] = a.x; pk[1] = a.y; pk[2] = a.z; pk[3] = a.w;
    pk[4] = b.x; pk[5] = b.y; pk[6] = b.z; pk[7] = b.w;
    if (bit + 2 < 20) __syncthreads();  // (after the last split ex[] keeps the sorted tile for the run starts below)
  }
  // first position of every digit's run in the sorted tile
#pragma unroll
  for (int r = 0; r < RS_IPT; ++r) {
    const int p = t * RS_IPT + r;
    const uint32_t d = pk[r] >> 12;
    if (p == 0 || (ex[p - 1] >> 12) != d) first[d] = p;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_IPT; ++r) {
    const int p = r * RS_NT + t;  // sorted position, strided over the threads: a run of equal digits is a contiguous store
    if (p < cnt) {
      const uint32_t w = ex[p];
      const int d = (int)(w >> 12), src = (int)(w & 4095u);
      const int64_t pos = (int64_t)gbase[d] + (p - first[d]);
      keys_out[pos] = ks[src];
      vals_out[pos] = vs[src];
    }
  }
}

static size_t rs_hist_ints(int64_t n) { return (size_t)256 * (size_t)((std::max<int64_t>(n, 1) + RS_TILE - 1) / RS_TILE); }

size_t pp_sort_pairs_workspace(int64_t n) {
  const size_t m = (size_t)std::max<int64_t>(n, 1);
  return pp_align(m * 8) + pp_align(m * 4) + 2 * pp_align(rs_hist_ints(n) * 4) + pp_scan_workspace((int64_t)rs_hist_ints(n)) + 256;
}

template <typename K>
static int rs_sort(const K* keys_in, K* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n, int end_bit, void* ws,
                   size_t ws_bytes, hipStream_t stream) {
  if (n <= 0) return PP_OK;
  PP_REQUIRE(keys_in && keys_out && vals_in && vals_out, "pp_sort_pairs: null pointer");
  PP_REQUIRE((const void*)keys_in != (const void*)keys_out && vals_in != vals_out, "pp_sort_pairs: in-place sorting is not supported");
  PP_REQUIRE(n < (1ll << 31) && end_bit >= 0 && end_bit <= (int)(8 * sizeof(K)), "pp_sort_pairs: bad size / bit range");
  if (ws_bytes < pp_sort_pairs_workspace(n)) return PP_ERR_WORKSPACE;
  PPArena ar(ws, ws_bytes);
  K* tk = (K*)ar.take<uint64_t>((size_t)n);
  int32_t* tv = ar.take<int32_t>((size_t)n);
  const size_t hi = rs_hist_ints(n);
  int32_t* hist = ar.take<int32_t>(hi);
  int32_t* goff = ar.take<int32_t>(hi);
  if (!tk || !tv || !hist || !goff) return PP_ERR_WORKSPACE;
  const int64_t nb = (n + RS_TILE - 1) / RS_TILE;
  int passes = (std::max(end_bit, 1) + 7) / 8;
  const K* src_k = keys_in;
  const int32_t* src_v = vals_in;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;  // the last pass writes the caller's arrays
    K* dst_k = to_out ? keys_out : tk;
    int32_t* dst_v = to_out ? vals_out : tv;
    const uint32_t dmask = end_bit - 8 * p >= 8 ? 255u : (1u << (end_bit - 8 * p)) - 1u;  // bits at or above end_bit do not count
    hipLaunchKernelGGL((k_rs_hist<K>), dim3((unsigned)nb), dim3(RS_NT), 0, stream, src_k, n, 8 * p, dmask, nb, hist);
    PP_LAUNCH_CHECK();
    int rc = pp_exclusive_scan_i32(hist, goff, (int64_t)hi, nullptr, ar.cur(), ar.left(), stream);
    if (rc) return rc;
    hipLaunchKernelGGL((k_rs_scatter<K>), dim3((unsigned)nb), dim3(RS_NT), 0, stream, src_k, src_v, dst_k, dst_v, n, 8 * p, dmask,
                       nb, goff);
    PP_LAUNCH_CHECK();
    src_k = dst_k;
    src_v = dst_v;
  }
  return PP_OK;
}

int pp_sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                      int64_t n, int end_bit, void* ws, size_t ws_bytes, hipStream_t stream) {
  return rs_sort<uint32_t>(keys_in, keys_out, vals_in, vals_out, n, end_bit, ws, ws_bytes, stream);
}
int pp_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                      int64_t n, int end_bit, void* ws, size_t ws_bytes, hipStream_t stream) {
  return rs_sort<uint64_t>(keys_in, keys_out, vals_in, vals_out, n, end_bit, ws, ws_bytes, stream);
}

// the two primitives through the C ABI (tests, callers outside the library)
extern "C" size_t pp_exclusive_scan_workspace(int64_t n) { return pp_scan_workspace(n); }
extern "C" int pp_exclusive_scan(const int32_t* in, int32_t* out, int64_t n, int32_t* total, void* workspace,
                                 size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE((in && out) || n <= 0, "pp_exclusive_scan: null pointer");
  return pp_exclusive_scan_i32(in, out, n, total, workspace, workspace_bytes, pp_s(stream));
}
extern "C" size_t pp_sort_pairs_workspace_bytes(int64_t n) { return pp_sort_pairs_workspace(n); }
extern "C" int pp_sort_pairs(const void* keys_in, void* keys_out, int32_t key_bytes, const int32_t* vals_in, int32_t* vals_out,
                             int64_t n, int32_t end_bit, void* workspace, size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(key_bytes == 4 || key_bytes == 8, "pp_sort_pairs: keys are uint32 or uint64");
  if (key_bytes == 4)
    return pp_sort_pairs_u32((const uint32_t*)keys_in, (uint32_t*)keys_out, vals_in, vals_out, n, end_bit, workspace,
                             workspace_bytes, pp_s(stream));
  return pp_sort_pairs_u64((const uint64_t*)keys_in, (uint64_t*)keys_out, vals_in, vals_out, n, end_bit, workspace,
                           workspace_bytes, pp_s(stream));
}

// ---------------------------------------------------------------------------------------------------------------------
// Stream compaction and run lengths on the scan above (round 5): what the host layer took from torch.nonzero /
// unique_consecutive / cumsum / repeat_interleave around the mean shift (rocPRIM-backed kernels of the tensor library).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sel_mark(const uint8_t* __restrict__ flags, int64_t n, int32_t* __restrict__ mark) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mark[i] = flags[i] ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_sel_scatter(const uint8_t* __restrict__ flags, const int32_t* __restrict__ pos, int64_t n,
                                                     int64_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) idx[pos[i]] = i;
}
__global__ __launch_bounds__(256) void k_run_mark(const int64_t* __restrict__ v, int64_t n, int32_t* __restrict__ mark) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mark[i] = (i == 0 || v[i] != v[i - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_run_scatter(const int64_t* __restrict__ v, const int32_t* __restrict__ mark,
                                                     const int32_t* __restrict__ pos, int64_t n, int32_t* __restrict__ run_id,
                                                     int64_t* __restrict__ heads, int32_t* __restrict__ starts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = pos[i] + mark[i] - 1;  // 0-based run of element i
  if (run_id) run_id[i] = r;
  if (mark[i]) {
    heads[r] = v[i];
    starts[r] = (int32_t)i;
  }
  if (i == n - 1) starts[r + 1] = (int32_t)n;
}

extern "C" size_t pp_select_workspace(int64_t n) { return pp_align((size_t)(n > 0 ? n : 1) * 8) + pp_scan_workspace(n) + 256; }

extern "C" int pp_select_indices(const uint8_t* flags, int64_t n, int64_t* idx, int32_t* count, void* workspace, size_t workspace_bytes,
                                 pp_stream_t stream) {
  PP_REQUIRE(count && workspace, "pp_select_indices: null pointer");
  PP_REQUIRE(n < (int64_t(1) << 31), "pp_select_indices: n must be below 2^31");
  PP_REQUIRE(workspace_bytes >= pp_select_workspace(n), "pp_select_indices: workspace too small");
  hipStream_t s = pp_s(stream);
  if (n <= 0) {
    PP_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    return PP_OK;
  }
  PP_REQUIRE(flags && idx, "pp_select_indices: null pointer");
  PPArena ar(workspace, workspace_bytes);
  int32_t* mark = ar.take<int32_t>((size_t)n);
  int32_t* pos = ar.take<int32_t>((size_t)n);
  hipLaunchKernelGGL(k_sel_mark, dim3(pp_blocks(n, 256)), dim3(256), 0, s, flags, n, mark);
  int rc = pp_exclusive_scan_i32(mark, pos, n, count, ar.cur(), ar.left(), s);
  if (rc != PP_OK) return rc;
  hipLaunchKernelGGL(k_sel_scatter, dim3(pp_blocks(n, 256)), dim3(256), 0, s, flags, pos, n, idx);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_run_lengths(const int64_t* values, int64_t n, int32_t* run_id, int64_t* heads, int32_t* starts, int32_t* n_runs,
                              void* workspace, size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(n_runs && workspace, "pp_run_lengths: null pointer");
  PP_REQUIRE(n < (int64_t(1) << 31), "pp_run_lengths: n must be below 2^31");
  PP_REQUIRE(workspace_bytes >= pp_select_workspace(n), "pp_run_lengths: workspace too small");
  hipStream_t s = pp_s(stream);
  if (n <= 0) {
    PP_HIP(hipMemsetAsync(n_runs, 0, sizeof(int32_t), s));
    if (starts) PP_HIP(hipMemsetAsync(starts, 0, sizeof(int32_t), s));
    return PP_OK;
  }
  PP_REQUIRE(values && heads && starts, "pp_run_lengths: null pointer");
  PPArena ar(workspace, workspace_bytes);
  int32_t* mark = ar.take<int32_t>((size_t)n);
  int32_t* pos = ar.take<int32_t>((size_t)n);
  hipLaunchKernelGGL(k_run_mark, dim3(pp_blocks(n, 256)), dim3(256), 0, s, values, n, mark);
  int rc = pp_exclusive_scan_i32(mark, pos, n, n_runs, ar.cur(), ar.left(), s);
  if (rc != PP_OK) return rc;
  hipLaunchKernelGGL(k_run_scatter, dim3(pp_blocks(n, 256)), dim3(256), 0, s, values, mark, pos, n, run_id, heads, starts);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
