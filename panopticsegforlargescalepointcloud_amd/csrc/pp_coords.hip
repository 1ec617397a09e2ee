// K1-K3: coordinate hash, strided output coordinates, kernel maps.  Pure integer, HBM/L2 bound.
// Semantics follow SURVEY.md 8c / oracle/panoptic_oracle.c (ppo_hash_first_rows, ppo_stride_coords,
// ppo_kernel_map); reference call sites: torch_points3d/applications/minkowski.py:121-122,
// torch_points3d/modules/MinkowskiEngine/api_modules.py:30-51,256-271.
#include "pp_common.h"

extern "C" int64_t pp_hash_capacity(int64_t n) {
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

__global__ __launch_bounds__(256) void k_hash_fill(uint64_t* keys, int32_t* vals, int64_t cap) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) {
    keys[i] = PP_EMPTY_KEY;
    vals[i] = 0x7FFFFFFF;
  }
}

// insert key of (optionally quantised) row i with value min(row index); slot_of_row optional
__global__ __launch_bounds__(256) void k_hash_insert_min(const int4* __restrict__ coords, int64_t n, int ts,
                                                         uint64_t* keys, int32_t* vals, int64_t cap,
                                                         int32_t* slot_of_row, int32_t* info) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  if (ts > 1) {
    c.y = pp_floor_div(c.y, ts) * ts;
    c.z = pp_floor_div(c.z, ts) * ts;
    c.w = pp_floor_div(c.w, ts) * ts;
  }
  if (!pp_key_ok(c.x, c.y, c.z, c.w)) {
    atomicAdd(&info[1], 1);
    if (slot_of_row) slot_of_row[i] = -1;
    return;
  }
  int64_t s = pp_hash_insert_slot(keys, cap, pp_key_pack(c.x, c.y, c.z, c.w));
  atomicMin(&vals[s], (int32_t)i);
  if (slot_of_row) slot_of_row[i] = (int32_t)s;
}

__global__ __launch_bounds__(256) void k_count_dups(const int4* __restrict__ coords, int64_t n,
                                                    const uint64_t* __restrict__ keys,
                                                    const int32_t* __restrict__ vals, int64_t cap, int32_t* info) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  if (!pp_key_ok(c.x, c.y, c.z, c.w)) return;
  int64_t s = pp_hash_find_slot(keys, cap, pp_key_pack(c.x, c.y, c.z, c.w));
  if (s >= 0 && vals[s] != (int32_t)i) atomicAdd(&info[0], 1);
}

extern "C" int pp_hash_build(const int32_t* coords, int64_t n, uint64_t* keys, int32_t* vals, int64_t cap,
                             int32_t* info, pp_stream_t stream) {
  PP_REQUIRE(keys && vals && info, "pp_hash_build: null table");
  PP_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0, "pp_hash_build: cap must be a power of two >= 2n");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(info, 0, 2 * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_hash_fill, dim3((unsigned)std::min<int64_t>((cap + 255) / 256, 4096)), dim3(256), 0, s, keys, vals,
                     cap);
  PP_LAUNCH_CHECK();
  if (n > 0) {
    hipLaunchKernelGGL(k_hash_insert_min, dim3(pp_blocks(n, 256)), dim3(256), 0, s, (const int4*)coords, n, 1, keys,
                       vals, cap, (int32_t*)nullptr, info);
    PP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_count_dups, dim3(pp_blocks(n, 256)), dim3(256), 0, s, (const int4*)coords, n, keys, vals, cap,
                       info);
    PP_LAUNCH_CHECK();
  }
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// K2: strided coordinates in first-appearance order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mark_leaders(int64_t n, const int32_t* __restrict__ slot_of_row,
                                                      const int32_t* __restrict__ vals, int32_t* lead_row,
                                                      int32_t* flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t s = slot_of_row[i];
  int32_t lr = s >= 0 ? vals[s] : -1;
  lead_row[i] = lr;
  flag[i] = (lr == (int32_t)i) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_emit_coarse(const int4* __restrict__ coords, int64_t n, int ts,
                                                     const int32_t* __restrict__ slot_of_row,
                                                     const int32_t* __restrict__ lead_row,
                                                     const int32_t* __restrict__ rank, int32_t* vals,
                                                     int4* out_coords, int32_t* fine_to_coarse) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t lr = lead_row[i];
  if (lr < 0) {
    if (fine_to_coarse) fine_to_coarse[i] = -1;
    return;
  }
  int32_t r = rank[lr];
  if (fine_to_coarse) fine_to_coarse[i] = r;
  if (lr == (int32_t)i) {
    int4 c = coords[i];
    c.y = pp_floor_div(c.y, ts) * ts;
    c.z = pp_floor_div(c.z, ts) * ts;
    c.w = pp_floor_div(c.w, ts) * ts;
    out_coords[r] = c;
    vals[slot_of_row[i]] = r;  // table now maps coarse key -> coarse row
  }
}

extern "C" size_t pp_stride_coords_workspace(int64_t n) {
  return 4 * pp_align((size_t)std::max<int64_t>(n, 1) * sizeof(int32_t)) + pp_scan_workspace(n) + 1024;
}

extern "C" int pp_stride_coords(const int32_t* coords, int64_t n, int32_t ts_out, uint64_t* keys, int32_t* vals,
                                int64_t cap, int32_t* out_coords, int32_t* n_out, int32_t* fine_to_coarse,
                                void* workspace, size_t workspace_bytes, int32_t* info, pp_stream_t stream) {
  PP_REQUIRE(ts_out >= 1, "pp_stride_coords: ts_out must be >= 1");
  PP_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0, "pp_stride_coords: cap must be a power of two >= 2n");
  if (workspace_bytes < pp_stride_coords_workspace(n)) return PP_ERR_WORKSPACE;
  hipStream_t s = pp_s(stream);
  PPArena ar(workspace, workspace_bytes);
  int32_t* slot_of_row = ar.take<int32_t>((size_t)std::max<int64_t>(n, 1));
  int32_t* lead_row = ar.take<int32_t>((size_t)std::max<int64_t>(n, 1));
  int32_t* flag = ar.take<int32_t>((size_t)std::max<int64_t>(n, 1));
  int32_t* rank = ar.take<int32_t>((size_t)std::max<int64_t>(n, 1));
  PP_HIP(hipMemsetAsync(info, 0, 2 * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_hash_fill, dim3((unsigned)std::min<int64_t>((cap + 255) / 256, 4096)), dim3(256), 0, s, keys, vals,
                     cap);
  PP_LAUNCH_CHECK();
  if (n == 0) {
    PP_HIP(hipMemsetAsync(n_out, 0, sizeof(int32_t), s));
    return PP_OK;
  }
  unsigned nb = pp_blocks(n, 256);
  hipLaunchKernelGGL(k_hash_insert_min, dim3(nb), dim3(256), 0, s, (const int4*)coords, n, ts_out, keys, vals, cap,
                     slot_of_row, info);
  PP_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_mark_leaders, dim3(nb), dim3(256), 0, s, n, slot_of_row, vals, lead_row, flag);
  PP_LAUNCH_CHECK();
  int rc = pp_exclusive_scan_i32(flag, rank, n, n_out, ar.cur(), ar.left(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_emit_coarse, dim3(nb), dim3(256), 0, s, (const int4*)coords, n, ts_out, slot_of_row, lead_row,
                     rank, vals, (int4*)out_coords, fine_to_coarse);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// K3: kernel map.  One thread per output row, 27 independent probes (ILP), offset-major coalesced stores.
// ---------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void k_kernel_map(const int4* __restrict__ out_coords, int64_t n_out,
                                                    const uint64_t* __restrict__ keys,
                                                    const int32_t* __restrict__ vals, int64_t cap, int step,
                                                    int32_t* __restrict__ nbr, unsigned long long* n_pairs) {
  // grid-stride rows: the pair count costs one atomic per wave of the (bounded) grid, not one per 64 rows -- atomics on
  // one address serialise in L2 (see k_kernel_map_bi)
  int found = 0;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += (int64_t)gridDim.x * blockDim.x) {
  int4 c = out_coords[o];
  constexpr int K = KS * KS * KS;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    int dx = 0, dy = 0, dz = 0;
    if (KS == 3) {
      dx = k % 3 - 1;
      dy = (k / 3) % 3 - 1;
      dz = k / 9 - 1;
    }
    int x = c.y + dx * step, y = c.z + dy * step, z = c.w + dz * step;
    int32_t r = -1;
    if (pp_key_ok(c.x, x, y, z)) {
      int64_t s = pp_hash_find_slot(keys, cap, pp_key_pack(c.x, x, y, z));
      if (s >= 0) r = vals[s];
    }
    found += r >= 0 ? 1 : 0;
    nbr[(int64_t)k * n_out + o] = r;
  }
  }
  if (n_pairs) {  // number of (in, out) pairs of the map: one atomic per wave
    for (int off = 32; off > 0; off >>= 1) found += __shfl_xor(found, off);
    if ((threadIdx.x & 63) == 0 && found) atomicAdd(n_pairs, (unsigned long long)found);
  }
}

extern "C" int pp_kernel_map(const int32_t* out_coords, int64_t n_out, const uint64_t* keys, const int32_t* vals,
                             int64_t cap, int32_t ksize, int32_t step, int32_t sign, int32_t* nbr,
                             int64_t* n_pairs, pp_stream_t stream) {
  PP_REQUIRE(ksize == 1 || ksize == 3, "pp_kernel_map: ksize must be 1 or 3");
  PP_REQUIRE(sign == 1 || sign == -1, "pp_kernel_map: sign must be +1 or -1");
  hipStream_t s = pp_s(stream);
  if (n_pairs) PP_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int64_t), s));
  if (n_out == 0) return PP_OK;
  unsigned nb = (unsigned)std::min<int64_t>(pp_blocks(n_out, 256), 256 * 16);
  if (ksize == 3)
    hipLaunchKernelGGL(k_kernel_map<3>, dim3(nb), dim3(256), 0, s, (const int4*)out_coords, n_out, keys, vals, cap,
                       sign * step, nbr, (unsigned long long*)n_pairs);
  else
    hipLaunchKernelGGL(k_kernel_map<1>, dim3(nb), dim3(256), 0, s, (const int4*)out_coords, n_out, keys, vals, cap, 0,
                       nbr, (unsigned long long*)n_pairs);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// kernel-map transpose: out[k][in_map[k][o]] = o.  Turns the strided (fine -> coarse) map into the transposed
// convolution's (coarse -> fine) map with P scattered writes instead of 27 hash probes per fine row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kernel_map_transpose(const int32_t* __restrict__ in_map, int64_t n_out, int K,
                                                              int64_t n_in, const int32_t* __restrict__ in_order,
                                                              int32_t* __restrict__ out_map) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)K * n_out) return;
  int32_t i = in_map[e];
  if (i >= 0) {
    int64_t k = e / n_out, o = e - k * n_out;
    out_map[k * n_in + i] = in_order ? in_order[o] : (int32_t)o;  // slot-ordered input map: slot o is row in_order[o]
  }
}
extern "C" int pp_kernel_map_transpose(const int32_t* in_map, int64_t n_out, int32_t K, int64_t n_in, const int32_t* in_order,
                                       int32_t* out_map, pp_stream_t stream) {
  PP_REQUIRE(in_map && out_map && K >= 1, "pp_kernel_map_transpose: bad arguments");
  hipStream_t s = pp_s(stream);
  if (n_in > 0) PP_HIP(hipMemsetAsync(out_map, 0xFF, sizeof(int32_t) * (size_t)K * (size_t)n_in, s));
  if (n_out == 0 || n_in == 0) return PP_OK;
  hipLaunchKernelGGL(k_kernel_map_transpose, dim3(pp_blocks((int64_t)K * n_out, 256)), dim3(256), 0, s, in_map, n_out, K,
                     n_in, in_order, out_map);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// 8-wide map of a transposed stride-2 convolution ("T8").  A fine row f = c + offset_k * t has a coarse neighbour only
// through offsets whose components are 0 on the axes where f is even and +-1 where it is odd (in units of t): at most
// 2 x 2 x 2 of the 27, fixed by the row's parity class cls = (x odd) | (y odd) << 1 | (z odd) << 2.  Instead of a dense
// [27][n_fine] map (108 B per row, four passes over it: scatter, mask, permute read + write) the map is
//   map8[j][f] = c | cls << 28,  j = (dx > 0) | (dy > 0) << 1 | (dz > 0) << 2     (32 B per row; -1 = none)
// -- every entry of a row carries the row's class in bits 28..30 (a separate class array cost a scattered BYTE write per
// pair: the scatter took 737 instead of 627 us per 10 M rows); the convolution kernel expands (cls, j) back to the offset
// index k in its prologue.
//   k_kernel_map_transpose8: map8[j(k)][in_map[k][c]] = c | class(k) << 28
//   k_map8_key            : key[f] = cls << 8 | presence bits of the 8 entries          (what the slot order sorts by)
// Reference: none (MinkowskiEngine keeps (in, out) pair lists per offset); same pairs as pp_kernel_map_transpose.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kernel_map_transpose8(const int32_t* __restrict__ in_map, int64_t n_out, int64_t n_in,
                                                               const int32_t* __restrict__ in_order, int32_t* __restrict__ map8) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 27 * n_out) return;
  const int32_t f = in_map[e];
  if (f >= 0) {
    const int k = (int)(e / n_out);
    const int64_t o = e - (int64_t)k * n_out;
    const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
    const int j = (dx > 0 ? 1 : 0) | (dy > 0 ? 2 : 0) | (dz > 0 ? 4 : 0);
    const uint32_t cls = (dx != 0 ? 1u : 0u) | (dy != 0 ? 2u : 0u) | (dz != 0 ? 4u : 0u);
    const uint32_t c = (uint32_t)(in_order ? in_order[o] : (int32_t)o);
    map8[(int64_t)j * n_in + f] = (int32_t)(c | (cls << 28));
  }
}
__global__ __launch_bounds__(256) void k_map8_key(const int32_t* __restrict__ map8, int64_t n, uint32_t* __restrict__ key) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  uint32_t m = 0, cls = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int32_t v = map8[(int64_t)j * n + f];
    if (v >= 0) {
      m |= 1u << j;
      cls |= (uint32_t)v >> 28;
    }
  }
  key[f] = m | (cls << 8);
}
extern "C" int pp_kernel_map_transpose8(const int32_t* in_map, int64_t n_out, int64_t n_in, const int32_t* in_order,
                                        int32_t* map8, uint32_t* key, pp_stream_t stream) {
  PP_REQUIRE(in_map && map8, "pp_kernel_map_transpose8: null pointer");
  PP_REQUIRE(n_out < (1ll << 28), "pp_kernel_map_transpose8: more than 2^28 coarse rows (the class shares a word with the row)");
  hipStream_t s = pp_s(stream);
  if (n_in > 0) PP_HIP(hipMemsetAsync(map8, 0xFF, sizeof(int32_t) * 8 * (size_t)n_in, s));
  if (n_in == 0) return PP_OK;
  if (n_out > 0)
    hipLaunchKernelGGL(k_kernel_map_transpose8, dim3(pp_blocks(27 * n_out, 256)), dim3(256), 0, s, in_map, n_out, n_in, in_order,
                       map8);
  if (key) hipLaunchKernelGGL(k_map8_key, dim3(pp_blocks(n_in, 256)), dim3(256), 0, s, map8, n_in, key);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// Morton (Z-order) row permutation: perm[p] = row with the p-th smallest (batch, interleave(x,y,z)) key.
// Internal row order of the coordinate manager: 16 consecutive rows form a compact surface patch (tile-level
// offset skipping in the convolution) and gathered neighbours stay L2-resident.
// ---------------------------------------------------------------------------------------------
__device__ inline uint64_t pp_spread3(uint64_t x) {
  x &= 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__global__ __launch_bounds__(256) void k_morton_keys(const int4* __restrict__ coords, int64_t n, uint64_t* key,
                                                     int32_t* idx, int32_t* info, int unit_shift, int block_bits) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  if (!pp_key_ok(c.x, c.y, c.z, c.w)) {
    atomicAdd(&info[1], 1);
    key[i] = ~0ull;
  } else {
    key[i] = pp_order_key(c.x, c.y, c.z, c.w, unit_shift, block_bits);
  }
  idx[i] = (int32_t)i;
}
// rows in sorted order straight from the sorted keys (the key is a bijection of the coordinate): no gather needed
__global__ __launch_bounds__(256) void k_morton_decode(const uint64_t* __restrict__ skey, int64_t n, int unit_shift,
                                                       int block_bits, int4* out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = skey[i];
  const uint64_t body = k & 0xFFFFFFFFFFFFull;
  out[i] = make_int4((int)(k >> 48), (int)(pp_order_axis_inv(body, 0, block_bits) << unit_shift) - 32768,
                     (int)(pp_order_axis_inv(body, 1, block_bits) << unit_shift) - 32768,
                     (int)(pp_order_axis_inv(body, 2, block_bits) << unit_shift) - 32768);
}
extern "C" size_t pp_morton_order_workspace(int64_t n) {
  size_t m = (size_t)std::max<int64_t>(n, 1);
  return 2 * pp_align(m * 8) + pp_align(m * 4) + pp_sort_pairs_workspace(n) + 1024;
}
extern "C" int pp_morton_order(const int32_t* coords, int64_t n, int32_t unit, int32_t block_bits, int32_t* perm,
                               int32_t* sorted_coords, void* workspace, size_t workspace_bytes, int32_t* info,
                               pp_stream_t stream) {
  PP_REQUIRE(perm && info, "pp_morton_order: null output");
  PP_REQUIRE(unit >= 1 && (unit & (unit - 1)) == 0 && unit <= 16384, "pp_morton_order: unit must be a power of two");
  PP_REQUIRE(block_bits >= 0 && block_bits <= 8, "pp_morton_order: block_bits in [0,8]");
  int unit_shift = 0;
  while ((1 << unit_shift) < unit) ++unit_shift;
  if (workspace_bytes < pp_morton_order_workspace(n)) return PP_ERR_WORKSPACE;
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(info, 0, 2 * sizeof(int32_t), s));
  if (n == 0) return PP_OK;
  PPArena ar(workspace, workspace_bytes);
  size_t m = (size_t)n;
  uint64_t* key = ar.take<uint64_t>(m);
  uint64_t* key2 = ar.take<uint64_t>(m);
  int32_t* idx = ar.take<int32_t>(m);
  hipLaunchKernelGGL(k_morton_keys, dim3(pp_blocks(n, 256)), dim3(256), 0, s, (const int4*)coords, n, key, idx, info,
                     unit_shift, block_bits);
  PP_LAUNCH_CHECK();
  int rc = pp_sort_pairs_u64(key, key2, idx, perm, n, 64, ar.cur(), ar.left(), s);
  if (rc) return rc;
  if (sorted_coords) {  // only meaningful when info[1] == 0 (every row inside the key range)
    hipLaunchKernelGGL(k_morton_decode, dim3(pp_blocks(n, 256)), dim3(256), 0, s, key2, n, unit_shift, block_bits,
                       (int4*)sorted_coords);
    PP_LAUNCH_CHECK();
  }
  return PP_OK;
}
