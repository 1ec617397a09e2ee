// f1: voxelisation and cylinder cutting on the GPU (the step right before the hot path).
//
// pp_voxelize        GridSampling3D(size, quantize_coords=True), torch_points3d/core/data_transform/grid_transform.py:181-198:
//                    coords = round(pos / size) (round-half-even on the float32 quotient, like torch.round), one row per
//                    occupied voxel, rows ordered like torch_geometric's voxel_grid + consecutive_cluster: batch slowest,
//                    then z, y, x; the representative of a voxel is its LAST point in input order (what
//                    consecutive_cluster's scatter leaves on the CPU; the reference shuffles first, so "random").
// pp_cylinder_pairs  CylinderSampling, torch_points3d/core/data_transform/transforms.py:388-441: point i belongs to
//                    cylinder c when its horizontal distance to the centre is <= radius (KDTree.query_radius is
//                    inclusive).  Emits (point, cylinder) pairs, point-major; pp_group_by_key turns them into one
//                    ascending index list per cylinder.
#include "pp_common.h"

__global__ __launch_bounds__(256) void k_vox_keys(const float* __restrict__ pos, const int64_t* __restrict__ batch, int64_t n,
                                                  float size, unsigned long long* key, int32_t* idx, int32_t* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)rintf(pos[3 * i] / size), y = (int)rintf(pos[3 * i + 1] / size), z = (int)rintf(pos[3 * i + 2] / size);
  const int64_t b = batch ? batch[i] : 0;
  if (!pp_key_ok((int)b, x, y, z) || b < 0 || b > 65535) {
    atomicAdd(err, 1);
    key[i] = ~0ull;
  } else {
    key[i] = ((unsigned long long)(uint16_t)b << 48) | ((unsigned long long)(uint16_t)(z + 32768) << 32) |
             ((unsigned long long)(uint16_t)(y + 32768) << 16) | (unsigned long long)(uint16_t)(x + 32768);
  }
  idx[i] = (int32_t)i;
}
__global__ __launch_bounds__(256) void k_vox_flags(const unsigned long long* __restrict__ skey, int64_t n, int32_t* flag) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) flag[p] = (p == 0 || skey[p] != skey[p - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_vox_emit(const unsigned long long* __restrict__ skey, const int32_t* __restrict__ sidx,
                                                  const int32_t* __restrict__ flag, const int32_t* __restrict__ excl,
                                                  int64_t n, int4* coords, int32_t* rep, int32_t* inverse) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int v = excl[p] + flag[p] - 1;  // voxel row of sorted position p
  inverse[sidx[p]] = v;
  if (p == n - 1 || skey[p + 1] != skey[p]) {  // last of its run = largest input index (the sort is stable)
    const unsigned long long k = skey[p];
    rep[v] = sidx[p];
    coords[v] = make_int4((int)(k >> 48), (int)(k & 0xFFFF) - 32768, (int)((k >> 16) & 0xFFFF) - 32768,
                          (int)((k >> 32) & 0xFFFF) - 32768);
  }
}

extern "C" size_t pp_voxelize_workspace(int64_t n) {
  const size_t m = (size_t)std::max<int64_t>(n, 1);
  return 2 * pp_align(m * 8) + 4 * pp_align(m * 4) + pp_sort_pairs_workspace(n) + pp_scan_workspace(n) + 4096;
}
extern "C" int pp_voxelize(const float* pos, const int64_t* batch, int64_t n, float voxel_size, int32_t* coords,
                           int32_t* rep_index, int32_t* inverse, int32_t* counts, void* workspace,
                           size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(coords && rep_index && inverse && counts, "pp_voxelize: null output");
  PP_REQUIRE(voxel_size > 0.f, "pp_voxelize: voxel_size must be positive");
  PP_REQUIRE(n < (1ll << 31), "pp_voxelize: too many points");
  if (workspace_bytes < pp_voxelize_workspace(n)) return PP_ERR_WORKSPACE;
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), s));  // [0] voxels, [1] points outside the key range
  if (n == 0) return PP_OK;
  PPArena ar(workspace, workspace_bytes);
  uint64_t* key = ar.take<uint64_t>((size_t)n);
  uint64_t* skey = ar.take<uint64_t>((size_t)n);
  int32_t* idx = ar.take<int32_t>((size_t)n);
  int32_t* sidx = ar.take<int32_t>((size_t)n);
  int32_t* flag = ar.take<int32_t>((size_t)n);
  int32_t* excl = ar.take<int32_t>((size_t)n);
  const unsigned nb = pp_blocks(n, 256);
  hipLaunchKernelGGL(k_vox_keys, dim3(nb), dim3(256), 0, s, pos, batch, n, voxel_size, (unsigned long long*)key, idx, counts + 1);
  PP_LAUNCH_CHECK();
  int rc = pp_sort_pairs_u64(key, skey, idx, sidx, n, 64, ar.cur(), ar.left(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_vox_flags, dim3(nb), dim3(256), 0, s, (const unsigned long long*)skey, n, flag);
  PP_LAUNCH_CHECK();
  rc = pp_exclusive_scan_i32(flag, excl, n, counts, ar.cur(), ar.left(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_vox_emit, dim3(nb), dim3(256), 0, s, (const unsigned long long*)skey, sidx, flag, excl, n, (int4*)coords,
                     rep_index, inverse);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- cylinders -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cyl_count(const float* __restrict__ pos, int64_t n, const float* __restrict__ centres,
                                                   int n_cyl, float r2, int32_t* cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = pos[3 * i], y = pos[3 * i + 1];
  int c = 0;
  for (int k = 0; k < n_cyl; ++k) {
    const float dx = x - centres[2 * k], dy = y - centres[2 * k + 1];
    c += (fmaf(dy, dy, dx * dx) <= r2) ? 1 : 0;
  }
  cnt[i] = c;
}
__global__ __launch_bounds__(256) void k_cyl_fill(const float* __restrict__ pos, int64_t n, const float* __restrict__ centres,
                                                  int n_cyl, float r2, const int32_t* __restrict__ off, int64_t cap,
                                                  int64_t* pair_point, int32_t* pair_cyl) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = pos[3 * i], y = pos[3 * i + 1];
  int64_t w = off[i];
  for (int k = 0; k < n_cyl; ++k) {
    const float dx = x - centres[2 * k], dy = y - centres[2 * k + 1];
    if (fmaf(dy, dy, dx * dx) <= r2) {
      if (w < cap) {
        pair_point[w] = i;
        pair_cyl[w] = k;
      }
      ++w;
    }
  }
}
extern "C" size_t pp_cylinder_pairs_workspace(int64_t n) {
  const size_t m = (size_t)std::max<int64_t>(n, 1);
  return 2 * pp_align(m * 4) + pp_scan_workspace(n) + 1024;
}
extern "C" int pp_cylinder_pairs(const float* pos, int64_t n, const float* centres_xy, int32_t n_cyl, float radius,
                                 int64_t* pair_point, int32_t* pair_cyl, int64_t capacity, int32_t* n_pairs,
                                 void* workspace, size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(n_pairs && (n_cyl == 0 || centres_xy), "pp_cylinder_pairs: null pointer");
  PP_REQUIRE(radius > 0.f && n_cyl >= 0 && n < (1ll << 31), "pp_cylinder_pairs: bad parameters");
  if (workspace_bytes < pp_cylinder_pairs_workspace(n)) return PP_ERR_WORKSPACE;
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int32_t), s));
  if (n == 0 || n_cyl == 0) return PP_OK;
  PPArena ar(workspace, workspace_bytes);
  int32_t* cnt = ar.take<int32_t>((size_t)n);
  int32_t* off = ar.take<int32_t>((size_t)n);
  const unsigned nb = pp_blocks(n, 256);
  hipLaunchKernelGGL(k_cyl_count, dim3(nb), dim3(256), 0, s, pos, n, centres_xy, n_cyl, radius * radius, cnt);
  PP_LAUNCH_CHECK();
  int rc = pp_exclusive_scan_i32(cnt, off, n, n_pairs, ar.cur(), ar.left(), s);
  if (rc) return rc;
  if (pair_point && pair_cyl && capacity > 0) {  // call with NULL outputs first to size them
    hipLaunchKernelGGL(k_cyl_fill, dim3(nb), dim3(256), 0, s, pos, n, centres_xy, n_cyl, radius * radius, off, capacity,
                       pair_point, pair_cyl);
    PP_LAUNCH_CHECK();
  }
  return PP_OK;
}
