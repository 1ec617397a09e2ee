// f2  exact 1-nearest-neighbour of every query point among a set of reference points (2-D or 3-D).
// replaces: torch_geometric knn(x=ref, y=query, k=1) / knn_interpolate(k=1) of the full-resolution back-projection,
// torch_points3d/metrics/panoptic_tracker_pointgroup_npm3d.py:564-566,593-618, and the KD-tree query of the cylinder
// centre label, torch_points3d/core/data_transform/transforms.py:239-240.
//
// Reference points are bucketed into a uniform grid (cell edge `cell`) and radix-sorted by the Morton code of their
// cell, so the points of a cell -- and of every aligned 4^l x 4^l x 4^l block of cells -- are contiguous.  A query scans
// the 3x3x3 cells around its own cell (hash: cell -> [start, end)); every reference point outside that block is
// farther than `cell`, so the search is over when the best distance is within that bound -- almost every point of a
// cloud that was sub-sampled from the queries.  Otherwise the same test is repeated one level up (cells 4x larger,
// ranges found by binary search in the sorted keys) until the bound holds, max_dist is exceeded, or one cell spans the
// whole key space.  HBM-bound integer/float work, one thread per query.
//
// Distance = ((dx*dx + dy*dy) + dz*dz) in float32 without fused multiply-add; ties -> smallest reference index.
#include "pp_common.h"

#define NN_OFF 32768

struct NNGrid {
  const uint64_t* keys;     // hash keys (cell keys), PP_EMPTY_KEY = free
  const int32_t* cstart;    // per slot
  const int32_t* cend;
  int64_t cap;
  const float4* spos;       // reference points in cell order, w = original index bits
  float cell;
  int dim;
};

// Morton code of non-negative (offset) cell coordinates, 16 bits per axis
__device__ __forceinline__ uint64_t nn_morton(unsigned x, unsigned y, unsigned z) {
  return pp_spread3_64(x) | (pp_spread3_64(y) << 1) | (pp_spread3_64(z) << 2);
}
__device__ __forceinline__ uint64_t nn_key(int cx, int cy, int cz) {
  return nn_morton((unsigned)(cx + NN_OFF), (unsigned)(cy + NN_OFF), (unsigned)(cz + NN_OFF));
}

__global__ __launch_bounds__(256) void k_nn_keys(const float* __restrict__ ref, int64_t n, int dim, float cell,
                                                 uint64_t* __restrict__ key, int32_t* __restrict__ val,
                                                 int32_t* __restrict__ err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = ref[i * dim], y = ref[i * dim + 1], z = dim == 3 ? ref[i * dim + 2] : 0.f;
  const float fx = floorf(x / cell), fy = floorf(y / cell), fz = floorf(z / cell);
  val[i] = (int32_t)i;
  if (!(fabsf(fx) < 32000.f && fabsf(fy) < 32000.f && fabsf(fz) < 32000.f)) {  // also catches NaN / inf
    atomicAdd(err, 1);
    key[i] = 0;
    return;
  }
  const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
  key[i] = nn_key(cx, cy, cz);
}

__global__ __launch_bounds__(256) void k_nn_fill(uint64_t* keys, int64_t cap) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) keys[i] = PP_EMPTY_KEY;
}

// one hash entry per run of equal keys in the sorted order; gathers the points into cell order
__global__ __launch_bounds__(256) void k_nn_cells(const uint64_t* __restrict__ skey, const int32_t* __restrict__ sval,
                                                  int64_t n, const float* __restrict__ ref, int dim, uint64_t* keys,
                                                  int32_t* cstart, int32_t* cend, int64_t cap, float4* spos) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int32_t i = sval[p];
  spos[p] = make_float4(ref[(int64_t)i * dim], ref[(int64_t)i * dim + 1], dim == 3 ? ref[(int64_t)i * dim + 2] : 0.f,
                        __int_as_float(i));
  const uint64_t k = skey[p];
  const bool first = p == 0 || skey[p - 1] != k;
  const bool last = p == n - 1 || skey[p + 1] != k;
  if (first || last) {
    const int64_t s = pp_hash_insert_slot(keys, cap, k);
    if (first) cstart[s] = (int32_t)p;
    if (last) cend[s] = (int32_t)(p + 1);
  }
}

__device__ __forceinline__ void nn_scan_range(const float4* __restrict__ spos, int p0, int p1, float qx, float qy,
                                              float qz, float& best, int& bidx) {
#pragma clang fp contract(off)  // hipcc contracts a*b + c into an FMA by default (also through __fmul_rn / __fadd_rn)
  for (int p = p0; p < p1; ++p) {
    const float4 r = spos[p];
    const float dx = qx - r.x, dy = qy - r.y, dz = qz - r.z;
    const float d = (dx * dx + dy * dy) + dz * dz;  // plain operators: the pragma acts on this scope only
    const int i = __float_as_int(r.w);
    if (d < best || (d == best && i < bidx)) {
      best = d;
      bidx = i;
    }
  }
}

__device__ __forceinline__ int nn_lower_bound(const uint64_t* __restrict__ k, int n, uint64_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (k[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_nn_query(NNGrid g, const uint64_t* __restrict__ skey, int n_ref,
                                                  const float* __restrict__ query, int64_t n, float max_dist,
                                                  int64_t* __restrict__ idx, float* __restrict__ dist2) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int dim = g.dim;
  const float qx = query[t * dim], qy = query[t * dim + 1], qz = dim == 3 ? query[t * dim + 2] : 0.f;
  // own cell in offset coordinates, clamped into the key range: for a query outside the grid the points outside the
  // 3^3 block of the border cell are still farther than one cell edge (the clamped axis only increases the distance)
  const float hi_c = 65535.f;
  const int cx = (int)fminf(fmaxf(floorf(qx / g.cell) + (float)NN_OFF, 0.f), hi_c);
  const int cy = (int)fminf(fmaxf(floorf(qy / g.cell) + (float)NN_OFF, 0.f), hi_c);
  const int cz = (int)fminf(fmaxf(floorf(qz / g.cell) + (float)NN_OFF, 0.f), hi_c);
  const float md2 = max_dist > 0.f ? max_dist * max_dist : INFINITY;
  float best = INFINITY;
  int bidx = -1;
  float edge = g.cell;
  for (int lvl = 0; lvl <= 8; ++lvl, edge *= 4.f) {
    const int sh = 2 * lvl, top = (65536 >> sh) - 1;  // coarse coordinates in [0, top]
    const int X = cx >> sh, Y = cy >> sh, Z = cz >> sh;
    const int z0 = dim == 3 ? max(Z - 1, 0) : Z, z1 = dim == 3 ? min(Z + 1, top) : Z;
    for (int z = z0; z <= z1; ++z)
      for (int y = max(Y - 1, 0); y <= min(Y + 1, top); ++y)
        for (int x = max(X - 1, 0); x <= min(X + 1, top); ++x) {
          const uint64_t m = nn_morton((unsigned)x, (unsigned)y, (unsigned)z);
          if (lvl == 0) {
            const int64_t s = pp_hash_find_slot(g.keys, g.cap, m);
            if (s >= 0) nn_scan_range(g.spos, g.cstart[s], g.cend[s], qx, qy, qz, best, bidx);
          } else {
            const int p0 = nn_lower_bound(skey, n_ref, m << (3 * sh));
            const int p1 = lvl == 8 ? n_ref : nn_lower_bound(skey, n_ref, (m + 1) << (3 * sh));
            nn_scan_range(g.spos, p0, p1, qx, qy, qz, best, bidx);
          }
        }
    // everything outside the block is farther than one cell edge of this level (margin: rounding of floor(p / cell))
    const float bound = edge * (1.f - 1e-3f);
    if (best <= bound * bound) break;
    if (bound * bound >= md2) break;  // nothing closer than max_dist can remain
  }
  if (best > md2) {
    best = INFINITY;
    bidx = -1;
  }
  idx[t] = bidx;
  dist2[t] = best;
}

extern "C" size_t pp_nearest_workspace(int64_t n_ref) {
  const size_t m = (size_t)(n_ref > 0 ? n_ref : 1);
  const size_t cap = (size_t)pp_hash_capacity((int64_t)m);
  return pp_align(sizeof(uint64_t) * m) * 2 + pp_align(sizeof(int32_t) * m) * 2 + pp_align(sizeof(float4) * m) +
         pp_align(sizeof(uint64_t) * cap) + pp_align(sizeof(int32_t) * cap) * 2 + pp_align(64 * sizeof(int32_t)) +
         pp_sort_pairs_workspace((int64_t)m) + 4096;
}

extern "C" int pp_nearest(const float* ref, int64_t n_ref, const float* query, int64_t n_query, int32_t dim, float cell,
                          float max_dist, int64_t* idx, float* dist2, void* workspace, size_t workspace_bytes,
                          pp_stream_t stream) {
  PP_REQUIRE(dim == 2 || dim == 3, "pp_nearest: dim must be 2 or 3");
  PP_REQUIRE(cell > 0.f, "pp_nearest: cell must be > 0");
  PP_REQUIRE(n_ref >= 0 && n_ref < (1ll << 31), "pp_nearest: n_ref out of range");
  PP_REQUIRE(n_query == 0 || (query && idx && dist2), "pp_nearest: null pointer");
  if (n_query == 0) return PP_OK;
  hipStream_t s = pp_s(stream);
  if (n_ref == 0) {
    PP_HIP(hipMemsetAsync(idx, 0xFF, sizeof(int64_t) * (size_t)n_query, s));
    PP_HIP(hipMemsetD32Async((hipDeviceptr_t)dist2, 0x7F800000, (size_t)n_query, s));
    return PP_OK;
  }
  PP_REQUIRE(ref && workspace, "pp_nearest: null pointer");
  if (workspace_bytes < pp_nearest_workspace(n_ref)) return PP_ERR_WORKSPACE;
  PPArena ar(workspace, workspace_bytes);
  const size_t m = (size_t)n_ref;
  const int64_t cap = pp_hash_capacity(n_ref);
  uint64_t* key = ar.take<uint64_t>(m);
  uint64_t* skey = ar.take<uint64_t>(m);
  int32_t* val = ar.take<int32_t>(m);
  int32_t* sval = ar.take<int32_t>(m);
  float4* spos = ar.take<float4>(m);
  uint64_t* hkeys = ar.take<uint64_t>((size_t)cap);
  int32_t* cstart = ar.take<int32_t>((size_t)cap);
  int32_t* cend = ar.take<int32_t>((size_t)cap);
  int32_t* misc = ar.take<int32_t>(64);  // [0] err
  PP_REQUIRE(key && skey && val && sval && spos && hkeys && cstart && cend && misc, "pp_nearest: workspace carve failed");
  PP_HIP(hipMemsetAsync(misc, 0, 64 * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_nn_keys, dim3(pp_blocks(n_ref, 256)), dim3(256), 0, s, ref, n_ref, dim, cell, key, val, misc);
  hipLaunchKernelGGL(k_nn_fill, dim3(pp_blocks(cap, 256)), dim3(256), 0, s, hkeys, cap);
  PP_LAUNCH_CHECK();
  int rc = pp_sort_pairs_u64(key, skey, val, sval, n_ref, 48, ar.cur(), ar.left(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nn_cells, dim3(pp_blocks(n_ref, 256)), dim3(256), 0, s, skey, sval, n_ref, ref, dim, hkeys, cstart,
                     cend, cap, spos);
  PP_LAUNCH_CHECK();
  int32_t err = 0;
  PP_HIP(hipMemcpyAsync(&err, misc, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PP_HIP(hipStreamSynchronize(s));
  if (err) {
    pp_set_error("pp_nearest: %d reference points are not finite or lie outside +-32000 cells of edge %g", err, cell);
    return PP_ERR_RANGE;
  }
  NNGrid g{hkeys, cstart, cend, cap, spos, cell, dim};
  hipLaunchKernelGGL(k_nn_query, dim3(pp_blocks(n_query, 256)), dim3(256), 0, s, g, skey, (int)n_ref, query, n_query,
                     max_dist, idx, dist2);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
