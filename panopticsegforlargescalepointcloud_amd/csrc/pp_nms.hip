// K14b: proposal overlaps, greedy NMS per batch element and instance painting, all on the device.
//
// Reference: PanopticResults.get_instances (torch_points3d/models/panoptic/structure_3heads.py:28-71: dense 0/1 masks,
// mask @ mask^T, IoU, non_max_suppression :6-16, size and score filters) followed by the tracker's
// get_cur_ins_pre_label (metrics/panoptic_tracker_pointgroup_npm3d.py:326-337: surviving clusters painted in ascending
// score order, so the best score covering a point wins).  The reference's O(nProp x N) mask product becomes
//   (1) pp_proposal_pairs   point -> proposal incidence table (a point belongs to at most PP_NMS_MAXM proposals: one per
//                           proposal source), every pair of proposals sharing a point counted in a hash table
//                           (one atomic per run of equal pairs in a wave), compacted into (a, b, |a n b|) triples;
//   (2) pp_nms_paint        IoU > threshold edges -> CSR adjacency, proposals ordered by (batch element, score), one
//                           thread per batch element walks its proposals by descending score (greedy NMS is
//                           sequential by definition; a batch element has ~10^2 proposals), filters, ranks, and a
//                           scatter-max paints the ranks.
// Order of equal scores: the reference sorts with numpy's argsort()[::-1] (introsort: not a stable sort, ties are
// implementation-defined).  Here ties are visited in DESCENDING proposal id -- what argsort()[::-1] gives whenever numpy's
// insertion-sort path runs (<= 16 elements) -- and documented as the tie rule of this implementation.
#include "pp_common.h"

#define NMS_MAXM 8  // proposals a point may belong to (proposal sources: raw / shifted region growing, mean shift, ...)

// proposal of entry e: last q with offsets[q] <= e
__device__ inline int nms_prop_of(const int32_t* __restrict__ offsets, int n_prop, int64_t e) {
  int lo = 0, hi = n_prop;  // offsets[lo] <= e < offsets[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)offsets[mid] <= e)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_nms_incidence(const int32_t* __restrict__ offsets, const int64_t* __restrict__ points,
                                                       int n_prop, int64_t total, int64_t n_points, int32_t* cnt, int32_t* tab,
                                                       int32_t* prop_of_entry, int32_t* info) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int q = nms_prop_of(offsets, n_prop, e);
  prop_of_entry[e] = q;
  const int64_t p = points[e];
  if (p < 0 || p >= n_points) {
    atomicAdd(&info[2], 1);
    return;
  }
  const int slot = atomicAdd(&cnt[p], 1);
  if (slot < NMS_MAXM)
    tab[p * NMS_MAXM + slot] = q;
  else
    atomicAdd(&info[0], 1);
}

__device__ inline void nms_hash_add(unsigned long long* keys, int32_t* vals, uint64_t cap, unsigned long long key, int add,
                                    int32_t* info) {
  uint64_t s = pp_mix64(key) & (cap - 1);
  for (uint64_t probe = 0; probe < cap; ++probe) {
    const unsigned long long prev = atomicCAS(&keys[s], (unsigned long long)PP_EMPTY_KEY, key);
    if (prev == PP_EMPTY_KEY || prev == key) {
      atomicAdd(&vals[s], add);
      return;
    }
    s = (s + 1) & (cap - 1);
  }
  atomicAdd(&info[1], 1);  // table full
}

// one lane per point: every pair (a < b) of its proposals gets +1.  Consecutive points mostly carry the same pair (two
// sources found the same instance), so equal keys of neighbouring lanes are merged into one atomic per run.
__global__ __launch_bounds__(256) void k_nms_pairs(const int32_t* __restrict__ cnt, const int32_t* __restrict__ tab,
                                                   int64_t n_points, int n_prop, unsigned long long* keys, int32_t* vals,
                                                   uint64_t cap, int32_t* info) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int m = 0;
  int q[NMS_MAXM];
  if (p < n_points) {
    m = cnt[p] < NMS_MAXM ? cnt[p] : NMS_MAXM;
    for (int i = 0; i < NMS_MAXM; ++i) q[i] = i < m ? tab[p * NMS_MAXM + i] : 0x7FFFFFFF;
    for (int i = 1; i < NMS_MAXM; ++i)  // insertion sort of <= 8 ids
      for (int j = i; j > 0 && q[j - 1] > q[j]; --j) {
        const int t = q[j];
        q[j] = q[j - 1];
        q[j - 1] = t;
      }
  }
  const int npair = m * (m - 1) / 2;
  int maxp = npair;
  for (int off = 32; off > 0; off >>= 1) maxp = max(maxp, __shfl_xor(maxp, off));
  int i = 0, j = 1;
  for (int t = 0; t < maxp; ++t) {
    const bool have = t < npair;
    unsigned long long key = PP_EMPTY_KEY;
    if (have) {
      key = (unsigned long long)q[i] * (unsigned long long)n_prop + (unsigned long long)q[j];
      if (++j >= m) {
        ++i;
        j = i + 1;
      }
    }
    // run-length merge inside the wave: a lane whose predecessor holds the same key passes its count on
    const unsigned long long prev = __shfl_up(key, 1);
    const bool head = have && (lane == 0 || prev != key);
    const unsigned long long heads = __ballot(head);
    const unsigned long long haves = __ballot(have);
    if (head) {
      // run = lanes [lane, next head or first lane without this key)
      const unsigned long long after = lane == 63 ? 0ull : (~0ull << (lane + 1));
      const unsigned long long stop = (heads | ~haves) & after;
      const int end = stop ? __builtin_ctzll(stop) : 64;
      nms_hash_add(keys, vals, cap, key, end - lane, info);
    }
  }
}

__global__ __launch_bounds__(256) void k_nms_compact(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ vals,
                                                     uint64_t cap, int n_prop, int64_t capacity, int32_t* pa, int32_t* pb,
                                                     int32_t* pinter, int32_t* n_pairs, int32_t* info) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cap) return;
  const unsigned long long k = keys[s];
  if (k == PP_EMPTY_KEY) return;
  const int i = atomicAdd(n_pairs, 1);
  if (i < capacity) {
    pa[i] = (int32_t)(k / (unsigned long long)n_prop);
    pb[i] = (int32_t)(k % (unsigned long long)n_prop);
    pinter[i] = vals[s];
  } else
    atomicAdd(&info[1], 1);
}

extern "C" int64_t pp_proposal_pairs_capacity(int32_t n_prop) {
  int64_t c = 1024;
  while (c < 16ll * n_prop) c <<= 1;
  return c;  // triples the caller must provide room for; the hash behind them has twice as many slots
}
extern "C" size_t pp_proposal_pairs_workspace(int64_t total_entries, int64_t n_points, int32_t n_prop) {
  const size_t cap = 2 * (size_t)pp_proposal_pairs_capacity(n_prop);
  return pp_align((size_t)std::max<int64_t>(n_points, 1) * 4) + pp_align((size_t)std::max<int64_t>(n_points, 1) * NMS_MAXM * 4) +
         pp_align(cap * 8) + pp_align(cap * 4) + 4096;
}
// info int32[4]: [0] points in more than PP_NMS_MAXM proposals, [1] pair table overflow, [2] point ids out of range
extern "C" int pp_proposal_pairs(const int32_t* prop_offsets, const int64_t* prop_points, int32_t n_prop,
                                 int64_t total_entries, int64_t n_points, int32_t* prop_of_entry, int32_t* pair_a, int32_t* pair_b, int32_t* pair_inter,
                                 int32_t* n_pairs, int32_t* info, void* workspace, size_t workspace_bytes,
                                 pp_stream_t stream) {
  PP_REQUIRE(prop_offsets && n_pairs && info, "pp_proposal_pairs: null pointer");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int32_t), s));
  PP_HIP(hipMemsetAsync(info, 0, 4 * sizeof(int32_t), s));
  const int64_t total = total_entries;  // == prop_offsets[n_prop], known to the caller (no device read-back here)
  if (n_prop == 0 || total == 0) return PP_OK;
  PP_REQUIRE(prop_points && prop_of_entry && pair_a && pair_b && pair_inter, "pp_proposal_pairs: null pointer");
  if (workspace_bytes < pp_proposal_pairs_workspace(total, n_points, n_prop)) return PP_ERR_WORKSPACE;
  const int64_t capacity = pp_proposal_pairs_capacity(n_prop);
  const uint64_t cap = 2 * (uint64_t)capacity;
  PPArena ar(workspace, workspace_bytes);
  int32_t* cnt = ar.take<int32_t>((size_t)n_points);
  int32_t* tab = ar.take<int32_t>((size_t)n_points * NMS_MAXM);
  unsigned long long* keys = ar.take<unsigned long long>(cap);
  int32_t* vals = ar.take<int32_t>(cap);
  PP_REQUIRE(cnt && tab && keys && vals, "pp_proposal_pairs: workspace carve failed");
  PP_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)n_points, s));
  PP_HIP(hipMemsetAsync(keys, 0xFF, sizeof(unsigned long long) * cap, s));
  PP_HIP(hipMemsetAsync(vals, 0, sizeof(int32_t) * cap, s));
  hipLaunchKernelGGL(k_nms_incidence, dim3(pp_blocks(total, 256)), dim3(256), 0, s, prop_offsets, prop_points, n_prop, total,
                     n_points, cnt, tab, prop_of_entry, info);
  hipLaunchKernelGGL(k_nms_pairs, dim3(pp_blocks(n_points, 256)), dim3(256), 0, s, cnt, tab, n_points, n_prop, keys, vals, cap,
                     info);
  hipLaunchKernelGGL(k_nms_compact, dim3(pp_blocks((int64_t)cap, 256)), dim3(256), 0, s, keys, vals, cap, n_prop, capacity,
                     pair_a, pair_b, pair_inter, n_pairs, info);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- NMS + paint -----------------------------------------------------------------------------------------------------------
__device__ inline uint32_t nms_orderable(float f) {  // monotone float -> uint32 (NaN sorts above everything)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// per proposal: batch element, sort key (batch, score) and id
__global__ __launch_bounds__(256) void k_nms_keys(const int32_t* __restrict__ offsets, const int64_t* __restrict__ points,
                                                  const int64_t* __restrict__ batch, const float* __restrict__ scores, int n_prop,
                                                  int n_groups, int32_t* group_of, unsigned long long* key, int32_t* idx,
                                                  int32_t* group_cnt, int32_t* info) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_prop) return;
  int g = 0;
  if (batch && offsets[q + 1] > offsets[q]) g = (int)batch[points[offsets[q]]];
  if (g < 0 || g >= n_groups) {
    atomicAdd(&info[3], 1);
    g = 0;
  }
  group_of[q] = g;
  atomicAdd(&group_cnt[g], 1);
  key[q] = ((unsigned long long)(uint32_t)g << 32) | (scores ? nms_orderable(scores[q]) : (uint32_t)q);
  idx[q] = q;
}
// edges: pairs with IoU > threshold (float32 arithmetic like the reference's torch tensors)
__global__ __launch_bounds__(256) void k_nms_degree(const int32_t* __restrict__ pa, const int32_t* __restrict__ pb,
                                                    const int32_t* __restrict__ pinter, const int32_t* __restrict__ n_pairs,
                                                    const int32_t* __restrict__ offsets, float thr, int32_t* deg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs[0]) return;
  const int a = pa[i], b = pb[i];
  const float inter = (float)pinter[i];
  const float sa = (float)(offsets[a + 1] - offsets[a]), sb = (float)(offsets[b + 1] - offsets[b]);
  if (inter / (sa + sb - inter) > thr) {
    atomicAdd(&deg[a], 1);
    atomicAdd(&deg[b], 1);
  }
}
__global__ __launch_bounds__(256) void k_nms_fill(const int32_t* __restrict__ pa, const int32_t* __restrict__ pb,
                                                  const int32_t* __restrict__ pinter, const int32_t* __restrict__ n_pairs,
                                                  const int32_t* __restrict__ offsets, float thr, const int32_t* __restrict__ adj_off,
                                                  int32_t* fill, int32_t* adj) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs[0]) return;
  const int a = pa[i], b = pb[i];
  const float inter = (float)pinter[i];
  const float sa = (float)(offsets[a + 1] - offsets[a]), sb = (float)(offsets[b + 1] - offsets[b]);
  if (inter / (sa + sb - inter) > thr) {
    adj[adj_off[a] + atomicAdd(&fill[a], 1)] = b;
    adj[adj_off[b] + atomicAdd(&fill[b], 1)] = a;
  }
}
// one thread per batch element: greedy NMS over its proposals by descending (score, id), filters, ranks
__global__ __launch_bounds__(64) void k_nms_greedy(const int32_t* __restrict__ sorted_idx, const int32_t* __restrict__ group_start,
                                                   int n_groups, const int32_t* __restrict__ adj_off, const int32_t* __restrict__ adj,
                                                   const int32_t* __restrict__ offsets, const float* __restrict__ scores,
                                                   int min_points, float min_score, uint8_t* state /*0 new, 1 kept, 2 suppressed*/,
                                                   int32_t* rank, int32_t* counts) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const int lo = group_start[g], hi = group_start[g + 1];
  if (!scores) {  // no ScoreNet: every proposal is an instance, in proposal order (structure_3heads.py:34-35)
    for (int i = lo; i < hi; ++i) rank[sorted_idx[i]] = i - lo;
    counts[g] = hi - lo;
    return;
  }
  for (int i = hi - 1; i >= lo; --i) {  // descending score, ties by descending id
    const int q = sorted_idx[i];
    if (state[q] == 2) continue;
    state[q] = 1;
    for (int e = adj_off[q]; e < adj_off[q + 1]; ++e)
      if (state[adj[e]] == 0) state[adj[e]] = 2;
  }
  // rank = position among the kept proposals in ascending score; equal scores keep the pick order (stable argsort of
  // the pick list, tracker :331-333), i.e. descending id inside a group of equal scores
  int r = 0;
  int i = lo;
  while (i < hi) {
    int j = i + 1;
    const float sc = scores[sorted_idx[i]];
    while (j < hi && scores[sorted_idx[j]] == sc) ++j;
    for (int t = j - 1; t >= i; --t) {
      const int q = sorted_idx[t];
      const int size = offsets[q + 1] - offsets[q];
      if (state[q] == 1 && size > min_points && sc > min_score) rank[q] = r++;
    }
    i = j;
  }
  counts[g] = r;
}
__global__ __launch_bounds__(256) void k_nms_group_start(const unsigned long long* __restrict__ sorted_key, int n_prop, int n_groups,
                                                         int32_t* group_start) {
  // group_start[g] = first position whose group >= g (lower bound); one thread per group boundary
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > n_groups) return;
  int lo = 0, hi = n_prop;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)(sorted_key[mid] >> 32) < g)
      lo = mid + 1;
    else
      hi = mid;
  }
  group_start[g] = lo;
}
__global__ __launch_bounds__(256) void k_nms_paint(const int32_t* __restrict__ prop_of_entry, const int64_t* __restrict__ points,
                                                   int64_t total, const int32_t* __restrict__ rank, int32_t* labels) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int r = rank[prop_of_entry[e]];
  if (r >= 0) atomicMax(&labels[points[e]], r);  // ascending score = ascending rank: the best cluster covering a point wins
}

extern "C" size_t pp_nms_paint_workspace(int32_t n_prop, int32_t n_groups, int64_t pair_capacity) {
  const size_t p = (size_t)std::max(n_prop, 1), g = (size_t)std::max(n_groups, 1), c = (size_t)std::max<int64_t>(pair_capacity, 1);
  return 2 * pp_align(p * 8) + 6 * pp_align((p + 1) * 4) + pp_align(p) + 2 * pp_align((g + 1) * 4) + pp_align(2 * c * 4) +
         pp_sort_pairs_workspace(n_prop) + pp_scan_workspace(n_prop + 1) + 4096;
}
// labels int32 [n_points] (-1 = no instance; ids restart at 0 in every batch element), counts int32 [n_groups],
// rank int32 [n_prop] (-1 = dropped).  batch NULL: all proposals form one group (n_groups = 1).  scores NULL: no
// ScoreNet.  info int32[4] as filled by pp_proposal_pairs ([3] += proposals whose batch element is out of range).
extern "C" int pp_nms_paint(const int32_t* prop_offsets, const int64_t* prop_points, int32_t n_prop, int64_t total_entries,
                            int64_t n_points, const int32_t* prop_of_entry, const int32_t* pair_a, const int32_t* pair_b,
                            const int32_t* pair_inter, const int32_t* n_pairs, int64_t pair_capacity, const int64_t* batch,
                            int32_t n_groups, const float* scores, float nms_threshold, int32_t min_cluster_points,
                            float min_score, int32_t* labels, int32_t* counts, int32_t* rank, int32_t* info, void* workspace,
                            size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(labels || n_points == 0, "pp_nms_paint: null labels");
  PP_REQUIRE(counts && n_groups >= 1, "pp_nms_paint: bad groups");
  hipStream_t s = pp_s(stream);
  if (n_points > 0) PP_HIP(hipMemsetAsync(labels, 0xFF, sizeof(int32_t) * (size_t)n_points, s));
  PP_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)n_groups, s));
  if (n_prop == 0) return PP_OK;
  PP_REQUIRE(prop_offsets && prop_points && prop_of_entry && rank && info, "pp_nms_paint: null pointer");
  PP_REQUIRE(!scores || (pair_a && pair_b && pair_inter && n_pairs), "pp_nms_paint: pairs are required with scores");
  if (workspace_bytes < pp_nms_paint_workspace(n_prop, n_groups, pair_capacity)) return PP_ERR_WORKSPACE;
  PPArena ar(workspace, workspace_bytes);
  unsigned long long* key = ar.take<unsigned long long>((size_t)n_prop);
  unsigned long long* key2 = ar.take<unsigned long long>((size_t)n_prop);
  int32_t* idx = ar.take<int32_t>((size_t)n_prop + 1);
  int32_t* idx2 = ar.take<int32_t>((size_t)n_prop + 1);
  int32_t* group_of = ar.take<int32_t>((size_t)n_prop + 1);
  int32_t* deg = ar.take<int32_t>((size_t)n_prop + 1);
  int32_t* adj_off = ar.take<int32_t>((size_t)n_prop + 1);
  int32_t* fill = ar.take<int32_t>((size_t)n_prop + 1);
  uint8_t* state = ar.take<uint8_t>((size_t)n_prop);
  int32_t* group_cnt = ar.take<int32_t>((size_t)n_groups + 1);
  int32_t* group_start = ar.take<int32_t>((size_t)n_groups + 1);
  int32_t* adj = ar.take<int32_t>(2 * (size_t)std::max<int64_t>(pair_capacity, 1));
  PP_REQUIRE(key && key2 && idx && idx2 && group_of && deg && adj_off && fill && state && group_cnt && group_start && adj,
             "pp_nms_paint: workspace carve failed");
  const unsigned pb = pp_blocks(n_prop, 256);
  PP_HIP(hipMemsetAsync(rank, 0xFF, sizeof(int32_t) * (size_t)n_prop, s));
  PP_HIP(hipMemsetAsync(group_cnt, 0, sizeof(int32_t) * ((size_t)n_groups + 1), s));
  PP_HIP(hipMemsetAsync(deg, 0, sizeof(int32_t) * ((size_t)n_prop + 1), s));
  PP_HIP(hipMemsetAsync(fill, 0, sizeof(int32_t) * ((size_t)n_prop + 1), s));
  PP_HIP(hipMemsetAsync(state, 0, (size_t)n_prop, s));
  hipLaunchKernelGGL(k_nms_keys, dim3(pb), dim3(256), 0, s, prop_offsets, prop_points, batch, scores, n_prop, n_groups, group_of,
                     key, idx, group_cnt, info);
  PP_LAUNCH_CHECK();
  int rc = pp_sort_pairs_u64((const uint64_t*)key, (uint64_t*)key2, idx, idx2, n_prop, 64, ar.cur(), ar.left(), s);  // stable: ties keep ascending id
  if (rc) return rc;
  hipLaunchKernelGGL(k_nms_group_start, dim3(pp_blocks(n_groups + 1, 256)), dim3(256), 0, s, key2, n_prop, n_groups, group_start);
  if (scores) {
    const unsigned cb = pp_blocks(pair_capacity, 256);
    hipLaunchKernelGGL(k_nms_degree, dim3(cb), dim3(256), 0, s, pair_a, pair_b, pair_inter, n_pairs, prop_offsets, nms_threshold, deg);
    PP_LAUNCH_CHECK();
    rc = pp_exclusive_scan_i32(deg, adj_off, (int64_t)n_prop + 1, nullptr, ar.cur(), ar.left(), s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_nms_fill, dim3(cb), dim3(256), 0, s, pair_a, pair_b, pair_inter, n_pairs, prop_offsets, nms_threshold,
                       adj_off, fill, adj);
  }
  hipLaunchKernelGGL(k_nms_greedy, dim3(pp_blocks(n_groups, 64)), dim3(64), 0, s, idx2, group_start, n_groups, adj_off, adj,
                     prop_offsets, scores, min_cluster_points, min_score, state, rank, counts);
  PP_LAUNCH_CHECK();
  if (total_entries > 0) {
    hipLaunchKernelGGL(k_nms_paint, dim3(pp_blocks(total_entries, 256)), dim3(256), 0, s, prop_of_entry, prop_points,
                       total_entries, rank, labels);
    PP_LAUNCH_CHECK();
  }
  return PP_OK;
}
