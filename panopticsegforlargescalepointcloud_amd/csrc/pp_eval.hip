// f2 / f3: scene assembly and final evaluation on the device.
//
//   pp_histogram2d   out[a][b] += 1 per point -- the class confusion matrix (ground truth x prediction) and the
//                    instance x class tables behind the reference's final evaluation
//                    (torch_points3d/datasets/panoptic/npm3d.py:107-397: per-class TP / counts :120-160, the mode of an
//                    instance's semantic labels :190-215) and the tracker's confusion matrix
//                    (metrics/panoptic_tracker_pointgroup_npm3d.py:711-879).
//   pp_pair_counts   distinct (a, b) pairs of two label arrays with their multiplicities -- the (predicted instance,
//                    ground-truth instance) contingency table the reference builds with one boolean mask per pair
//                    (npm3d.py:232-300), in sparse form.
//   pp_block_merge   the tracker's ORDER-DEPENDENT greedy merge of one cylinder's instance labels into the scene labels
//                    (panoptic_tracker_pointgroup_npm3d.py:339-452) as: gather of the current scene labels, the sparse
//                    (new instance, old label) contingency table of the block, a decision pass over that small table
//                    (one wave: the merge is sequential over instances because every merge changes the sizes later
//                    IoUs see), and a relabelling scatter.  Blocks are fed in the original block order by the caller;
//                    `max_instance` lives in device memory so that consecutive blocks chain on the stream without a
//                    host round trip.
// Integer / index work: bit-exact against the NumPy restatements (scene.block_merging, panoptic/metrics.py).
#include "pp_common.h"

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hist2d(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t n, int na,
                                                int nb, unsigned long long* out, int32_t* info) {
  extern __shared__ unsigned int h_lds[];
  const int bins = na * nb;
  const bool use_lds = bins <= 8192;
  if (use_lds) {
    for (int i = threadIdx.x; i < bins; i += blockDim.x) h_lds[i] = 0u;
    __syncthreads();
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t x = a[i], y = b[i];
    if (x < 0 || y < 0) {
      atomicAdd(&info[0], 1);  // skipped (no label / no instance)
      continue;
    }
    if (x >= na || y >= nb) {
      atomicAdd(&info[1], 1);  // out of range: caller error
      continue;
    }
    if (use_lds)
      atomicAdd(&h_lds[x * nb + y], 1u);
    else
      atomicAdd(&out[x * nb + y], 1ull);
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += blockDim.x)
      if (h_lds[i]) atomicAdd(&out[i], (unsigned long long)h_lds[i]);
  }
}
// out int64 [na][nb] (zeroed here); info int32[2] = {rows skipped because a < 0 or b < 0, rows out of range}
extern "C" int pp_histogram2d(const int64_t* a, const int64_t* b, int64_t n, int32_t na, int32_t nb, int64_t* out,
                              int32_t* info, pp_stream_t stream) {
  PP_REQUIRE(out && info && na >= 1 && nb >= 1, "pp_histogram2d: bad arguments");
  PP_REQUIRE((int64_t)na * nb < (1ll << 31), "pp_histogram2d: table too large");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(out, 0, sizeof(int64_t) * (size_t)na * nb, s));
  PP_HIP(hipMemsetAsync(info, 0, 2 * sizeof(int32_t), s));
  if (n == 0) return PP_OK;
  PP_REQUIRE(a && b, "pp_histogram2d: null labels");
  const int bins = na * nb;
  const unsigned grid = (unsigned)std::min<int64_t>(pp_blocks(n, 256 * 8), 2048);
  hipLaunchKernelGGL(k_hist2d, dim3(grid), dim3(256), bins <= 8192 ? sizeof(unsigned) * bins : 0, s, a, b, n, na, nb,
                     (unsigned long long*)out, info);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// sparse pair counts: open-addressing hash (key = a * nb + b), one atomic per run of equal keys in a wave
__device__ inline void ev_hash_add(unsigned long long* keys, unsigned long long* vals, uint64_t cap, unsigned long long key,
                                   unsigned long long add, int32_t* overflow) {
  uint64_t s = pp_mix64(key) & (cap - 1);
  for (uint64_t probe = 0; probe < cap; ++probe) {
    const unsigned long long prev = atomicCAS(&keys[s], (unsigned long long)PP_EMPTY_KEY, key);
    if (prev == PP_EMPTY_KEY || prev == key) {
      atomicAdd(&vals[s], add);
      return;
    }
    s = (s + 1) & (cap - 1);
  }
  atomicAdd(overflow, 1);
}
// (values are read with an agent-scope atomic load: the decision pass updates sizes with atomics that bypass the L1)
__device__ inline bool ev_hash_get(unsigned long long* keys, unsigned long long* vals, uint64_t cap, unsigned long long key,
                                   unsigned long long* out) {
  uint64_t s = pp_mix64(key) & (cap - 1);
  for (uint64_t probe = 0; probe < cap; ++probe) {
    const unsigned long long k = keys[s];
    if (k == key) {
      *out = __hip_atomic_load(&vals[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    if (k == PP_EMPTY_KEY) return false;
    s = (s + 1) & (cap - 1);
  }
  return false;
}
__device__ inline void ev_add_runs(unsigned long long* keys, unsigned long long* vals, uint64_t cap, unsigned long long key,
                                   bool have, int32_t* overflow) {
  const int lane = threadIdx.x & 63;
  const unsigned long long prev = __shfl_up(key, 1);
  const bool head = have && (lane == 0 || prev != key);
  const unsigned long long heads = __ballot(head);
  const unsigned long long haves = __ballot(have);
  if (head) {
    const unsigned long long after = lane == 63 ? 0ull : (~0ull << (lane + 1));
    const unsigned long long stop = (heads | ~haves) & after;
    const int end = stop ? __builtin_ctzll(stop) : 64;
    ev_hash_add(keys, vals, cap, key, (unsigned long long)(end - lane), overflow);
  }
}
__global__ __launch_bounds__(256) void k_pair_counts(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t n,
                                                     int64_t nb, unsigned long long* keys, unsigned long long* vals,
                                                     uint64_t cap, int32_t* info) {
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t rounds = (n + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {  // uniform trip count: the run merge needs the whole wave
    const int64_t i = i0 + r * stride;
    bool have = false;
    unsigned long long key = PP_EMPTY_KEY;
    if (i < n) {
      const int64_t x = a[i], y = b[i];
      if (x >= 0 && y >= 0) {
        if (y >= nb)
          atomicAdd(&info[1], 1);
        else {
          have = true;
          key = (unsigned long long)x * (unsigned long long)nb + (unsigned long long)y;
        }
      }
    }
    ev_add_runs(keys, vals, cap, key, have, &info[0]);
  }
}
__global__ __launch_bounds__(256) void k_pair_compact(const unsigned long long* __restrict__ keys,
                                                      const unsigned long long* __restrict__ vals, uint64_t cap, int64_t nb,
                                                      int64_t capacity, int64_t* pa, int64_t* pb, int64_t* cnt, int32_t* n_pairs,
                                                      int32_t* info) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cap) return;
  const unsigned long long k = keys[s];
  if (k == PP_EMPTY_KEY) return;
  const int i = atomicAdd(n_pairs, 1);
  if (i < capacity) {
    pa[i] = (int64_t)(k / (unsigned long long)nb);
    pb[i] = (int64_t)(k % (unsigned long long)nb);
    cnt[i] = (int64_t)vals[s];
  } else
    atomicAdd(&info[0], 1);
}
extern "C" size_t pp_pair_counts_workspace(int64_t capacity) {
  size_t cap = 1024;
  while ((int64_t)cap < 2 * capacity) cap <<= 1;
  return 2 * pp_align(cap * 8) + 1024;
}
// Distinct (a[i], b[i]) pairs with a, b >= 0 (rows with a negative label are skipped) and how often each occurs.
// b must be < nb.  pair_a / pair_b / count have `capacity` slots; n_pairs (device int32) = number of pairs, unordered.
// info int32[2] = {table / output overflow (raise capacity), b >= nb}.
extern "C" int pp_pair_counts(const int64_t* a, const int64_t* b, int64_t n, int64_t nb, int64_t capacity, int64_t* pair_a,
                              int64_t* pair_b, int64_t* count, int32_t* n_pairs, int32_t* info, void* workspace,
                              size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(n_pairs && info && capacity >= 1 && nb >= 1, "pp_pair_counts: bad arguments");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int32_t), s));
  PP_HIP(hipMemsetAsync(info, 0, 2 * sizeof(int32_t), s));
  if (n == 0) return PP_OK;
  PP_REQUIRE(a && b && pair_a && pair_b && count, "pp_pair_counts: null pointer");
  if (workspace_bytes < pp_pair_counts_workspace(capacity)) return PP_ERR_WORKSPACE;
  uint64_t cap = 1024;
  while ((int64_t)cap < 2 * capacity) cap <<= 1;
  PPArena ar(workspace, workspace_bytes);
  unsigned long long* keys = ar.take<unsigned long long>(cap);
  unsigned long long* vals = ar.take<unsigned long long>(cap);
  PP_HIP(hipMemsetAsync(keys, 0xFF, 8 * cap, s));
  PP_HIP(hipMemsetAsync(vals, 0, 8 * cap, s));
  const unsigned grid = (unsigned)std::min<int64_t>(pp_blocks(n, 256), 4096);
  hipLaunchKernelGGL(k_pair_counts, dim3(grid), dim3(256), 0, s, a, b, n, nb, keys, vals, cap, info);
  hipLaunchKernelGGL(k_pair_compact, dim3(pp_blocks((int64_t)cap, 256)), dim3(256), 0, s, keys, vals, cap, nb, capacity, pair_a,
                     pair_b, count, n_pairs, info);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// block merging
struct BMState {
  int32_t any_has, any_none, any_valid, t_num;  // flags of the block
  int32_t overflow, bad;                        // table overflow / origin id out of range
  int32_t n_pairs, pad;
};
// cur[i] = scene[origin[i]]; flags; (instance, old label + 1) pair table; sizes of the old labels inside the block
__global__ __launch_bounds__(256) void k_bm_tables(const int64_t* __restrict__ origin, const int32_t* __restrict__ pre,
                                                   int64_t n, const int64_t* __restrict__ scene, int64_t n_scene, int64_t* cur,
                                                   unsigned long long* pkeys, unsigned long long* pvals, uint64_t pcap,
                                                   unsigned long long* skeys, unsigned long long* svals, uint64_t scap,
                                                   BMState* st) {
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t rounds = (n + stride - 1) / stride;
  int has = 0, none = 0, valid = 0, tmax = -1;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t i = i0 + r * stride;
    bool have_p = false, have_s = false;
    unsigned long long pk = PP_EMPTY_KEY, sk = PP_EMPTY_KEY;
    if (i < n) {
      const int64_t o = origin[i];
      if (o < 0 || o >= n_scene)
        atomicAdd(&st->bad, 1);
      else {
        const int64_t c = scene[o];
        cur[i] = c;
        const int p = pre[i];
        if (c != -1) has = 1; else none = 1;
        if (p != -1) {
          valid = 1;
          tmax = max(tmax, p);
          have_p = true;
          pk = ((unsigned long long)(uint32_t)p << 32) | (unsigned long long)(uint32_t)(c + 1);  // labels < 2^32 - 1
        }
        if (c != -1) {
          have_s = true;
          sk = (unsigned long long)(c + 1);
        }
      }
    }
    ev_add_runs(pkeys, pvals, pcap, pk, have_p, &st->overflow);
    ev_add_runs(skeys, svals, scap, sk, have_s, &st->overflow);
  }
  if (__ballot(has)) { if ((threadIdx.x & 63) == 0) atomicOr(&st->any_has, 1); }
  if (__ballot(none)) { if ((threadIdx.x & 63) == 0) atomicOr(&st->any_none, 1); }
  if (__ballot(valid)) { if ((threadIdx.x & 63) == 0) atomicOr(&st->any_valid, 1); }
  for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));
  if ((threadIdx.x & 63) == 0 && tmax >= 0) atomicMax(&st->t_num, tmax + 1);
}
__global__ __launch_bounds__(256) void k_bm_compact(const unsigned long long* __restrict__ pkeys,
                                                    const unsigned long long* __restrict__ pvals, uint64_t pcap, int32_t* pi,
                                                    int64_t* pg, int64_t* pc, int64_t capacity, BMState* st) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= pcap) return;
  const unsigned long long k = pkeys[s];
  if (k == PP_EMPTY_KEY) return;
  const int i = atomicAdd(&st->n_pairs, 1);
  if (i < capacity) {
    pi[i] = (int32_t)(k >> 32);
    pg[i] = (int64_t)(uint32_t)k - 1;  // old label, -1 = none
    pc[i] = (int64_t)pvals[s];
  } else
    atomicAdd(&st->overflow, 1);
}
// one wave: the decision pass.  relabel[ii] = label the so far unlabelled points of instance ii receive, or -2 (none).
__global__ __launch_bounds__(64) void k_bm_decide(const int32_t* __restrict__ pi, const int64_t* __restrict__ pg,
                                                  const int64_t* __restrict__ pc, unsigned long long* skeys,
                                                  unsigned long long* svals, uint64_t scap, BMState* st, int64_t* max_instance,
                                                  int64_t* relabel, int64_t relabel_cap) {
  const int lane = threadIdx.x;
  const int t_num = st->t_num;
  if (t_num > relabel_cap && lane == 0) atomicAdd(&st->bad, 1);  // labels must be < number of block points
  if (!st->any_valid || st->overflow || st->bad || t_num > relabel_cap) return;  // nothing to merge / error (reported by host)
  int64_t mx = max_instance[0];
  if (!st->any_has) {  // untouched region: labels shifted by max_instance (:372-376)
    for (int ii = lane; ii < t_num; ii += 64) relabel[ii] = (int64_t)ii + mx;
    if (lane == 0) max_instance[0] = mx + t_num;
    return;
  }
  for (int ii = lane; ii < t_num; ii += 64) relabel[ii] = -2;
  if (!st->any_none) return;  // every point already labelled: nothing changes (:377-378)
  const int P = st->n_pairs;
  for (int ii = 0; ii < t_num; ++ii) {
    // this instance's row of the table: points in total, unlabelled points, best old label by IoU
    long long n_pts = 0, n_none = 0;
    double best_iou = 0.0;
    long long best_g = 0x7FFFFFFFFFFFFFFFll;
    for (int e = lane; e < P; e += 64) {
      if (pi[e] != ii) continue;
      n_pts += pc[e];
      if (pg[e] == -1) n_none += pc[e];
    }
    for (int off = 32; off > 0; off >>= 1) {
      n_pts += __shfl_xor(n_pts, off);
      n_none += __shfl_xor(n_none, off);
    }
    const long long n_has = n_pts - n_none;
    if (n_has == 0) {  // new instance (also when the id is unused: the reference still burns a label, :430-433)
      if (lane == 0) relabel[ii] = mx + 1;
      mx += 1;
      continue;
    }
    if (n_none == 0) continue;  // fully labelled already
    for (int e = lane; e < P; e += 64) {
      if (pi[e] != ii || pg[e] == -1) continue;
      unsigned long long size_g = 0;
      ev_hash_get(skeys, svals, scap, (unsigned long long)(pg[e] + 1), &size_g);
      const double inter = (double)pc[e];
      const double uni = (double)((long long)size_g + n_pts - pc[e]);
      const double iou = inter / uni;
      if (iou > best_iou || (iou == best_iou && iou > 0.0 && pg[e] < best_g)) {
        best_iou = iou;
        best_g = pg[e];
      }
    }
    for (int off = 32; off > 0; off >>= 1) {  // max IoU, smallest label among equal IoUs (ascending np.unique order, strict >)
      const double oi = __shfl_xor(best_iou, off);
      const long long og = __shfl_xor(best_g, off);
      if (oi > best_iou || (oi == best_iou && og < best_g)) {
        best_iou = oi;
        best_g = og;
      }
    }
    if (best_iou > 0.1) {  // hard-coded in the reference (:447)
      if (lane == 0) {
        relabel[ii] = best_g;
        int32_t dummy = 0;
        ev_hash_add(skeys, svals, scap, (unsigned long long)(best_g + 1), (unsigned long long)n_none, &dummy);  // later IoUs see it
      }
      __threadfence_block();
    } else {
      if (lane == 0) relabel[ii] = mx + 1;
      mx += 1;
    }
  }
  if (lane == 0) max_instance[0] = mx;
}
__global__ __launch_bounds__(256) void k_bm_apply(const int64_t* __restrict__ origin, const int32_t* __restrict__ pre,
                                                  const int64_t* __restrict__ cur, int64_t n, const int64_t* __restrict__ relabel,
                                                  const BMState* __restrict__ st, int64_t relabel_cap, int64_t* scene) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !st->any_valid || st->overflow || st->bad || st->t_num > relabel_cap) return;
  const int p = pre[i];
  if (p == -1 || cur[i] != -1) return;
  const int64_t l = relabel[p];
  if (l != -2) scene[origin[i]] = l;
}
#define BM_HASH_SLOTS 65536  // distinct (instance, old label) pairs of ONE block: a few hundred; overflow is reported
extern "C" size_t pp_block_merge_workspace(int64_t n) {
  const size_t m = (size_t)std::max<int64_t>(n, 1);
  const size_t pcap = BM_HASH_SLOTS;
  return pp_align(m * 8) /*cur*/ + 2 * pp_align(pcap * 8) + 2 * pp_align(pcap * 8) /*sizes share the capacity*/ +
         pp_align((pcap / 2) * 4) + 2 * pp_align((pcap / 2) * 8) /*compacted pairs*/ + pp_align(m * 8) /*relabel*/ + 4096;
}
// scene_labels int64 [n_scene] (in/out, -1 = none), max_instance int64[1] on the device (in/out), state int32[8] (device,
// out): {any_has, any_none, any_valid, t_num, overflow, bad origin ids, n_pairs, 0} -- overflow / bad must be 0.
extern "C" int pp_block_merge(const int64_t* origin_ids, const int32_t* block_labels, int64_t n, int64_t* scene_labels,
                              int64_t n_scene, int64_t* max_instance, int32_t* state, void* workspace, size_t workspace_bytes,
                              pp_stream_t stream) {
  PP_REQUIRE(scene_labels && max_instance && state, "pp_block_merge: null pointer");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(state, 0, sizeof(BMState), s));
  if (n == 0) return PP_OK;
  PP_REQUIRE(origin_ids && block_labels, "pp_block_merge: null pointer");
  if (workspace_bytes < pp_block_merge_workspace(n)) return PP_ERR_WORKSPACE;
  const uint64_t pcap = BM_HASH_SLOTS;
  const int64_t capacity = (int64_t)(pcap / 2);
  PPArena ar(workspace, workspace_bytes);
  int64_t* cur = ar.take<int64_t>((size_t)n);
  unsigned long long* pkeys = ar.take<unsigned long long>(pcap);
  unsigned long long* pvals = ar.take<unsigned long long>(pcap);
  unsigned long long* skeys = ar.take<unsigned long long>(pcap);
  unsigned long long* svals = ar.take<unsigned long long>(pcap);
  int32_t* pi = ar.take<int32_t>((size_t)capacity);
  int64_t* pg = ar.take<int64_t>((size_t)capacity);
  int64_t* pc = ar.take<int64_t>((size_t)capacity);
  int64_t* relabel = ar.take<int64_t>((size_t)n);
  PP_REQUIRE(cur && pkeys && pvals && skeys && svals && pi && pg && pc && relabel, "pp_block_merge: workspace carve failed");
  PP_HIP(hipMemsetAsync(pkeys, 0xFF, 8 * pcap, s));
  PP_HIP(hipMemsetAsync(pvals, 0, 8 * pcap, s));
  PP_HIP(hipMemsetAsync(skeys, 0xFF, 8 * pcap, s));
  PP_HIP(hipMemsetAsync(svals, 0, 8 * pcap, s));
  BMState* st = (BMState*)state;
  const unsigned grid = (unsigned)std::min<int64_t>(pp_blocks(n, 256), 2048);
  hipLaunchKernelGGL(k_bm_tables, dim3(grid), dim3(256), 0, s, origin_ids, block_labels, n, scene_labels, n_scene, cur, pkeys,
                     pvals, pcap, skeys, svals, pcap, st);
  hipLaunchKernelGGL(k_bm_compact, dim3(pp_blocks((int64_t)pcap, 256)), dim3(256), 0, s, pkeys, pvals, pcap, pi, pg, pc, capacity,
                     st);
  hipLaunchKernelGGL(k_bm_decide, dim3(1), dim3(64), 0, s, pi, pg, pc, skeys, svals, pcap, st, max_instance, relabel, n);
  hipLaunchKernelGGL(k_bm_apply, dim3(pp_blocks(n, 256)), dim3(256), 0, s, origin_ids, block_labels, cur, n, relabel, st, n,
                     scene_labels);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
