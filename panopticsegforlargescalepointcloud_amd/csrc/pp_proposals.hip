// Front end of the proposal scorer: duplicate proposals and the scorer's input batch in a handful of launches.
//
// Reference: torch_points3d/models/panoptic/PointGroup3heads.py:393-454 feeds EVERY proposal to the scorer network -- a batch
// whose elements are the proposals: batch index = proposal, coordinates / features = those of the proposal's points.  Region
// growing and mean shift often return the same point list for a well-separated instance, and identical lists get identical
// scores, so one representative per list is scored (results unchanged).  The first version of this step was ~130 elementwise /
// scan / sort launches of a tensor library (hashes by prefix sums, a sort of the proposals, boolean compaction, the gather of the
// surviving lists, repeat_interleave for the batch index, a gather and a concatenation for the coordinates): 5 ms of a 135 ms
// bench step during which the GPU waits for the host to issue 5-microsecond kernels.  Here:
//   pp_proposals_unique  signature per proposal (two 64-bit sum hashes of its point ids, one workgroup per proposal) ->
//                        hash table keyed by (hashes, size) holding the smallest proposal index per key -> entry-by-entry
//                        verification against that representative (a mismatch keeps the proposal: exact) -> positions and
//                        offsets of the kept proposals by two prefix sums
//   pp_proposals_emit    the kept lists, their batch index and their (batch, x, y, z) coordinate rows in one pass
#include <algorithm>

#include "pp_common.h"

#define PR_TPB 256

struct PropSig {
  unsigned long long h1, h2;
};

__device__ inline unsigned long long pr_block_sum(unsigned long long v, unsigned long long* sh) {
  for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // (sh may still be read from the previous call)
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  unsigned long long s = 0;
  for (int w = 0; w < PR_TPB / 64; ++w) s += sh[w];
  return s;
}

// one workgroup per proposal: order-free signature of its point list; points outside [0, n_points) are counted in bad[0]
__global__ __launch_bounds__(PR_TPB) void k_prop_signature(const int32_t* __restrict__ offs, const int64_t* __restrict__ pts,
                                                           int64_t n_points, PropSig* __restrict__ sig, int32_t* bad) {
  __shared__ unsigned long long sh[PR_TPB / 64];
  const int64_t p = blockIdx.x;
  const int lo = offs[p], hi = offs[p + 1];
  unsigned long long a = 0, b = 0;
  int nbad = 0;
  for (int e = lo + (int)threadIdx.x; e < hi; e += PR_TPB) {
    const int64_t v = pts[e];
    nbad += (v < 0 || v >= n_points) ? 1 : 0;
    a += pp_mix64((unsigned long long)v + 0x9E3779B97F4A7C15ull);
    b += pp_mix64((unsigned long long)v ^ 0xC2B2AE3D27D4EB4Full) * 0x9FB21C651E98DF25ull;
  }
  a = pr_block_sum(a, sh);
  b = pr_block_sum(b, sh);
  if (threadIdx.x == 0) sig[p] = PropSig{a, b};
  if (nbad) atomicAdd(bad, nbad);
}

__device__ inline unsigned long long pr_key(PropSig s, int size) {
  unsigned long long k = s.h1 ^ (s.h2 * 0xD6E8FEB86659FD93ull) ^ ((unsigned long long)size * 0x9E3779B97F4A7C15ull);
  return k == PP_EMPTY_KEY ? 0ull : k;
}

// smallest proposal index per key (open addressing; the result does not depend on the order the threads arrive in)
__global__ __launch_bounds__(PR_TPB) void k_prop_insert(const int32_t* __restrict__ offs, const PropSig* __restrict__ sig, int64_t P,
                                                        unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tmin,
                                                        unsigned cap_mask) {
  const int64_t p = (int64_t)blockIdx.x * PR_TPB + threadIdx.x;
  if (p >= P) return;
  const unsigned long long k = pr_key(sig[p], offs[p + 1] - offs[p]);
  unsigned slot = (unsigned)pp_mix64(k) & cap_mask;
  for (;;) {
    const unsigned long long old = atomicCAS(&tkeys[slot], PP_EMPTY_KEY, k);
    if (old == PP_EMPTY_KEY || old == k) {
      atomicMin(&tmin[slot], (int32_t)p);
      return;
    }
    slot = (slot + 1) & cap_mask;
  }
}

// one workgroup per proposal: candidate representative from the table, verified size, hashes and entry by entry
__global__ __launch_bounds__(PR_TPB) void k_prop_verify(const int32_t* __restrict__ offs, const int64_t* __restrict__ pts,
                                                        const PropSig* __restrict__ sig, const unsigned long long* __restrict__ tkeys,
                                                        const int32_t* __restrict__ tmin, unsigned cap_mask,
                                                        int64_t* __restrict__ rep, int32_t* __restrict__ keep,
                                                        int32_t* __restrict__ keep_size) {
  __shared__ int s_cand;
  __shared__ int s_diff;
  const int p = blockIdx.x;
  const int lo = offs[p], size = offs[p + 1] - lo;
  if (threadIdx.x == 0) {
    const PropSig s = sig[p];
    const unsigned long long k = pr_key(s, size);
    unsigned slot = (unsigned)pp_mix64(k) & cap_mask;
    while (tkeys[slot] != k) slot = (slot + 1) & cap_mask;  // inserted by k_prop_insert
    int c = tmin[slot];
    if (c != p) {
      const PropSig t = sig[c];
      if (offs[c + 1] - offs[c] != size || t.h1 != s.h1 || t.h2 != s.h2) c = p;  // another list under the same key
    }
    s_cand = c;
    s_diff = 0;
  }
  __syncthreads();
  const int c = s_cand;
  if (c != p) {
    const int clo = offs[c];
    int diff = 0;
    for (int j = (int)threadIdx.x; j < size; j += PR_TPB) diff |= pts[lo + j] != pts[clo + j] ? 1 : 0;
    if (diff) s_diff = 1;  // (benign race: every writer stores 1)
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int r = (c != p && !s_diff) ? c : p;
    rep[p] = r;
    keep[p] = r == p ? 1 : 0;
    keep_size[p] = r == p ? size : 0;
  }
}

// pos_of[p] = position of p's representative among the kept proposals; kept proposals write their new offset
__global__ __launch_bounds__(PR_TPB) void k_prop_positions(const int64_t* __restrict__ rep, const int32_t* __restrict__ keep_pos,
                                                           const int32_t* __restrict__ size_pos, int64_t P, const int32_t* counts,
                                                           int64_t* __restrict__ pos_of, int32_t* __restrict__ uoffs) {
  const int64_t p = (int64_t)blockIdx.x * PR_TPB + threadIdx.x;
  if (p >= P) return;
  const int64_t r = rep[p];
  pos_of[p] = keep_pos[r];
  if (r == p) uoffs[keep_pos[p]] = size_pos[p];
  if (p == 0) uoffs[counts[0]] = counts[1];
}

extern "C" size_t pp_proposals_unique_workspace(int64_t n_prop) {
  const size_t P = (size_t)std::max<int64_t>(n_prop, 1);
  size_t cap = 64;
  while (cap < 2 * P) cap <<= 1;
  return pp_align(P * sizeof(PropSig)) + pp_align(cap * 8) + pp_align(cap * 4) + 4 * pp_align(P * 4) + 2 * pp_scan_workspace((int64_t)P) + 1024;
}

extern "C" int pp_proposals_unique(const int32_t* offsets, const int64_t* points, int64_t n_prop, int64_t n_points, int64_t* rep,
                                   int64_t* pos_of, int32_t* uniq_offsets, int32_t* counts, void* workspace, size_t workspace_bytes,
                                   pp_stream_t stream) {
  PP_REQUIRE(n_prop >= 0 && n_prop < (1ll << 30), "pp_proposals_unique: bad proposal count");
  PP_REQUIRE(counts && uniq_offsets, "pp_proposals_unique: null output");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(counts, 0, 3 * sizeof(int32_t), s));
  if (n_prop == 0) {
    PP_HIP(hipMemsetAsync(uniq_offsets, 0, sizeof(int32_t), s));
    return PP_OK;
  }
  PP_REQUIRE(offsets && points && rep && pos_of, "pp_proposals_unique: null pointer");
  if (workspace_bytes < pp_proposals_unique_workspace(n_prop)) return PP_ERR_WORKSPACE;
  size_t cap = 64;
  while (cap < 2 * (size_t)n_prop) cap <<= 1;
  PPArena ar(workspace, workspace_bytes);
  PropSig* sig = ar.take<PropSig>((size_t)n_prop);
  unsigned long long* tkeys = ar.take<unsigned long long>(cap);
  int32_t* tmin = ar.take<int32_t>(cap);
  int32_t* keep = ar.take<int32_t>((size_t)n_prop);
  int32_t* keep_size = ar.take<int32_t>((size_t)n_prop);
  int32_t* keep_pos = ar.take<int32_t>((size_t)n_prop);
  int32_t* size_pos = ar.take<int32_t>((size_t)n_prop);
  PP_REQUIRE(sig && tkeys && tmin && keep && keep_size && keep_pos && size_pos, "pp_proposals_unique: workspace");
  PP_HIP(hipMemsetAsync(tkeys, 0xFF, cap * 8, s));
  PP_HIP(hipMemsetAsync(tmin, 0x7F, cap * 4, s));  // 0x7F7F7F7F: above any proposal index
  const unsigned gp = pp_blocks(n_prop, PR_TPB);
  hipLaunchKernelGGL(k_prop_signature, dim3((unsigned)n_prop), dim3(PR_TPB), 0, s, offsets, points, n_points, sig, counts + 2);
  hipLaunchKernelGGL(k_prop_insert, dim3(gp), dim3(PR_TPB), 0, s, offsets, sig, n_prop, tkeys, tmin, (unsigned)(cap - 1));
  hipLaunchKernelGGL(k_prop_verify, dim3((unsigned)n_prop), dim3(PR_TPB), 0, s, offsets, points, sig, tkeys, tmin, (unsigned)(cap - 1),
                     rep, keep, keep_size);
  PP_LAUNCH_CHECK();
  int rc = pp_exclusive_scan_i32(keep, keep_pos, n_prop, counts, ar.cur(), ar.left(), s);  // counts[0] = kept proposals
  if (rc) return rc;
  rc = pp_exclusive_scan_i32(keep_size, size_pos, n_prop, counts + 1, ar.cur(), ar.left(), s);  // counts[1] = their entries
  if (rc) return rc;
  hipLaunchKernelGGL(k_prop_positions, dim3(gp), dim3(PR_TPB), 0, s, rep, keep_pos, size_pos, n_prop, counts, pos_of, uniq_offsets);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// one workgroup per proposal; the kept ones copy their list and write batch index and coordinate rows of the scorer's input
__global__ __launch_bounds__(PR_TPB) void k_prop_emit(const int32_t* __restrict__ offs, const int64_t* __restrict__ pts,
                                                      const int64_t* __restrict__ rep, const int64_t* __restrict__ pos_of,
                                                      const int32_t* __restrict__ uoffs, const int32_t* __restrict__ coords,
                                                      int64_t* __restrict__ out_pts, int64_t* __restrict__ out_batch,
                                                      int4* __restrict__ out_coords) {
  const int p = blockIdx.x;
  if (rep[p] != p) return;
  const int u = (int)pos_of[p];
  const int lo = offs[p], size = offs[p + 1] - lo, uo = uoffs[u];
  for (int j = (int)threadIdx.x; j < size; j += PR_TPB) {
    const int64_t v = pts[lo + j];
    out_pts[uo + j] = v;
    out_batch[uo + j] = u;
    if (out_coords) out_coords[uo + j] = make_int4(u, coords[3 * v], coords[3 * v + 1], coords[3 * v + 2]);
  }
}

extern "C" int pp_proposals_emit(const int32_t* offsets, const int64_t* points, int64_t n_prop, const int64_t* rep, const int64_t* pos_of,
                                 const int32_t* uniq_offsets, const int32_t* coords, int64_t* out_points, int64_t* out_batch,
                                 int32_t* out_coords4, pp_stream_t stream) {
  if (n_prop == 0) return PP_OK;
  PP_REQUIRE(offsets && points && rep && pos_of && uniq_offsets && out_points && out_batch, "pp_proposals_emit: null pointer");
  PP_REQUIRE((coords != nullptr) == (out_coords4 != nullptr), "pp_proposals_emit: coords and out_coords4 go together");
  hipLaunchKernelGGL(k_prop_emit, dim3((unsigned)n_prop), dim3(PR_TPB), 0, pp_s(stream), offsets, points, rep, pos_of, uniq_offsets, coords,
                     out_points, out_batch, (int4*)out_coords4);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
