// K8/K9 region growing, group-by-key, K10 instance IoU, K14 proposal intersections.
//
// region_grow follows SURVEY.md App. C (torch_points_kernels.region_grow; call sites
// torch_points3d/models/panoptic/PointGroup3heads.py:166-174,185-205,296-304,340-357) EXACTLY, including
// neighbour-list truncation at `nsample` (the ball query keeps the nsample LOWEST-index neighbours), but
// replaces the sequential host DFS by a parallel fixpoint: the sequential algorithm assigns every point v
// to the smallest-index point that can reach v through the directed neighbour-list graph (proof in
// DESIGN.md; cross-checked in tests/test_oracle.py::test_region_grow_min_index_ancestor_equivalence), so
//     L[v] = min(v, min_{u -> v} L[u])
// is iterated with atomicMin pushes + pointer jumping until nothing changes.
#include "pp_common.h"

// ---------------------------------------------------------------------------------------------
// selection of "thing" points
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rg_flag(const int64_t* __restrict__ labels, int64_t n,
                                                 const int64_t* __restrict__ ignore, int n_ignore, int num_classes,
                                                 int32_t* flag, int32_t* err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t l = labels[i];
  int keep = 1;
  for (int q = 0; q < n_ignore; ++q)
    if (ignore[q] == l) keep = 0;
  if (keep && (l < 0 || l >= num_classes)) {
    atomicAdd(err, 1);
    keep = 0;
  }
  flag[i] = keep;
}

__device__ inline uint64_t rg_cell_key(int bc, int cx, int cy, int cz) {
  // (batch,class) in 27 bits (exact below 2^27, i.e. 2^19 batch elements; folded beyond) + 3 x 12-bit wrapped cell
  // coordinates; bit 63 stays clear, so the key never equals the empty marker.  Equality of (batch,class) and of the cell
  // is re-checked on every candidate / query, so aliasing only sends the cells involved to the per-query kernel.
  const uint32_t b = (uint32_t)bc < (1u << 27) ? (uint32_t)bc : ((uint32_t)bc * 2654435761u) >> 5;
  return ((uint64_t)b << 36) | ((uint64_t)((uint32_t)cx & 0xFFFu) << 24) | ((uint64_t)((uint32_t)cy & 0xFFFu) << 12) |
         (uint64_t)((uint32_t)cz & 0xFFFu);
}

__global__ __launch_bounds__(256) void k_rg_compact(const float* __restrict__ pos, const int64_t* __restrict__ labels,
                                                    const int64_t* __restrict__ batch, int64_t n,
                                                    const int32_t* __restrict__ flag, const int32_t* __restrict__ rank,
                                                    float radius, int32_t* sel, int32_t* bc, uint64_t* keys,
                                                    int64_t cap, uint32_t* slot_of, int32_t* err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  int32_t a = rank[i];
  sel[a] = (int32_t)i;
  int64_t b = batch[i];
  if (b < 0 || b >= (1 << 23)) {
    atomicAdd(err, 1);
    b = 0;
  }
  int v = (int)((b << 8) | (labels[i] & 0xFF));
  bc[a] = v;
  float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  int cx = (int)floorf(x / radius), cy = (int)floorf(y / radius), cz = (int)floorf(z / radius);
  slot_of[a] = (uint32_t)pp_hash_insert_slot(keys, cap, rg_cell_key(v, cx, cy, cz));
}

__global__ __launch_bounds__(256) void k_fill_u64(uint64_t* p, uint64_t v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(256) void k_iota(int32_t* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}

// after the stable sort by slot: cell ranges + cell-sorted copies of the point data
__global__ __launch_bounds__(256) void k_rg_cells(const uint32_t* __restrict__ sorted_slot,
                                                  const int32_t* __restrict__ sorted_local, int64_t M,
                                                  const float* __restrict__ pos, const int32_t* __restrict__ sel,
                                                  const int32_t* __restrict__ bc, int32_t* cell_start,
                                                  int32_t* cell_end, float4* spos, int32_t* sbc) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  uint32_t s = sorted_slot[p];
  if (p == 0 || sorted_slot[p - 1] != s) cell_start[s] = (int32_t)p;
  if (p == M - 1 || sorted_slot[p + 1] != s) cell_end[s] = (int32_t)(p + 1);
  int32_t a = sorted_local[p];
  int64_t g = sel[a];
  spos[p] = make_float4(pos[3 * g], pos[3 * g + 1], pos[3 * g + 2], __int_as_float(a));
  sbc[p] = bc[a];
}

// ---------------------------------------------------------------------------------------------
// ball query.  Keeps the `nsample` smallest local indices among same-(batch,class) points with d^2 < r^2 (strict), as a
// SET (order inside the list is irrelevant to the result).
//
// Everything downstream of the cell sort is addressed by CELL-SORTED POSITION p: Lp[p] = label of the point at p (a local
// index: the smallest ancestor found so far), list[p * nsample + t] = position of the t-th neighbour -- a wave writes and
// reads one list with consecutive lanes, and the ~200 labels a list names live in the <= 27 cells around the point, i.e.
// in a few dozen cache lines instead of one line per entry (labels addressed by local index made the propagation's
// scattered label reads, 590 M of them on the bench scene, pull one L2 line each: 4.4 -> 1.1 ms for the first round).
// ---------------------------------------------------------------------------------------------
#define BQ_CAP 4096
struct BQScan {
  const float4* spos;
  const int32_t* sbc;
  int my_start, my_cnt;  // per-lane cell range (lanes 0..26)
  float qx, qy, qz, r2;
  int qbc;
};

// counts hits with local index <= T; optionally appends (index, position) to buf / bufp (LDS, capacity BQ_CAP) and/or the
// position of the first out_cap hits to out
__device__ inline int bq_scan(const BQScan& s, int lane, int T, int* buf, int* bufp, int32_t* out, int out_cap) {
  int total = 0;
  for (int c = 0; c < 27; ++c) {
    const int st = __shfl(s.my_start, c);
    const int cn = __shfl(s.my_cnt, c);
    for (int t0 = 0; t0 < cn; t0 += 64) {
      const int t = t0 + lane;
      bool hit = false;
      int idx = 0;
      if (t < cn) {
        float4 p = s.spos[st + t];
        float dx = s.qx - p.x, dy = s.qy - p.y, dz = s.qz - p.z;
        float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        idx = __float_as_int(p.w);
        hit = (d2 < s.r2) && (s.sbc[st + t] == s.qbc) && (idx <= T);
      }
      unsigned long long m = __ballot(hit);
      if (hit) {
        int pos = total + __popcll(m & ((1ull << lane) - 1ull));
        if (buf && pos < BQ_CAP) {
          buf[pos] = idx;
          bufp[pos] = st + t;
        }
        if (out && pos < out_cap) out[pos] = st + t;
      }
      total += __popcll(m);
    }
  }
  return total;
}

// Fallback: one wave per query, for the queries the cell kernel below could not serve (neighbourhood larger than its LDS
// buffer, or hash-aliased cells).  Hits are compacted into an LDS buffer; beyond nsample hits the nsample-th smallest
// index is found by bisection (in LDS up to BQ_CAP hits -- 4096: a query of the bench scene has up to 2300 --, by
// re-scanning the cells beyond).
__global__ __launch_bounds__(128) void k_ball_query(const float4* __restrict__ spos, const int32_t* __restrict__ sbc,
                                                    const uint64_t* __restrict__ keys,
                                                    const int32_t* __restrict__ cell_start,
                                                    const int32_t* __restrict__ cell_end, int64_t cap, int64_t M,
                                                    float radius, int nsample, const int32_t* __restrict__ fb_list,
                                                    const int32_t* __restrict__ fb_count, int32_t* __restrict__ list,
                                                    int32_t* __restrict__ deg) {
  __shared__ int lds[2][BQ_CAP];
  __shared__ int ldsp[2][BQ_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_fb = fb_count[0];
  int* buf = lds[wave];
  int* bufp = ldsp[wave];
  for (int64_t w = (int64_t)blockIdx.x * 2 + wave; w < n_fb; w += (int64_t)gridDim.x * 2) {
    const int64_t p = fb_list[w];
    const float4 q = spos[p];
    BQScan s;
    s.spos = spos; s.sbc = sbc; s.qx = q.x; s.qy = q.y; s.qz = q.z; s.r2 = radius * radius; s.qbc = sbc[p];
    s.my_start = 0; s.my_cnt = 0;
    if (lane < 27) {
      int cx = (int)floorf(q.x / radius) + (lane % 3 - 1);
      int cy = (int)floorf(q.y / radius) + ((lane / 3) % 3 - 1);
      int cz = (int)floorf(q.z / radius) + (lane / 9 - 1);
      int64_t slot = pp_hash_find_slot(keys, cap, rg_cell_key(s.qbc, cx, cy, cz));
      if (slot >= 0) {
        s.my_start = cell_start[slot];
        s.my_cnt = cell_end[slot] - s.my_start;
      }
    }
    int32_t* out = list + p * nsample;
    const int total = bq_scan(s, lane, 0x7FFFFFFF, buf, bufp, nullptr, 0);
    if (total <= nsample) {
      if (total <= BQ_CAP) {
        for (int t = lane; t < total; t += 64) out[t] = bufp[t];
      } else {
        bq_scan(s, lane, 0x7FFFFFFF, nullptr, nullptr, out, nsample);
      }
      if (lane == 0) deg[p] = total;
      continue;
    }
    // more than nsample hits: threshold T = nsample-th smallest local index (indices are distinct)
    int lo = 0, hi = (int)M - 1;
    if (total <= BQ_CAP) {
      while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int t = lane; t < total; t += 64) c += (buf[t] <= mid) ? 1 : 0;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        if (c >= nsample) hi = mid; else lo = mid + 1;
      }
      int wpos = 0;
      for (int t0 = 0; t0 < total; t0 += 64) {
        int t = t0 + lane;
        bool keep = t < total && buf[t] <= lo;
        unsigned long long m = __ballot(keep);
        if (keep) out[wpos + __popcll(m & ((1ull << lane) - 1ull))] = bufp[t];
        wpos += __popcll(m);
      }
    } else {
      while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        int c = bq_scan(s, lane, mid, nullptr, nullptr, nullptr, 0);
        if (c >= nsample) hi = mid; else lo = mid + 1;
      }
      bq_scan(s, lane, lo, nullptr, nullptr, out, nsample);
    }
    if (lane == 0) deg[p] = nsample;
  }
}

// Main ball query: one workgroup per occupied cell.
//   1. the local indices of the same-(batch,class) points of the 27 neighbouring cells are staged in LDS as 27 ascending
//      runs (50-bit keys: index | run | slot in the run);
//   2. a tree of pairwise merges (5 levels; every key finds its place with ONE binary search in the sibling run) sorts them
//      by index;
//   3. the candidates (x, y, z, position in the cell-sorted order) are gathered into LDS in that order (over the key
//      buffers);
//   4. every query of the cell is walked by ONE WAVE with a candidate per lane, 64 candidates per round in ascending
//      index order: a ballot appends the hits until the list holds nsample of them -- exactly the nsample smallest
//      indices inside the radius; all lanes work whatever the number of queries in the cell, a dense neighbourhood stops
//      early, list writes are coalesced.
// (The earlier form ranked every candidate against all 26 other runs and walked the list with a query per lane: a median
// cell holds 5 queries, so 7 % of the lanes had work in the walk.)
// Two instances: CAP = 1536 candidates (24 KiB of LDS, 6 workgroups per CU) serves every cell and lists the cells with more
// candidates; CAP = 8192 (128 KiB of dynamic LDS, one workgroup per CU) then serves those -- on shifted coordinates the
// points of an object collapse into a few cells with thousands of candidates and hundreds of queries each, and the
// per-query kernel pays a full scan plus a bisection over all hits for every one of them where this walk stops after
// nsample hits.
#define BQC_CAP 1536
#ifndef BQC_CAP_BIG
#define BQC_CAP_BIG 8192
#endif
template <int CAP, bool BIG>
__global__ __launch_bounds__(256) void k_ball_query_cells(const float4* __restrict__ spos, const int32_t* __restrict__ sbc,
                                                          const uint64_t* __restrict__ keys,
                                                          const int32_t* __restrict__ cell_start,
                                                          const int32_t* __restrict__ cell_end, int64_t cap, int64_t M,
                                                          float radius, int nsample,
                                                          const int32_t* __restrict__ cell_p0,
                                                          const int32_t* __restrict__ n_cells, int32_t* __restrict__ list,
                                                          int32_t* __restrict__ deg, int32_t* fb_list, int32_t* fb_count,
                                                          int32_t* big_list, int32_t* big_count) {
  // merge ping-pong; the sorted candidates (float4) overlay both halves
  extern __shared__ unsigned long long kdyn[];
  __shared__ unsigned long long kstat[BIG ? 1 : 2 * CAP];
  unsigned long long* const kbuf0 = BIG ? kdyn : kstat;
  unsigned long long* const kbuf1 = kbuf0 + CAP;
  __shared__ int nb_start[27], nb_cnt[27], nb_off[28];
  __shared__ int n_bad;
  float4* cand = (float4*)kbuf0;
  constexpr int SLOT_BITS = 13, IDX_SHIFT = SLOT_BITS + 5;  // key = index | run (5 bits) | slot in the run
  static_assert(CAP <= (1 << SLOT_BITS), "slot bits");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float r2 = radius * radius;
  const int ncell = n_cells[0];
  const int nwork = BIG ? big_count[0] : ncell;
  constexpr int PER = (CAP + 255) / 256;
  for (int w = blockIdx.x; w < nwork; w += gridDim.x) {
    const int c = BIG ? big_list[w] : w;
    const int p0 = cell_p0[c], p1 = c + 1 < ncell ? cell_p0[c + 1] : (int)M;
    const float4 q0 = spos[p0];
    const int bc0 = sbc[p0];
    const int ccx = (int)floorf(q0.x / radius), ccy = (int)floorf(q0.y / radius), ccz = (int)floorf(q0.z / radius);
    __syncthreads();  // previous cell's LDS contents are dead
    if (tid < 27) {
      int st = 0, cn = 0;
      int64_t slot = pp_hash_find_slot(keys, cap, rg_cell_key(bc0, ccx + (tid % 3 - 1), ccy + ((tid / 3) % 3 - 1), ccz + (tid / 9 - 1)));
      if (slot >= 0) {
        st = cell_start[slot];
        cn = cell_end[slot] - st;
      }
      nb_start[tid] = st;
      nb_cnt[tid] = cn;
    }
    if (tid == 0) n_bad = 0;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int k = 0; k < 27; ++k) {
        nb_off[k] = t;
        t += nb_cnt[k];
      }
      nb_off[27] = t;
    }
    __syncthreads();
    const int total = nb_off[27];
    if (!BIG && total > CAP) {  // too many candidates for this instance's buffer: the large one takes the cell
      if (tid == 0) big_list[atomicAdd(big_count, 1)] = c;
      continue;
    }
    bool fallback = total > CAP;
    if (!fallback) {
      int bad = 0;
      for (int f = tid; f < total; f += 256) {
        int lo = 0, hi = 27;  // run of slot f: last k with nb_off[k] <= f
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (nb_off[mid] <= f) lo = mid; else hi = mid;
        }
        const int slot = f - nb_off[lo];
        const int src = nb_start[lo] + slot;
        kbuf0[f] = ((unsigned long long)(unsigned)__float_as_int(spos[src].w) << IDX_SHIFT) |
                   ((unsigned long long)lo << SLOT_BITS) | (unsigned long long)slot;
        bad += sbc[src] != bc0 ? 1 : 0;  // another (batch, class) behind an aliased cell key
      }
      if (bad) atomicAdd(&n_bad, bad);
      __syncthreads();
      fallback = n_bad != 0;
    }
    if (fallback) {  // hand the whole cell to the per-query kernel
      for (int qq = p0 + tid; qq < p1; qq += 256) fb_list[atomicAdd(fb_count, 1)] = qq;
      continue;
    }
    // merge tree: at level l the runs are the groups of 2^l original runs; run i and run i ^ 1 merge
#pragma unroll 1
    for (int l = 0; l < 5; ++l) {
      const unsigned long long* src = (l & 1) ? kbuf1 : kbuf0;
      unsigned long long* dst = (l & 1) ? kbuf0 : kbuf1;
      for (int f = tid; f < total; f += 256) {
        const unsigned long long key = src[f];
        const int i = (int)((key >> SLOT_BITS) & 31u) >> l;
        const int a0 = nb_off[min(i << l, 27)];
        const int sib = i ^ 1;
        const int b0 = nb_off[min(sib << l, 27)], b1 = nb_off[min((sib + 1) << l, 27)];
        int x0 = b0, x1 = b1;  // first sibling element with a larger key (keys are distinct)
        while (x0 < x1) {
          const int mid = (x0 + x1) >> 1;
          if (src[mid] < key) x0 = mid + 1; else x1 = mid;
        }
        dst[min(a0, b0) + (f - a0) + (x0 - b0)] = key;
      }
      __syncthreads();
    }
    {  // sorted keys are in the second buffer; gather the candidates through registers, then overlay.  The loads are
       // unconditional (lanes past `total` re-read the cell's first point) and the components live in scalar arrays: a
       // conditionally written float4 array is not promoted to registers (it cost 96 B / 512 B of scratch per lane)
      float px[PER], py[PER], pz[PER];
      int ps[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = tid + 256 * u;
        const unsigned long long key = kbuf1[f < total ? f : 0];
        const int src = f < total ? nb_start[(int)((key >> SLOT_BITS) & 31u)] + (int)(key & ((1u << SLOT_BITS) - 1u)) : p0;
        const float4 p = spos[src];
        px[u] = p.x;
        py[u] = p.y;
        pz[u] = p.z;
        ps[u] = src;  // the walk needs the position, not the index: the order carries it
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = tid + 256 * u;
        if (f < total) cand[f] = make_float4(px[u], py[u], pz[u], __int_as_float(ps[u]));
      }
      __syncthreads();
    }
    for (int qq = p0 + wave; qq < p1; qq += 4) {
      const float4 q = spos[qq];
      // hash-aliased slot (another cell behind the same key): leave the query to the per-query kernel
      if ((int)floorf(q.x / radius) != ccx || (int)floorf(q.y / radius) != ccy || (int)floorf(q.z / radius) != ccz) {
        if (lane == 0) fb_list[atomicAdd(fb_count, 1)] = qq;
        continue;
      }
      int32_t* out = list + (int64_t)qq * nsample;
      int cnt = 0;
      for (int c0 = 0; c0 < total && cnt < nsample; c0 += 64) {
        const int t = c0 + lane;
        bool hit = false;
        int pt = 0;
        if (t < total) {
          const float4 pc = cand[t];
          const float dx = q.x - pc.x, dy = q.y - pc.y, dz = q.z - pc.z;
          hit = fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < r2;
          pt = __float_as_int(pc.w);
        }
        const unsigned long long m = __ballot(hit);
        if (hit) {
          const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
          if (pos < nsample) out[pos] = pt;
        }
        cnt += __popcll(m);
      }
      if (lane == 0) deg[qq] = min(cnt, nsample);
    }
  }
}

// run starts of the cell-sorted order -> list of occupied cells
__global__ __launch_bounds__(256) void k_rg_cell_flags(const uint32_t* __restrict__ sorted_slot, int64_t M, int32_t* flag) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < M) flag[p] = (p == 0 || sorted_slot[p - 1] != sorted_slot[p]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_rg_cell_list(const int32_t* __restrict__ flag, const int32_t* __restrict__ rank,
                                                      int64_t M, int32_t* cell_p0) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < M && flag[p]) cell_p0[rank[p]] = (int32_t)p;
}

// ---------------------------------------------------------------------------------------------
// label propagation.  A lane per point (cell-sorted position p) finds the point's current label by pointer jumping; the
// points that have a smaller label to push than last time form the frontier, and the wave walks the list of each of
// them with an entry per lane.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rg_init_labels(const float4* __restrict__ spos, int64_t M, int32_t* Lp, int32_t* pos_of,
                                                        int32_t* pushed) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int a = __float_as_int(spos[p].w);
  Lp[p] = a;  // every point starts as its own label ...
  pos_of[a] = (int32_t)p;
  pushed[p] = 0x7FFFFFFF;  // ... and has told nobody yet
}
// histogram add with one atomic per run of equal keys inside a wave (cell-sorted points carry long runs of the same
// cluster id; one atomic per element on a few hot addresses serialises in L2)
__device__ __forceinline__ void rg_hist_add_runs(int32_t* hist, int k, bool valid) {
  const int lane = threadIdx.x & 63;
  const int kk = valid ? k : -1;
  const int prev = __shfl_up(kk, 1);
  const bool lead = lane == 0 || prev != kk;
  const unsigned long long lm = __ballot(lead);
  if (lead && kk >= 0) {
    const unsigned long long higher = lane < 63 ? (lm >> (lane + 1)) : 0ull;
    const int next = higher ? lane + 1 + __builtin_ctzll(higher) : 64;
    atomicAdd(&hist[kk], next - lane);  // lanes past the end of the data carry kk = -1 and start their own run
  }
}

// back to local-index space for the cluster bookkeeping, and the cluster sizes (points per root label): counted in the
// cell-sorted order, where the points of a cell -- mostly one cluster -- are consecutive, so the runs are long
__global__ __launch_bounds__(256) void k_rg_labels_by_index(const float4* __restrict__ spos, const int32_t* __restrict__ Lp, int64_t M,
                                                            int32_t* L, int32_t* size) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int l = p < M ? Lp[p] : -1;
  if (p < M) L[__float_as_int(spos[p].w)] = l;
  rg_hist_add_runs(size, l, p < M);
}

__global__ __launch_bounds__(256) void k_rg_propagate(const int32_t* __restrict__ list, const int32_t* __restrict__ deg,
                                                      const int32_t* __restrict__ pos_of, int32_t* Lp, int32_t* pushed,
                                                      int64_t M, int nsample, int32_t* changed) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  volatile int32_t* VL = Lp;
  bool ch = false, need = false;
  int lk = 0, d = 0;
  if (p < M) {
    const int lk0 = VL[p];
    lk = lk0;
    for (;;) {  // pointer jumping: a label is the index of an ancestor, and that ancestor's label one of its ancestors
      int pnt = VL[pos_of[lk]];
      if (pnt >= lk) break;
      lk = pnt;
    }
    if (lk < lk0) {
      atomicMin(&Lp[p], lk);
      ch = true;
    }
    // frontier: a point re-walks its neighbour list only when it has a smaller label to push than last time
    // (labels only decrease, so a label already pushed can never be needed again by the same neighbours)
    if (pushed[p] > lk) {
      pushed[p] = lk;
      need = true;
      d = deg[p];
    }
  }
  unsigned long long todo = __ballot(need);
  const int64_t p_wave = p - lane;
  while (todo) {
    const int b = __builtin_ctzll(todo);
    todo &= todo - 1;
    const int lkb = __shfl(lk, b), db = __shfl(d, b);
    const int32_t* row = list + (p_wave + b) * nsample;
    // 256 entries per pass: all list loads, then all label loads, then the atomics (no value read back) -- the wave has
    // 4 + 4 loads in flight instead of one dependent pair per 64 entries
    for (int t0 = 0; t0 < db; t0 += 256) {
      int j[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + 64 * u + lane;
        j[u] = t < db ? row[t] : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = j[u] >= 0 ? VL[j[u]] : -1;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (v[u] > lkb) {
          (void)__hip_atomic_fetch_min(&Lp[j[u]], lkb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ch = true;  // (another push may have got there first: an over-report costs at most one extra round)
        }
    }
  }
  if (ch) changed[0] = 1;
}

__global__ __launch_bounds__(256) void k_rg_rootkeys(const int32_t* __restrict__ L, const int32_t* __restrict__ size,
                                                     const int32_t* __restrict__ bc, int64_t M, int min_size,
                                                     uint32_t* key, int32_t* n_clusters) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= M) return;
  bool root = (L[v] == (int32_t)v) && size[v] >= min_size;
  key[v] = root ? (uint32_t)(bc[v] & 0xFF) : 256u;
  if (root) atomicAdd(n_clusters, 1);
}
__global__ __launch_bounds__(256) void k_rg_root_rank(const uint32_t* __restrict__ sorted_key,
                                                      const int32_t* __restrict__ sorted_root, int64_t M,
                                                      int32_t* cluster_of_root) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  if (sorted_key[p] < 256u) cluster_of_root[sorted_root[p]] = (int32_t)p;
}
__global__ __launch_bounds__(256) void k_rg_point_cluster(const int32_t* __restrict__ L,
                                                          const int32_t* __restrict__ cluster_of_root,
                                                          const int32_t* __restrict__ sel, int64_t M, int32_t* pc_local,
                                                          int32_t* point_cluster, int64_t* sel64) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= M) return;
  int32_t c = cluster_of_root[L[v]];
  pc_local[v] = c;
  point_cluster[sel[v]] = c;
  sel64[v] = sel[v];
}
__global__ __launch_bounds__(256) void k_fill_i32(int32_t* p, int32_t v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

static inline unsigned fill_blocks(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }
static inline int bits_for(int64_t v) {
  int b = 1;
  while ((1ll << b) <= v) ++b;
  return b;
}

// ---------------------------------------------------------------------------------------------
// group-by-key
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gbk_keys(const int32_t* __restrict__ key, int64_t n, int n_groups,
                                                  uint32_t* ukey, int32_t* err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (i < n) {
    int32_t k = key[i];
    bad = k >= n_groups;
    ukey[i] = (k < 0 || bad) ? (uint32_t)n_groups : (uint32_t)k;
  }
  const unsigned long long m = __ballot(bad);
  if (m && (threadIdx.x & 63) == 0) atomicAdd(err, __popcll(m));
}
// offsets from the SORTED keys: offsets[g] = number of keys below g, one binary search per group (dropped keys carry
// n_groups and sort last, so offsets[n_groups] = number of kept elements).  No histogram: one atomic per run of equal
// keys still serialised on the few hot addresses of large clusters (0.72 ms for 6 M points of the bench scene).
__global__ __launch_bounds__(256) void k_gbk_bounds(const uint32_t* __restrict__ sorted, int64_t n, int n_groups,
                                                    int32_t* __restrict__ offsets) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g > n_groups) return;
  int64_t lo = 0, hi = n;  // first position with a key >= g
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted[mid] < (uint32_t)g) lo = mid + 1; else hi = mid;
  }
  offsets[g] = (int32_t)lo;
}
__global__ __launch_bounds__(256) void k_gbk_emit(const int32_t* __restrict__ sorted_idx, const int64_t* __restrict__ ids,
                                                  const int32_t* __restrict__ total, int64_t n, int64_t* out) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || p >= total[0]) return;
  int32_t i = sorted_idx[p];
  out[p] = ids ? ids[i] : (int64_t)i;
}

extern "C" size_t pp_group_by_key_workspace(int64_t n) {
  size_t m = (size_t)std::max<int64_t>(n, 1);
  return 4 * pp_align(m * 4) + pp_align((m + 2) * 4) + pp_sort_pairs_workspace(n) + pp_scan_workspace(n + 2) + 4096;
}

extern "C" int pp_group_by_key(const int32_t* key, const int64_t* ids, int64_t n, int32_t n_groups, int32_t* offsets,
                               int64_t* out, int32_t* total, int32_t* n_out_of_range, void* workspace,
                               size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(offsets && total && n_groups >= 0, "pp_group_by_key: bad arguments");
  PP_REQUIRE(n_groups <= n + 1 || n_groups < (1 << 30), "pp_group_by_key: n_groups too large");
  if (workspace_bytes < pp_group_by_key_workspace(std::max<int64_t>(n, n_groups))) return PP_ERR_WORKSPACE;
  hipStream_t s = pp_s(stream);
  PPArena ar(workspace, workspace_bytes);
  size_t m = (size_t)std::max<int64_t>(n, 1);
  uint32_t* ukey = ar.take<uint32_t>(m);
  uint32_t* ukey2 = ar.take<uint32_t>(m);
  int32_t* idx = ar.take<int32_t>(m);
  int32_t* idx2 = ar.take<int32_t>(m);
  int32_t* err = ar.take<int32_t>(1);
  PP_HIP(hipMemsetAsync(err, 0, sizeof(int32_t), s));
  if (n > 0) {
    hipLaunchKernelGGL(k_gbk_keys, dim3(pp_blocks(n, 256)), dim3(256), 0, s, key, n, n_groups, ukey, err);
    hipLaunchKernelGGL(k_iota, dim3(pp_blocks(n, 256)), dim3(256), 0, s, idx, n);
    PP_LAUNCH_CHECK();
    int rc = pp_sort_pairs_u32(ukey, ukey2, idx, idx2, n, bits_for(n_groups), ar.cur(), ar.left(), s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_gbk_bounds, dim3(pp_blocks((int64_t)n_groups + 1, 256)), dim3(256), 0, s, ukey2, n, n_groups, offsets);
  PP_LAUNCH_CHECK();
  PP_HIP(hipMemcpyAsync(total, offsets + n_groups, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  // keys >= n_groups are dropped like negative ones, but they are a caller error: report their number
  if (n_out_of_range) PP_HIP(hipMemcpyAsync(n_out_of_range, err, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (n > 0 && out) {
    hipLaunchKernelGGL(k_gbk_emit, dim3(pp_blocks(n, 256)), dim3(256), 0, s, idx2, ids, total, n, out);
    PP_LAUNCH_CHECK();
  }
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// region grow driver
// ---------------------------------------------------------------------------------------------
// n = all points, m = points that are not ignored (the neighbour lists, the big part, scale with m only)
extern "C" size_t pp_region_grow_workspace_for(int64_t n, int64_t m_sel, int32_t nsample) {
  size_t m = (size_t)std::max<int64_t>(n, 1);
  size_t ms = (size_t)std::max<int64_t>(std::min<int64_t>(m_sel, n), 1);
  size_t cap = (size_t)pp_hash_capacity((int64_t)ms);
  size_t b = 0;
  b += 16 * pp_align(m * 4);            // flag, rank, sel, bc, slot_of, sorted_slot, local, sorted_local, sbc, deg, L, size, keys32 x2, roots x2
  b += pp_align(m * 8);                 // sel64
  b += pp_align(m * 16);                // spos
  b += pp_align(cap * 8) + 2 * pp_align(cap * 4);  // cell hash
  b += pp_align(ms * (size_t)nsample * 4);         // neighbour lists
  b += pp_sort_pairs_workspace(n) + pp_scan_workspace(n) + pp_group_by_key_workspace(n) + 8192;
  return b;
}
extern "C" size_t pp_region_grow_workspace(int64_t n, int32_t nsample) { return pp_region_grow_workspace_for(n, n, nsample); }

extern "C" int pp_region_grow(const float* pos, const int64_t* labels, const int64_t* batch, int64_t n,
                              const int64_t* ignore_labels, int32_t n_ignore, int32_t num_classes, int32_t nsample,
                              float radius, int32_t min_cluster_size, int32_t* point_cluster,
                              int32_t* cluster_offsets, int64_t* cluster_points, int32_t* counts, void* workspace,
                              size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(point_cluster && cluster_offsets && cluster_points && counts, "pp_region_grow: null output");
  PP_REQUIRE(nsample >= 1 && radius > 0.f && num_classes >= 1 && num_classes <= 256, "pp_region_grow: bad parameters");
  PP_REQUIRE(n < (1ll << 31), "pp_region_grow: n too large");
  if (workspace_bytes < pp_region_grow_workspace_for(n, 0, nsample)) return PP_ERR_WORKSPACE;
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), s));
  PP_HIP(hipMemsetAsync(cluster_offsets, 0, sizeof(int32_t), s));
  if (n == 0) return PP_OK;
  PPArena ar(workspace, workspace_bytes);
  size_t m = (size_t)n;
  int32_t* flag = ar.take<int32_t>(m);
  int32_t* rank = ar.take<int32_t>(m);
  int32_t* sel = ar.take<int32_t>(m);
  int32_t* bc = ar.take<int32_t>(m);
  uint32_t* slot_of = ar.take<uint32_t>(m);
  uint32_t* sorted_slot = ar.take<uint32_t>(m);
  int32_t* local = ar.take<int32_t>(m);
  int32_t* sorted_local = ar.take<int32_t>(m);
  int32_t* sbc = ar.take<int32_t>(m);
  int32_t* deg = ar.take<int32_t>(m);
  int32_t* L = ar.take<int32_t>(m);
  int32_t* size = ar.take<int32_t>(m);
  uint32_t* rkey = ar.take<uint32_t>(m);
  uint32_t* rkey2 = ar.take<uint32_t>(m);
  int32_t* roots2 = ar.take<int32_t>(m);
  int32_t* misc = ar.take<int32_t>(64);  // [0]=M [1]=err [2]=changed [3]=n_clusters
  int64_t* sel64 = ar.take<int64_t>(m);
  float4* spos = ar.take<float4>(m);
  PP_REQUIRE(flag && spos && misc, "pp_region_grow: workspace carve failed");
  PP_HIP(hipMemsetAsync(misc, 0, 64 * sizeof(int32_t), s));
  unsigned nb = pp_blocks(n, 256);
  hipLaunchKernelGGL(k_fill_i32, dim3(fill_blocks(n)), dim3(256), 0, s, point_cluster, -1, n);
  hipLaunchKernelGGL(k_rg_flag, dim3(nb), dim3(256), 0, s, labels, n, ignore_labels, n_ignore, num_classes, flag,
                     misc + 1);
  PP_LAUNCH_CHECK();
  int rc = pp_exclusive_scan_i32(flag, rank, n, misc, ar.cur(), ar.left(), s);
  if (rc) return rc;
  int32_t h[4];
  PP_HIP(hipMemcpyAsync(h, misc, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PP_HIP(hipStreamSynchronize(s));
  if (h[1]) {
    pp_set_error("pp_region_grow: %d labels outside [0,num_classes=%d)", h[1], num_classes);
    return PP_ERR_INVALID;
  }
  const int64_t M = h[0];
  if (M == 0) return PP_OK;
  if (workspace_bytes < pp_region_grow_workspace_for(n, M, nsample)) {
    pp_set_error("pp_region_grow: workspace too small for %lld selected points (see pp_region_grow_workspace_for)", (long long)M);
    return PP_ERR_WORKSPACE;
  }
  const int64_t cap = pp_hash_capacity(M);
  uint64_t* ckeys = ar.take<uint64_t>((size_t)cap);
  int32_t* cell_start = ar.take<int32_t>((size_t)cap);
  int32_t* cell_end = ar.take<int32_t>((size_t)cap);
  int32_t* list = ar.take<int32_t>((size_t)M * (size_t)nsample);
  PP_REQUIRE(ckeys && cell_start && cell_end && list, "pp_region_grow: workspace carve failed (lists)");
  unsigned mb = pp_blocks(M, 256);
  hipLaunchKernelGGL(k_fill_u64, dim3(fill_blocks(cap)), dim3(256), 0, s, ckeys, PP_EMPTY_KEY, cap);
  hipLaunchKernelGGL(k_rg_compact, dim3(nb), dim3(256), 0, s, pos, labels, batch, n, flag, rank, radius, sel, bc, ckeys,
                     cap, slot_of, misc + 1);
  hipLaunchKernelGGL(k_iota, dim3(mb), dim3(256), 0, s, local, M);
  PP_LAUNCH_CHECK();
  rc = pp_sort_pairs_u32(slot_of, sorted_slot, local, sorted_local, M, bits_for(cap - 1), ar.cur(), ar.left(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_rg_cells, dim3(mb), dim3(256), 0, s, sorted_slot, sorted_local, M, pos, sel, bc, cell_start,
                     cell_end, spos, sbc);
  // occupied cells = runs of equal slot in the sorted order (flag / rank / slot_of / local are free by now)
  int32_t* cell_p0 = (int32_t*)slot_of;
  int32_t* fb_list = (int32_t*)rkey2;  // free until the root sort below
  hipLaunchKernelGGL(k_rg_cell_flags, dim3(mb), dim3(256), 0, s, sorted_slot, M, flag);
  PP_LAUNCH_CHECK();
  rc = pp_exclusive_scan_i32(flag, rank, M, misc + 4, ar.cur(), ar.left(), s);  // misc[4] = number of cells
  if (rc) return rc;
  hipLaunchKernelGGL(k_rg_cell_list, dim3(mb), dim3(256), 0, s, flag, rank, M, cell_p0);
  int32_t* pushed = size;  // reused as the cluster-size array once the fixpoint is reached
  int32_t* Lp = flag;      // labels by cell-sorted position (flag / rank are free from here on)
  int32_t* pos_of = rank;  // cell-sorted position of a local index
  hipLaunchKernelGGL(k_rg_init_labels, dim3(mb), dim3(256), 0, s, spos, M, Lp, pos_of, pushed);
  int32_t* big_list = (int32_t*)rkey;  // free until the root keys below
  hipLaunchKernelGGL((k_ball_query_cells<BQC_CAP, false>), dim3((unsigned)std::min<int64_t>(M, 8192)), dim3(256), 0, s, spos,
                     sbc, ckeys, cell_start, cell_end, cap, M, radius, nsample, cell_p0, misc + 4, list, deg, fb_list,
                     misc + 5, big_list, misc + 6);
  {
    constexpr size_t big_lds = 2 * (size_t)BQC_CAP_BIG * sizeof(unsigned long long);
    // > 64 KiB of LDS per workgroup has to be asked for (per device; a host-side call)
    PP_HIP(hipFuncSetAttribute((const void*)k_ball_query_cells<BQC_CAP_BIG, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)big_lds));
    hipLaunchKernelGGL((k_ball_query_cells<BQC_CAP_BIG, true>), dim3((unsigned)std::min<int64_t>(M, 1024)), dim3(256), big_lds,
                       s, spos, sbc, ckeys, cell_start, cell_end, cap, M, radius, nsample, cell_p0, misc + 4, list, deg,
                       fb_list, misc + 5, big_list, misc + 6);
  }
  hipLaunchKernelGGL(k_ball_query, dim3((unsigned)std::min<int64_t>(pp_blocks(M, 2), 4096)), dim3(128), 0, s, spos, sbc,
                     ckeys, cell_start, cell_end, cap, M, radius, nsample, fb_list, misc + 5, list, deg);
  PP_LAUNCH_CHECK();
  // fixpoint
  bool converged = false;
  for (int round = 0; round < 4096 && !converged; ++round) {
    PP_HIP(hipMemsetAsync(misc + 2, 0, sizeof(int32_t), s));
    for (int it = 0; it < 4; ++it)
      hipLaunchKernelGGL(k_rg_propagate, dim3(mb), dim3(256), 0, s, list, deg, pos_of, Lp, pushed, M, nsample, misc + 2);
    PP_LAUNCH_CHECK();
    PP_HIP(hipMemcpyAsync(h, misc + 1, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PP_HIP(hipStreamSynchronize(s));
    if (h[0]) {
      pp_set_error("pp_region_grow: %d batch ids outside [0,2^23)", h[0]);
      return PP_ERR_RANGE;
    }
    converged = !h[1];
  }
  if (!converged) {  // never continue on labels that are not the fixpoint: the clusters would be silently wrong
    pp_set_error("pp_region_grow: label propagation did not converge in 4096 x 4 rounds (%lld selected points)", (long long)M);
    return PP_ERR_INVALID;
  }
  // clusters: valid roots ordered by (class, root index)
  PP_HIP(hipMemsetAsync(size, 0, sizeof(int32_t) * (size_t)M, s));
  hipLaunchKernelGGL(k_rg_labels_by_index, dim3(mb), dim3(256), 0, s, spos, Lp, M, L, size);
  hipLaunchKernelGGL(k_rg_rootkeys, dim3(mb), dim3(256), 0, s, L, size, bc, M, min_cluster_size, rkey, misc + 3);
  PP_LAUNCH_CHECK();
  rc = pp_sort_pairs_u32(rkey, rkey2, local, roots2, M, 9, ar.cur(), ar.left(), s);
  if (rc) return rc;
  int32_t* cluster_of_root = sorted_local;  // reuse
  int32_t* pc_local = deg;                  // reuse
  hipLaunchKernelGGL(k_fill_i32, dim3(fill_blocks(M)), dim3(256), 0, s, cluster_of_root, -1, M);
  hipLaunchKernelGGL(k_rg_root_rank, dim3(mb), dim3(256), 0, s, rkey2, roots2, M, cluster_of_root);
  hipLaunchKernelGGL(k_rg_point_cluster, dim3(mb), dim3(256), 0, s, L, cluster_of_root, sel, M, pc_local, point_cluster,
                     sel64);
  PP_LAUNCH_CHECK();
  PP_HIP(hipMemcpyAsync(h, misc + 3, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PP_HIP(hipStreamSynchronize(s));
  const int32_t nC = h[0];
  PP_HIP(hipMemcpyAsync(counts, misc + 3, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  rc = pp_group_by_key(pc_local, sel64, M, nC, cluster_offsets, cluster_points, counts + 1, nullptr, ar.cur(), ar.left(),
                        stream);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// K10 instance IoU
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_iou_accum(const int32_t* __restrict__ prop_offsets,
                                                   const int64_t* __restrict__ prop_points, int n_prop,
                                                   const int64_t* __restrict__ gt, const int64_t* __restrict__ batch,
                                                   const int32_t* __restrict__ gt_offsets, int total_gt, float* iou) {
  // one wave per proposal
  const int lane = threadIdx.x & 63;
  const int p = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (p >= n_prop) return;
  const int lo = prop_offsets[p], hi = prop_offsets[p + 1];
  if (hi <= lo) return;
  const int64_t b = batch[prop_points[lo]];
  const int g0 = gt_offsets[b], g1 = gt_offsets[b + 1];
  for (int t = lo + lane; t < hi; t += 64) {
    int64_t gi = gt[prop_points[t]];
    if (gi >= 1 && gi <= g1 - g0) atomicAdd(&iou[(int64_t)p * total_gt + g0 + (int)gi - 1], 1.0f);
  }
}
__global__ __launch_bounds__(256) void k_iou_finish(const int32_t* __restrict__ prop_offsets,
                                                    const int32_t* __restrict__ gt_sizes, int n_prop, int total_gt,
                                                    float* iou) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n_prop * total_gt) return;
  float v = iou[e];
  if (v > 0.f) {
    int p = (int)(e / total_gt), g = (int)(e % total_gt);
    float sz = (float)(prop_offsets[p + 1] - prop_offsets[p]);
    iou[e] = v / (sz + (float)gt_sizes[g] - v);
  }
}
extern "C" int pp_instance_iou(const int32_t* prop_offsets, const int64_t* prop_points, int32_t n_prop,
                               const int64_t* gt_instances, const int64_t* batch, const int32_t* gt_offsets,
                               const int32_t* gt_sizes, int32_t total_gt, float* iou, pp_stream_t stream) {
  if (n_prop == 0 || total_gt == 0) return PP_OK;
  PP_REQUIRE(prop_offsets && prop_points && gt_instances && batch && gt_offsets && gt_sizes && iou,
             "pp_instance_iou: null pointer");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(iou, 0, sizeof(float) * (size_t)n_prop * total_gt, s));
  hipLaunchKernelGGL(k_iou_accum, dim3(pp_blocks((int64_t)n_prop * 64, 256)), dim3(256), 0, s, prop_offsets, prop_points,
                     n_prop, gt_instances, batch, gt_offsets, total_gt, iou);
  hipLaunchKernelGGL(k_iou_finish, dim3(pp_blocks((int64_t)n_prop * total_gt, 256)), dim3(256), 0, s, prop_offsets,
                     gt_sizes, n_prop, total_gt, iou);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// K14 proposal x proposal intersections via the point -> proposal incidence (sorted by point)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pi_keys(const int32_t* __restrict__ prop_offsets,
                                                 const int64_t* __restrict__ prop_points, int n_prop, uint32_t* key,
                                                 int32_t* val) {
  const int lane = threadIdx.x & 63;
  const int p = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (p >= n_prop) return;
  for (int t = prop_offsets[p] + lane; t < prop_offsets[p + 1]; t += 64) {
    key[t] = (uint32_t)prop_points[t];
    val[t] = p;
  }
}
__global__ __launch_bounds__(256) void k_pi_pairs(const uint32_t* __restrict__ key, const int32_t* __restrict__ val,
                                                  int64_t total, int n_prop, int32_t* inter) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const uint32_t k = key[e];
  const int a = val[e];
  for (int64_t f = e; f >= 0 && key[f] == k; --f) atomicAdd(&inter[(int64_t)a * n_prop + val[f]], 1);
  for (int64_t f = e + 1; f < total && key[f] == k; ++f) atomicAdd(&inter[(int64_t)a * n_prop + val[f]], 1);
}
extern "C" size_t pp_proposal_intersections_workspace(int64_t total_points, int64_t n_points) {
  (void)n_points;
  size_t m = (size_t)std::max<int64_t>(total_points, 1);
  return 4 * pp_align(m * 4) + pp_sort_pairs_workspace(total_points) + 4096;
}
extern "C" int pp_proposal_intersections(const int32_t* prop_offsets, const int64_t* prop_points, int32_t n_prop,
                                         int64_t n_points, int32_t* inter, void* workspace, size_t workspace_bytes,
                                         pp_stream_t stream) {
  if (n_prop == 0) return PP_OK;
  PP_REQUIRE(prop_offsets && prop_points && inter, "pp_proposal_intersections: null pointer");
  PP_REQUIRE(n_points < (1ll << 32), "pp_proposal_intersections: n_points too large");
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(inter, 0, sizeof(int32_t) * (size_t)n_prop * n_prop, s));
  int32_t total = 0;
  PP_HIP(hipMemcpyAsync(&total, prop_offsets + n_prop, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PP_HIP(hipStreamSynchronize(s));
  if (total == 0) return PP_OK;
  if (workspace_bytes < pp_proposal_intersections_workspace(total, n_points)) return PP_ERR_WORKSPACE;
  PPArena ar(workspace, workspace_bytes);
  uint32_t* key = ar.take<uint32_t>((size_t)total);
  uint32_t* key2 = ar.take<uint32_t>((size_t)total);
  int32_t* val = ar.take<int32_t>((size_t)total);
  int32_t* val2 = ar.take<int32_t>((size_t)total);
  hipLaunchKernelGGL(k_pi_keys, dim3(pp_blocks((int64_t)n_prop * 64, 256)), dim3(256), 0, s, prop_offsets, prop_points,
                     n_prop, key, val);
  PP_LAUNCH_CHECK();
  int rc = pp_sort_pairs_u32(key, key2, val, val2, total, bits_for(std::max<int64_t>(n_points, 1)), ar.cur(), ar.left(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_pi_pairs, dim3(pp_blocks(total, 256)), dim3(256), 0, s, key2, val2, (int64_t)total, n_prop, inter);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
