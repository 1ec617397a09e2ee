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
  // (batch,class) folded to 16 bits + 3 x 16-bit wrapped cell coordinates; equality of (batch,class) is
  // re-checked on every candidate, so aliasing only adds rejected candidates.
  return ((uint64_t)((uint32_t)bc * 2654435761u >> 16) << 48) | ((uint64_t)(uint16_t)cx << 32) |
         ((uint64_t)(uint16_t)cy << 16) | (uint64_t)(uint16_t)cz;
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
// ball query: one wave per query point (cell-sorted order -> neighbouring waves share cells in L1/L2).
// Keeps the `nsample` smallest local indices among same-(batch,class) points with d^2 < r^2 (strict),
// as a SET (order inside the list is irrelevant to the result).
// ---------------------------------------------------------------------------------------------
#define BQ_CAP 2048
struct BQScan {
  const float4* spos;
  const int32_t* sbc;
  int my_start, my_cnt;  // per-lane cell range (lanes 0..26)
  float qx, qy, qz, r2;
  int qbc;
};

// counts hits with local index <= T; optionally appends them to buf (LDS, capacity BQ_CAP) and/or out
__device__ inline int bq_scan(const BQScan& s, int lane, int T, int* buf, int32_t* out, int out_cap, int64_t out_stride) {
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
        if (buf && pos < BQ_CAP) buf[pos] = idx;
        if (out && pos < out_cap) out[(int64_t)pos * out_stride] = idx;
      }
      total += __popcll(m);
    }
  }
  return total;
}

// Neighbour lists are stored "t-major" in cell-sorted point order: list[t * M + p] = t-th neighbour (a local index) of
// the point at cell-sorted position p, so that the lanes of a wave (consecutive p) read and write consecutive words.
//
// Fallback: one wave per query, for the queries the cell kernel below could not serve (neighbourhood larger than its LDS
// buffer, or hash-aliased cells).  Hits are compacted into an LDS buffer; beyond nsample hits the nsample-th smallest
// index is found by bisection.
__global__ __launch_bounds__(256) void k_ball_query(const float4* __restrict__ spos, const int32_t* __restrict__ sbc,
                                                    const uint64_t* __restrict__ keys,
                                                    const int32_t* __restrict__ cell_start,
                                                    const int32_t* __restrict__ cell_end, int64_t cap, int64_t M,
                                                    float radius, int nsample, const int32_t* __restrict__ fb_list,
                                                    const int32_t* __restrict__ fb_count, int32_t* __restrict__ list,
                                                    int32_t* __restrict__ deg) {
  __shared__ int lds[4][BQ_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_fb = fb_count[0];
  int* buf = lds[wave];
  for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < n_fb; w += (int64_t)gridDim.x * 4) {
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
  int32_t* out = list + p;  // column p, stride M
  const int total = bq_scan(s, lane, 0x7FFFFFFF, buf, nullptr, 0, 0);
  if (total <= nsample) {
    if (total <= BQ_CAP) {
      for (int t = lane; t < total; t += 64) out[(int64_t)t * M] = buf[t];
    } else {
      bq_scan(s, lane, 0x7FFFFFFF, nullptr, out, nsample, M);
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
      if (keep) out[(int64_t)(wpos + __popcll(m & ((1ull << lane) - 1ull))) * M] = buf[t];
      wpos += __popcll(m);
    }
  } else {
    while (lo < hi) {
      int mid = lo + ((hi - lo) >> 1);
      int c = bq_scan(s, lane, mid, nullptr, nullptr, 0, 0);
      if (c >= nsample) hi = mid; else lo = mid + 1;
    }
    bq_scan(s, lane, lo, nullptr, out, nsample, M);
  }
  if (lane == 0) deg[p] = nsample;
  }
}

// Main ball query: one workgroup per occupied cell.  The same-(batch,class) points of the 27 neighbouring cells are
// staged in LDS once, sorted by local index (bitonic), and every query of the cell (one lane each) walks them in
// ascending index order -- broadcast LDS reads, no global traffic -- appending hits until it has nsample of them:
// exactly the nsample smallest indices inside the radius, and a dense neighbourhood stops early.
#ifndef BQC_CAP
#define BQC_CAP 1536  // measured optimum: 768 / 1024 / 1536 / 2048 / 2560 -> region growing 29.6 / 25.8 / 19.3 / 23.6 / 35.6 ms
                      // (16.4 / 19.2 / 19.1 ms for 1536 / 2048 / 2560 once only the keys are staged in LDS)
#endif
#ifndef BQC_ROUND
#define BQC_ROUND 8  // candidates per round of the query walk (by PMC the kernel waits on LDS 3/4 of the time)
#endif
__global__ __launch_bounds__(256) void k_ball_query_cells(const float4* __restrict__ spos, const int32_t* __restrict__ sbc,
                                                          const uint64_t* __restrict__ keys,
                                                          const int32_t* __restrict__ cell_start,
                                                          const int32_t* __restrict__ cell_end, int64_t cap, int64_t M,
                                                          float radius, int nsample,
                                                          const int32_t* __restrict__ cell_p0,
                                                          const int32_t* __restrict__ n_cells, int32_t* __restrict__ list,
                                                          int32_t* __restrict__ deg, int32_t* fb_list, int32_t* fb_count) {
  __shared__ int rawkey[BQC_CAP];   // local indices of the candidates as loaded: 27 runs, each ascending
  __shared__ float4 cand[BQC_CAP];  // merged candidates: ascending in local index (.w)
  __shared__ int nb_start[27], nb_cnt[27], nb_off[28];
  __shared__ int n_bad;
  const int tid = threadIdx.x;
  const float r2 = radius * radius;
  const int ncell = n_cells[0];
  for (int c = blockIdx.x; c < ncell; c += gridDim.x) {
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
    bool fallback = total > BQC_CAP;  // too many candidates for the buffer
    if (!fallback) {
      // flat over all candidates: every lane busy, a few rounds instead of 27
      int bad = 0;
      for (int f = tid; f < total; f += 256) {
        int lo = 0, hi = 27;  // run of slot f: last k with nb_off[k] <= f
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (nb_off[mid] <= f) lo = mid; else hi = mid;
        }
        const int src = nb_start[lo] + (f - nb_off[lo]);
        rawkey[f] = __float_as_int(spos[src].w);  // keys only: 6 KB instead of 24 KB of LDS (5 workgroups per CU, not 3)
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
    // 27-way merge by ranking: position = own offset in its run + number of smaller indices in every other run
    // (indices are distinct; every run is ascending) -- ~26 short binary searches per candidate instead of a full sort
    for (int f = tid; f < total; f += 256) {
      int lo = 0, hi = 27;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (nb_off[mid] <= f) lo = mid; else hi = mid;
      }
      const float4 me = spos[nb_start[lo] + (f - nb_off[lo])];  // re-read from L2: the staging pass has just touched it
      const int key = rawkey[f];
      int pos = f - nb_off[lo];
      for (int k = 0; k < 27; ++k) {
        const int n = nb_cnt[k];
        if (k == lo || n == 0) continue;
        const int* run = rawkey + nb_off[k];
        int a0 = 0, a1 = n;  // first element of the run with index > key
        while (a0 < a1) {
          const int mid = (a0 + a1) >> 1;
          if (run[mid] < key) a0 = mid + 1; else a1 = mid;
        }
        pos += a0;
      }
      cand[pos] = me;
    }
    __syncthreads();
    for (int qq = p0 + tid; qq < p1; qq += 256) {
      const float4 q = spos[qq];
      // hash-aliased slot (another cell behind the same key): leave the query to the per-query kernel
      if ((int)floorf(q.x / radius) != ccx || (int)floorf(q.y / radius) != ccy || (int)floorf(q.z / radius) != ccz) {
        fb_list[atomicAdd(fb_count, 1)] = qq;
        continue;
      }
      int cnt = 0;
      int j = 0;
      for (; j + BQC_ROUND <= total && cnt < nsample; j += BQC_ROUND) {  // several candidates per round: one LDS latency
        float4 pc[BQC_ROUND];
        bool hit[BQC_ROUND];
#pragma unroll
        for (int u = 0; u < BQC_ROUND; ++u) pc[u] = cand[j + u];
#pragma unroll
        for (int u = 0; u < BQC_ROUND; ++u) {
          const float dx = q.x - pc[u].x, dy = q.y - pc[u].y, dz = q.z - pc[u].z;
          hit[u] = fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < r2;
        }
#pragma unroll
        for (int u = 0; u < BQC_ROUND; ++u)
          if (hit[u] && cnt < nsample) list[(int64_t)(cnt++) * M + qq] = __float_as_int(pc[u].w);
      }
      for (; j < total && cnt < nsample; ++j) {
        const float4 pc = cand[j];
        const float dx = q.x - pc.x, dy = q.y - pc.y, dz = q.z - pc.z;
        if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < r2) list[(int64_t)(cnt++) * M + qq] = __float_as_int(pc.w);
      }
      deg[qq] = cnt;
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
// label propagation: one lane per point (cell-sorted position p; labels live in local-index space)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rg_propagate(const int32_t* __restrict__ list, const int32_t* __restrict__ deg,
                                                      const float4* __restrict__ spos, int32_t* L, int32_t* pushed,
                                                      int64_t M, int32_t* changed) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int g = __float_as_int(spos[p].w);
  volatile int32_t* VL = L;
  const int lk0 = VL[g];
  int lk = lk0;
  for (;;) {  // pointer jumping: L[x] <= x always, and L[L[k]] is an ancestor of k
    int pnt = VL[lk];
    if (pnt >= lk) break;
    lk = pnt;
  }
  bool ch = false;
  if (lk < lk0) {
    atomicMin(&L[g], lk);
    ch = true;
  }
  // frontier: a point re-walks its neighbour list only when it has a smaller label to push than last time
  // (labels only decrease, so a label already pushed can never be needed again by the same neighbours)
  if (pushed[p] > lk) {
    pushed[p] = lk;
    const int d = deg[p];
    for (int t = 0; t < d; ++t) {
      const int j = list[(int64_t)t * M + p];
      // L[j] <= j always, so a neighbour with j <= lk cannot be improved: skip its label load
      if (j > lk && VL[j] > lk) {
        int old = atomicMin(&L[j], lk);
        if (old > lk) ch = true;
      }
    }
  }
  if (ch) changed[0] = 1;
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

__global__ __launch_bounds__(256) void k_rg_sizes(const int32_t* __restrict__ L, int64_t M, int32_t* size) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  rg_hist_add_runs(size, v < M ? L[v] : -1, v < M);
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
                                                  uint32_t* ukey, int32_t* hist, int32_t* err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int32_t k = -1;
  if (i < n) {
    k = key[i];
    if (k >= n_groups) {
      atomicAdd(err, 1);
      k = -1;
    }
    ukey[i] = k < 0 ? (uint32_t)n_groups : (uint32_t)k;
  }
  rg_hist_add_runs(hist, k, k >= 0);
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
  int32_t* hist = ar.take<int32_t>((size_t)n_groups + 2);
  PP_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * ((size_t)n_groups + 2), s));
  int32_t* err = hist + n_groups + 1;
  if (n > 0) {
    hipLaunchKernelGGL(k_gbk_keys, dim3(pp_blocks(n, 256)), dim3(256), 0, s, key, n, n_groups, ukey, hist, err);
    hipLaunchKernelGGL(k_iota, dim3(pp_blocks(n, 256)), dim3(256), 0, s, idx, n);
    PP_LAUNCH_CHECK();
    int rc = pp_sort_pairs_u32(ukey, ukey2, idx, idx2, n, bits_for(n_groups), ar.cur(), ar.left(), s);
    if (rc) return rc;
  }
  // offsets[g] = exclusive scan of hist (n_groups+1 entries, the last is 0 -> offsets[n_groups] = total)
  int rc = pp_exclusive_scan_i32(hist, offsets, (int64_t)n_groups + 1, nullptr, ar.cur(), ar.left(), s);
  if (rc) return rc;
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
  hipLaunchKernelGGL(k_ball_query_cells, dim3((unsigned)std::min<int64_t>(M, 4096)), dim3(256), 0, s, spos, sbc, ckeys,
                     cell_start, cell_end, cap, M, radius, nsample, cell_p0, misc + 4, list, deg, fb_list, misc + 5);
  hipLaunchKernelGGL(k_ball_query, dim3((unsigned)std::min<int64_t>(pp_blocks(M, 4), 2048)), dim3(256), 0, s, spos, sbc,
                     ckeys, cell_start, cell_end, cap, M, radius, nsample, fb_list, misc + 5, list, deg);
  hipLaunchKernelGGL(k_iota, dim3(mb), dim3(256), 0, s, L, M);
  int32_t* pushed = size;  // reused as the cluster-size array once the fixpoint is reached
  hipLaunchKernelGGL(k_fill_i32, dim3(fill_blocks(M)), dim3(256), 0, s, pushed, 0x7FFFFFFF, M);
  PP_LAUNCH_CHECK();
  // fixpoint
  bool converged = false;
  for (int round = 0; round < 4096 && !converged; ++round) {
    PP_HIP(hipMemsetAsync(misc + 2, 0, sizeof(int32_t), s));
    for (int it = 0; it < 4; ++it)
      hipLaunchKernelGGL(k_rg_propagate, dim3(mb), dim3(256), 0, s, list, deg, spos, L, pushed, M, misc + 2);
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
  hipLaunchKernelGGL(k_rg_sizes, dim3(mb), dim3(256), 0, s, L, M, size);
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
