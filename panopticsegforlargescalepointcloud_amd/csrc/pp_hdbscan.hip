// K13 HDBSCAN* on gfx950: density clustering of the learned embeddings / shifted coordinates.
// Replaces hdbscan.HDBSCAN(min_cluster_size, min_samples, cluster_selection_epsilon).fit_predict(X) as called from
// torch_points3d/utils/hdbscan_cluster.py:8-13 (one call per batch element, :117-167).
//
// All samples of a batch are processed together:
//   1. core distances: brute-force k-th smallest squared distance, float64, candidates staged through LDS
//      (two query points per lane so one LDS broadcast feeds two distance evaluations);
//   2. minimum spanning tree of the mutual-reachability graph by Boruvka rounds.  Edges are compared with the strict
//      total order (w^2, min(a,b), max(a,b)), so the tree is unique and every round's hooks form a forest with only
//      2-cycles (the same edge picked from both sides).  Candidate tiles that lie entirely inside the query's own
//      component are skipped (components become contiguous index ranges quickly because points arrive voxel-sorted);
//   3. the n-1 edges of every sample are ordered by three stable radix sorts ((a,b), then w^2, then sample);
//   4. one workgroup per sample turns the sorted edges into the single-linkage tree, condenses it, accumulates
//      stabilities, runs excess-of-mass selection + the epsilon merge and labels the points (sequential per sample by
//      nature; samples run concurrently).
// Distances are accumulated in dimension order without FMA contraction so that results are bit-identical to the CPU
// oracle (oracle/panoptic_oracle.c: hdbscan_one), which fixes the same conventions.

#include <algorithm>
#include <vector>

#include "pp_common.h"

#pragma clang fp contract(off)

#define HD_MAXD 8
#define HD_TPB 256
#define HD_Q 2                       // query points per lane
#define HD_QPB (HD_TPB * HD_Q)       // query points per block
#define HD_TILE 256                  // candidates per LDS tile

struct HDDesc {
  int32_t lo, hi, q0, s;
  int32_t t0, pad0, pad1, pad2;  // t0: global number of the sample's first candidate tile (tile = HD_TILE consecutive points)
};

struct HDArgs {
  const float* x;
  const HDDesc* desc;
  const double* tlo;  // per candidate tile: bounding box (HD_MAXD doubles each)
  const double* thi;
  int dim;
  double* core2;
  int32_t* comp;
  const int32_t* active;  // per sample: still more than one component
};

// Bounding box of every candidate tile.  Pruning rule used by both scans: the squared distance from a query to the box,
// accumulated with the SAME operations and order as a point distance (t = gap; d += t * t), is a lower bound of the
// computed distance to every point of the tile -- each rounding step is monotone -- so a tile whose bound cannot beat
// the query's current best is skipped without changing any result.
__global__ __launch_bounds__(HD_TILE) void k_hd_tile_bbox(const float* __restrict__ x, const HDDesc* __restrict__ desc,
                                                          int dim, double* tlo, double* thi) {
  // grid.x = query blocks; a query block covers HD_QPB / HD_TILE consecutive tiles of its sample
  __shared__ double slo[HD_TILE / 64][HD_MAXD], shi[HD_TILE / 64][HD_MAXD];
  const HDDesc d = desc[blockIdx.x];
  for (int sub = 0; sub < HD_QPB / HD_TILE; ++sub) {
    const int j0 = d.q0 + sub * HD_TILE;
    if (j0 >= d.hi) break;
    const int j = j0 + (int)threadIdx.x;
    const int tile = d.t0 + (j0 - d.lo) / HD_TILE;
    for (int c = 0; c < HD_MAXD; ++c) {
      double v = (c < dim && j < d.hi) ? (double)x[(size_t)j * dim + c] : 0.0;
      double lo = (c < dim && j < d.hi) ? v : INFINITY, hi = (c < dim && j < d.hi) ? v : -INFINITY;
      for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, off));
        hi = fmax(hi, __shfl_xor(hi, off));
      }
      if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6][c] = lo;
        shi[threadIdx.x >> 6][c] = hi;
      }
    }
    __syncthreads();
    if (threadIdx.x < HD_MAXD) {
      double lo = slo[0][threadIdx.x], hi = shi[0][threadIdx.x];
      for (int w = 1; w < HD_TILE / 64; ++w) {
        lo = fmin(lo, slo[w][threadIdx.x]);
        hi = fmax(hi, shi[w][threadIdx.x]);
      }
      if ((int)threadIdx.x >= dim) lo = hi = 0.0;
      tlo[(size_t)tile * HD_MAXD + threadIdx.x] = lo;
      thi[(size_t)tile * HD_MAXD + threadIdx.x] = hi;
    }
    __syncthreads();
  }
}
__device__ inline double hd_box_dist2(const double* q, const double* lo, const double* hi, int D) {
  double d2 = 0.0;
#pragma unroll
  for (int c = 0; c < HD_MAXD; ++c)
    if (c < D) {
      const double a = lo[c] - q[c], b = q[c] - hi[c];
      const double t = fmax(fmax(a, b), 0.0);
      d2 += t * t;
    }
  return d2;
}
// tiles are visited outwards from the query block's own tile (nearest in index = nearest in space for voxel-sorted
// points), so the bounds tighten early: visit order c, c+1, c-1, c+2, ...
struct HDTileWalk {
  int up, down, n, step;
  __device__ HDTileWalk(int centre, int ntiles) : up(centre), down(centre - 1), n(ntiles), step(0) {}
  __device__ int next() {  // -1 when exhausted
    for (;;) {
      if (up >= n && down < 0) return -1;
      const bool take_up = (step++ & 1) == 0;
      if (take_up && up < n) return up++;
      if (!take_up && down >= 0) return down--;
    }
  }
};

template <int KMAX>
__global__ __launch_bounds__(HD_TPB) void k_hd_core(HDArgs A, const int32_t* kths) {
  __shared__ double tile[HD_TILE * HD_MAXD];
  const HDDesc d = A.desc[blockIdx.x];
  const int D = A.dim;
  const int kth = kths[d.s];
  double q[HD_Q][HD_MAXD];
  double best[HD_Q][KMAX];
  int qi[HD_Q];
#pragma unroll
  for (int u = 0; u < HD_Q; ++u) {
    qi[u] = d.q0 + u * HD_TPB + (int)threadIdx.x;
    const int src = qi[u] < d.hi ? qi[u] : d.hi - 1;
#pragma unroll
    for (int c = 0; c < HD_MAXD; ++c) q[u][c] = c < D ? (double)A.x[(size_t)src * D + c] : 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) best[u][k] = INFINITY;
  }
  const int ntiles = (d.hi - d.lo + HD_TILE - 1) / HD_TILE;
  HDTileWalk walk((d.q0 - d.lo) / HD_TILE, ntiles);
  for (int tl = walk.next(); tl >= 0; tl = walk.next()) {
    const int j0 = d.lo + tl * HD_TILE;
    const int cnt = min(HD_TILE, d.hi - j0);
    // skip the tile when no query of the block can still lower its kth-smallest distance with it
    const double* blo = A.tlo + (size_t)(d.t0 + tl) * HD_MAXD;
    const double* bhi = A.thi + (size_t)(d.t0 + tl) * HD_MAXD;
    bool need = false;
#pragma unroll
    for (int u = 0; u < HD_Q; ++u)
      if (qi[u] < d.hi) {
        double bk = best[u][0];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          if (k == kth) bk = best[u][k];
        need = need || hd_box_dist2(q[u], blo, bhi, D) < bk;
      }
    if (!__syncthreads_or(need)) continue;
    if ((int)threadIdx.x < cnt) {
#pragma unroll
      for (int c = 0; c < HD_MAXD; ++c)
        tile[threadIdx.x * HD_MAXD + c] = c < D ? (double)A.x[(size_t)(j0 + threadIdx.x) * D + c] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      double p[HD_MAXD];
#pragma unroll
      for (int c = 0; c < HD_MAXD; ++c) p[c] = tile[j * HD_MAXD + c];
#pragma unroll
      for (int u = 0; u < HD_Q; ++u) {
        double d2 = 0.0;
#pragma unroll
        for (int c = 0; c < HD_MAXD; ++c)
          if (c < D) {
            const double t = q[u][c] - p[c];
            d2 += t * t;
          }
        if (d2 < best[u][KMAX - 1]) {
          double v = d2;
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            const double lo = fmin(v, best[u][k]), hi = fmax(v, best[u][k]);
            best[u][k] = lo;
            v = hi;
          }
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < HD_Q; ++u)
    if (qi[u] < d.hi) {
      double r = best[u][0];
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k == kth) r = best[u][k];
      A.core2[qi[u]] = r;
      A.comp[qi[u]] = qi[u];
    }
}

struct HDRound {
  unsigned long long* pw;  // per point: best outgoing weight (double bits)
  unsigned long long* pe;  // per point: its edge key (min << 32 | max)
  unsigned long long* cw;  // per component representative: min weight
  unsigned long long* ce;  // per component representative: min edge key among the min-weight ones
  int32_t* next;           // hook target of a representative
  int32_t* parent;         // resolved forest
  int32_t* ncomp;          // per sample
  int32_t* nedge;          // per sample
  int32_t* eu;
  int32_t* ev;
  double* ew;
};

__global__ __launch_bounds__(HD_TPB) void k_hd_best_edge(HDArgs A, HDRound R) {
  __shared__ double tile[HD_TILE * HD_MAXD];
  __shared__ double tcore[HD_TILE];
  __shared__ int tcomp[HD_TILE];
  const HDDesc d = A.desc[blockIdx.x];
  if (!A.active[d.s]) return;
  const int D = A.dim;
  double q[HD_Q][HD_MAXD], qcore[HD_Q], bw[HD_Q];
  unsigned long long be[HD_Q];
  int qi[HD_Q], qc[HD_Q];
#pragma unroll
  for (int u = 0; u < HD_Q; ++u) {
    qi[u] = d.q0 + u * HD_TPB + (int)threadIdx.x;
    const int src = qi[u] < d.hi ? qi[u] : d.hi - 1;
#pragma unroll
    for (int c = 0; c < HD_MAXD; ++c) q[u][c] = c < D ? (double)A.x[(size_t)src * D + c] : 0.0;
    qcore[u] = A.core2[src];
    qc[u] = A.comp[src];
    bw[u] = INFINITY;
    be[u] = ~0ull;
  }
  // is the whole query block inside one component?  (then single-component tiles of it can be skipped outright)
  const int c00 = A.comp[d.q0];
  const int blk_uniform = __syncthreads_and(qc[0] == c00 && qc[1] == c00);
  const int ntiles = (d.hi - d.lo + HD_TILE - 1) / HD_TILE;
  HDTileWalk walk((d.q0 - d.lo) / HD_TILE, ntiles);
  for (int tl = walk.next(); tl >= 0; tl = walk.next()) {
    const int j0 = d.lo + tl * HD_TILE;
    const int cnt = min(HD_TILE, d.hi - j0);
    // bound test first: a tile whose box is farther than every query's current best edge cannot contribute
    // (w >= d2 >= box distance; ties need w == best, which a strictly larger bound excludes)
    const double* blo = A.tlo + (size_t)(d.t0 + tl) * HD_MAXD;
    const double* bhi = A.thi + (size_t)(d.t0 + tl) * HD_MAXD;
    bool need = false;
#pragma unroll
    for (int u = 0; u < HD_Q; ++u)
      if (qi[u] < d.hi) need = need || hd_box_dist2(q[u], blo, bhi, D) <= bw[u];
    if (!__syncthreads_or(need)) continue;
    int mine = A.comp[j0];
    if ((int)threadIdx.x < cnt) {
#pragma unroll
      for (int c = 0; c < HD_MAXD; ++c)
        tile[threadIdx.x * HD_MAXD + c] = c < D ? (double)A.x[(size_t)(j0 + threadIdx.x) * D + c] : 0.0;
      tcore[threadIdx.x] = A.core2[j0 + threadIdx.x];
      mine = A.comp[j0 + threadIdx.x];
      tcomp[threadIdx.x] = mine;
    }
    const int t0 = A.comp[j0];
    const int tile_uniform = __syncthreads_and(mine == t0);
    if (tile_uniform && blk_uniform && t0 == c00) continue;
    for (int j = 0; j < cnt; ++j) {
      const int cb = tcomp[j];
      const bool need0 = cb != qc[0], need1 = cb != qc[1];
      if (!__any(need0 || need1)) continue;
      double p[HD_MAXD];
#pragma unroll
      for (int c = 0; c < HD_MAXD; ++c) p[c] = tile[j * HD_MAXD + c];
      const double pc = tcore[j];
      const int b = j0 + j;
#pragma unroll
      for (int u = 0; u < HD_Q; ++u) {
        if (cb == qc[u]) continue;
        double d2 = 0.0;
#pragma unroll
        for (int c = 0; c < HD_MAXD; ++c)
          if (c < D) {
            const double t = q[u][c] - p[c];
            d2 += t * t;
          }
        const double w = fmax(fmax(qcore[u], pc), d2);
        const int a = qi[u];
        const unsigned long long key =
            a < b ? ((unsigned long long)(unsigned)a << 32 | (unsigned)b) : ((unsigned long long)(unsigned)b << 32 | (unsigned)a);
        if (w < bw[u] || (w == bw[u] && key < be[u])) {
          bw[u] = w;
          be[u] = key;
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < HD_Q; ++u)
    if (qi[u] < d.hi) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(bw[u]);
      R.pw[qi[u]] = bits;
      R.pe[qi[u]] = be[u];
      atomicMin(&R.cw[qc[u]], bits);
    }
}

// thread per point helpers: the descriptor covers HD_QPB points
#define HD_FOR_POINTS(i)                                                                    \
  const HDDesc d = A.desc[blockIdx.x];                                                      \
  if (!A.active[d.s]) return;                                                               \
  for (int i = d.q0 + (int)threadIdx.x; i < min(d.hi, d.q0 + HD_QPB); i += HD_TPB)

__global__ __launch_bounds__(HD_TPB) void k_hd_comp_edge(HDArgs A, HDRound R) {
  HD_FOR_POINTS(i) {
    const int c = A.comp[i];
    if (R.pw[i] == R.cw[c]) atomicMin(&R.ce[c], R.pe[i]);
  }
}
__global__ __launch_bounds__(HD_TPB) void k_hd_link(HDArgs A, HDRound R) {
  HD_FOR_POINTS(i) {
    if (A.comp[i] != i) continue;
    const unsigned long long e = R.ce[i];
    const int u = (int)(e >> 32), v = (int)(e & 0xffffffffu);
    const int cu = A.comp[u];
    R.next[i] = cu == i ? A.comp[v] : cu;
  }
}
__global__ __launch_bounds__(HD_TPB) void k_hd_resolve(HDArgs A, HDRound R) {
  HD_FOR_POINTS(i) {
    if (A.comp[i] != i) continue;
    const int t = R.next[i];
    if (R.next[t] == i && i < t) {  // both sides picked the same edge: the smaller representative becomes the root
      R.parent[i] = i;
    } else {
      R.parent[i] = t;
      const unsigned long long e = R.ce[i];
      const int slot = d.lo + atomicAdd(&R.nedge[d.s], 1);
      R.eu[slot] = (int)(e >> 32);
      R.ev[slot] = (int)(e & 0xffffffffu);
      R.ew[slot] = __longlong_as_double((long long)R.cw[i]);
      atomicSub(&R.ncomp[d.s], 1);
    }
  }
}
__global__ __launch_bounds__(HD_TPB) void k_hd_flatten(HDArgs A, HDRound R) {
  HD_FOR_POINTS(i) {
    int r = A.comp[i];
    while (R.parent[r] != r) r = R.parent[r];
    A.comp[i] = r;  // representatives keep comp[r] == r only if they are roots; readers below use parent[], not comp[]
  }
}
// comp[] is updated in place above while other lanes still walk parent[] (never comp[]) -> no race.  Afterwards the
// per-component minima are reset for the next round.
__global__ __launch_bounds__(HD_TPB) void k_hd_reset(HDArgs A, HDRound R) {
  HD_FOR_POINTS(i) {
    R.cw[i] = ~0ull;
    R.ce[i] = ~0ull;
  }
}
__global__ void k_hd_active(const int32_t* ncomp, int32_t* active, int ns) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < ns) active[s] = ncomp[s] > 1;
}

// ---- edge ordering -------------------------------------------------------------------------------------------------
struct HDSort {
  const int32_t* eu;
  const int32_t* ev;
  const double* ew;
  const int32_t* slot_sample;  // sample of an edge slot, ns for unused slots
};
__global__ void k_hd_keys_uv(HDSort S, unsigned long long* keys, int32_t* vals, int64_t m, int ns) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const bool used = S.slot_sample[i] < ns;
  keys[i] = used ? ((unsigned long long)(unsigned)S.eu[i] << 32 | (unsigned)S.ev[i]) : ~0ull;
  vals[i] = (int32_t)i;
}
__global__ void k_hd_keys_w(HDSort S, const int32_t* perm, unsigned long long* keys, int64_t m, int ns) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int p = perm[i];
  keys[i] = S.slot_sample[p] < ns ? (unsigned long long)__double_as_longlong(S.ew[p]) : ~0ull;
}
__global__ void k_hd_keys_s(HDSort S, const int32_t* perm, unsigned long long* keys, int64_t m) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  keys[i] = (unsigned long long)S.slot_sample[perm[i]];
}
__global__ void k_hd_slot_sample(const HDDesc* desc, int32_t* slot_sample, const int32_t* sample_ok, int ns) {
  const HDDesc d = desc[blockIdx.x];
  for (int i = d.q0 + (int)threadIdx.x; i < min(d.hi, d.q0 + HD_QPB); i += HD_TPB)
    slot_sample[i] = (sample_ok[d.s] && i < d.hi - 1) ? d.s : ns;
}

// ---- per-sample tree work --------------------------------------------------------------------------------------------
struct HDTree {
  const int32_t* offs;      // [ns+1] point offsets
  const int32_t* eoff;      // [ns+1] offsets into the sorted edge list
  const int32_t* sample_ok; // sample takes part (enough points)
  const int32_t* order;     // sorted edge slots
  const int32_t* eu;
  const int32_t* ev;
  const double* ew;
  int32_t* labels;
  int32_t* n_clusters;
  // scratch, all indexed from the sample's point offset (node arrays from twice that)
  int32_t* uf;      // [2m]  union-find, later the BFS queue
  int32_t* size;    // [2m]
  int32_t* relabel; // [2m]
  int32_t* stack;   // [2m]
  int32_t* left;    // [m]
  int32_t* right;   // [m]
  int32_t* pclus;   // [m]
  int32_t* cparent; // [m]
  int32_t* lab;     // [m]
  int32_t* flags;   // [m]  bit0 mark, bit1 selected, bit2 covered, bit3 selected after epsilon, bit4 processed
  double* dist;     // [m]
  double* birth;    // [m]
  double* stab;     // [m]
  double* csum;     // [m]
  int min_cluster_size;
  double eps;
};

__device__ inline int hd_find(int32_t* uf, int x) {
  int r = x;
  while (uf[r] != r) r = uf[r];
  while (uf[x] != r) {
    const int nx = uf[x];
    uf[x] = r;
    x = nx;
  }
  return r;
}

// One WAVE per sample: the tree phases are a serial walk (lane 0), the fills before and the label gather after it use the 64
// lanes.  (The first version launched 256 threads per sample: three of a workgroup's four waves only took part in the fills and
// held their slots for the whole serial walk -- a CU ran 8 samples at a time instead of 32.)
#define HD_TREE_TPB 64
__global__ __launch_bounds__(HD_TREE_TPB) void k_hd_tree(HDTree T) {
  const int s = blockIdx.x;
  const int lo = T.offs[s], n = T.offs[s + 1] - lo;
  int32_t* labels = T.labels + lo;
  if (!T.sample_ok[s]) {
    for (int i = threadIdx.x; i < n; i += HD_TREE_TPB) labels[i] = -1;
    if (threadIdx.x == 0) T.n_clusters[s] = 0;
    return;
  }
  int32_t* uf = T.uf + 2 * (size_t)lo;
  int32_t* size = T.size + 2 * (size_t)lo;
  int32_t* relabel = T.relabel + 2 * (size_t)lo;
  int32_t* stack = T.stack + 2 * (size_t)lo;
  int32_t* left = T.left + lo;
  int32_t* right = T.right + lo;
  int32_t* pclus = T.pclus + lo;
  int32_t* cparent = T.cparent + lo;
  int32_t* lab = T.lab + lo;
  int32_t* flags = T.flags + lo;
  double* dist = T.dist + lo;
  double* birth = T.birth + lo;
  double* stab = T.stab + lo;
  double* csum = T.csum + lo;
  const int nn = 2 * n - 1;
  for (int i = threadIdx.x; i < nn; i += HD_TREE_TPB) {
    uf[i] = i;
    size[i] = i < n ? 1 : 0;
  }
  for (int i = threadIdx.x; i < n; i += HD_TREE_TPB) {
    stab[i] = 0.0;
    csum[i] = 0.0;
    flags[i] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int32_t* order = T.order + T.eoff[s];
    // single-linkage tree: merge i creates node n+i
    for (int i = 0; i < n - 1; ++i) {
      const int slot = order[i];
      const int ra = hd_find(uf, T.eu[slot] - lo), rb = hd_find(uf, T.ev[slot] - lo);
      left[i] = ra;
      right[i] = rb;
      dist[i] = sqrt(T.ew[slot]);
      uf[ra] = n + i;
      uf[rb] = n + i;
      size[n + i] = size[ra] + size[rb];
    }
    // condensed tree, breadth first; uf[] is free now and serves as the queue
    int32_t* queue = uf;
    int nc = 1, qh = 0, qt = 0;
    cparent[0] = -1;
    birth[0] = 0.0;
    relabel[nn - 1] = 0;
    queue[qt++] = nn - 1;
    const int mcs = T.min_cluster_size;
    while (qh < qt) {
      const int node = queue[qh++];
      const int m = node - n, c = relabel[node];
      const int ch0 = left[m], ch1 = right[m];
      const double dd = dist[m];
      const double lam = dd > 0.0 ? 1.0 / dd : INFINITY;
      const bool big0 = size[ch0] >= mcs, big1 = size[ch1] >= mcs;
      const double bc = birth[c];
      double acc = stab[c];
      for (int side = 0; side < 2; ++side) {
        const int ch = side ? ch1 : ch0;
        const bool big = side ? big1 : big0;
        if (big0 && big1) {
          const int id = nc++;
          cparent[id] = c;
          birth[id] = lam;
          acc += (lam - bc) * (double)size[ch];
          relabel[ch] = id;
          if (ch >= n) queue[qt++] = ch;
          else pclus[ch] = id;
        } else if (big) {
          relabel[ch] = c;
          queue[qt++] = ch;
        } else {
          int sp = 0;
          stack[sp++] = ch;
          while (sp) {
            const int v = stack[--sp];
            if (v < n) {
              pclus[v] = c;
              acc += (lam - bc) * 1.0;
            } else {
              stack[sp++] = left[v - n];
              stack[sp++] = right[v - n];
            }
          }
        }
      }
      stab[c] = acc;
    }
    // excess of mass, children (larger ids) first; the root is never a cluster
    for (int c = nc - 1; c >= 1; --c) {
      if (csum[c] > stab[c]) stab[c] = csum[c];
      else flags[c] |= 1;
      csum[cparent[c]] += stab[c];
    }
    for (int c = 1; c < nc; ++c) {
      const int p = cparent[c];
      const int cov = ((flags[p] >> 2) & 1) | (flags[p] & 1);
      int f = flags[c] | (cov << 2);
      if ((f & 1) && !cov) f |= 2;
      flags[c] = f;
    }
    if (T.eps != 0.0 && nc > 1) {
      for (int c = 1; c < nc; ++c) {
        if (!(flags[c] & 2)) continue;
        if (1.0 / birth[c] < T.eps) {
          if (flags[c] & 16) continue;
          int v = c;
          for (;;) {
            const int p = cparent[v];
            if (p == 0) break;
            v = p;
            if (1.0 / birth[p] > T.eps) break;
          }
          flags[v] |= 8;
          // proper descendants of v are "processed"; lab[] is free here and holds the under-v marks
          for (int dsc = v + 1; dsc < nc; ++dsc) {
            const int p = cparent[dsc];
            const int under = (p == v) || (p > v && lab[p]);
            lab[dsc] = under;
            if (under) flags[dsc] |= 16;
          }
        } else {
          flags[c] |= 8;
        }
      }
      for (int c = 1; c < nc; ++c) flags[c] = (flags[c] & ~2) | ((flags[c] & 8) ? 2 : 0);
    }
    int nl = 0;
    lab[0] = -1;
    for (int c = 1; c < nc; ++c) lab[c] = (flags[c] & 2) ? nl++ : lab[cparent[c]];
    T.n_clusters[s] = nl;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += HD_TREE_TPB) labels[i] = lab[pclus[i]];
}

// ---- host ------------------------------------------------------------------------------------------------------------
extern "C" size_t pp_hdbscan_workspace(int64_t m, int32_t n_samples) {
  const size_t mm = (size_t)std::max<int64_t>(m, 1), ns = (size_t)std::max<int32_t>(n_samples, 1);
  size_t b = 0;
  b += pp_align((mm / HD_QPB + ns + 2) * sizeof(HDDesc));
  b += 2 * pp_align((mm / HD_TILE + ns + 2) * HD_MAXD * 8);  // tile bounding boxes
  b += 8 * pp_align((ns + 2) * 4);
  b += 5 * pp_align(mm * 8);          // core2, pw, pe, cw, ce
  b += 8 * pp_align(mm * 4);          // comp, next, parent, eu, ev, slot_sample, perm a/b
  b += pp_align(mm * 8);              // ew
  b += 2 * pp_align(mm * 8);          // sort keys in/out
  b += 2 * pp_align(mm * 4);          // sort vals
  b += 4 * pp_align(2 * mm * 4);      // uf, size, relabel, stack
  b += 6 * pp_align(mm * 4);          // left, right, pclus, cparent, lab, flags
  b += 4 * pp_align(mm * 8);          // dist, birth, stab, csum
  b += pp_sort_pairs_workspace((int64_t)mm) + 8192;
  return b;
}

extern "C" int pp_hdbscan(const float* x, int64_t m, int32_t dim, const int64_t* sample_offsets, int32_t n_samples,
                          int32_t min_points_exclusive, int32_t min_cluster_size, int32_t min_samples,
                          int32_t count_self, double cluster_selection_epsilon, int32_t* labels, int32_t* n_clusters,
                          void* workspace, size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(dim >= 1 && dim <= HD_MAXD, "pp_hdbscan: dim must be in [1,8]");
  PP_REQUIRE(n_samples >= 0 && sample_offsets && labels && n_clusters, "pp_hdbscan: bad arguments");
  PP_REQUIRE(min_cluster_size >= 2 && min_samples >= 1 && min_samples <= 31, "pp_hdbscan: min_cluster_size >= 2, 1 <= min_samples <= 31");
  PP_REQUIRE(m < (1ll << 30), "pp_hdbscan: too many points");
  if (workspace_bytes < pp_hdbscan_workspace(m, n_samples)) return PP_ERR_WORKSPACE;
  hipStream_t st = pp_s(stream);
  if (n_samples == 0) return PP_OK;
  PP_REQUIRE(sample_offsets[0] == 0 && sample_offsets[n_samples] == m, "pp_hdbscan: sample_offsets must span [0,m]");
  const size_t mm = (size_t)std::max<int64_t>(m, 1);
  const int ns = n_samples;
  // host-side plan: which samples take part, query blocks, offsets of the sorted edge list
  std::vector<int32_t> h_offs(ns + 1), h_ok(ns), h_eoff(ns + 1), h_ncomp(ns), h_active(ns), h_kth(ns);
  int max_kth = 0;
  std::vector<HDDesc> h_desc;
  int64_t max_n = 0, n_edges = 0, n_tiles = 0;
  for (int s = 0; s <= ns; ++s) {
    PP_REQUIRE(s == 0 || sample_offsets[s] >= sample_offsets[s - 1], "pp_hdbscan: sample_offsets must be non-decreasing");
    h_offs[s] = (int32_t)sample_offsets[s];
  }
  for (int s = 0; s < ns; ++s) {
    const int n = h_offs[s + 1] - h_offs[s];
    h_ok[s] = n > min_points_exclusive && n >= 2;
    h_eoff[s] = (int32_t)n_edges;
    h_ncomp[s] = h_ok[s] ? n : 0;
    h_active[s] = h_ok[s];
    // rank in the ascending row of squared distances (entry 0 = the point itself); hdbscan_.py clamps
    // min_samples = min(size - 1, min_samples)
    const int ms = std::max(1, std::min(min_samples, n - 1));
    h_kth[s] = n >= 2 ? (count_self ? ms - 1 : ms) : 0;
    if (h_ok[s]) {
      n_edges += n - 1;
      max_n = std::max<int64_t>(max_n, n);
      max_kth = std::max(max_kth, h_kth[s]);
    }
    for (int q0 = h_offs[s]; q0 < h_offs[s + 1]; q0 += HD_QPB)
      h_desc.push_back({h_offs[s], h_offs[s + 1], q0, s, (int32_t)n_tiles, 0, 0, 0});
    n_tiles += (n + HD_TILE - 1) / HD_TILE;
  }
  h_eoff[ns] = (int32_t)n_edges;
  const unsigned nblk = (unsigned)h_desc.size();

  PPArena ar(workspace, workspace_bytes);
  HDDesc* desc = ar.take<HDDesc>(mm / HD_QPB + ns + 2);
  double* tlo = ar.take<double>((mm / HD_TILE + ns + 2) * HD_MAXD);
  double* thi = ar.take<double>((mm / HD_TILE + ns + 2) * HD_MAXD);
  int32_t* offs = ar.take<int32_t>(ns + 2);
  int32_t* eoff = ar.take<int32_t>(ns + 2);
  int32_t* ok = ar.take<int32_t>(ns + 2);
  int32_t* ncomp = ar.take<int32_t>(ns + 2);
  int32_t* nedge = ar.take<int32_t>(ns + 2);
  int32_t* active = ar.take<int32_t>(ns + 2);
  int32_t* kths = ar.take<int32_t>(ns + 2);
  double* core2 = ar.take<double>(mm);
  unsigned long long* pw = ar.take<unsigned long long>(mm);
  unsigned long long* pe = ar.take<unsigned long long>(mm);
  unsigned long long* cw = ar.take<unsigned long long>(mm);
  unsigned long long* ce = ar.take<unsigned long long>(mm);
  int32_t* comp = ar.take<int32_t>(mm);
  int32_t* next = ar.take<int32_t>(mm);
  int32_t* parent = ar.take<int32_t>(mm);
  int32_t* eu = ar.take<int32_t>(mm);
  int32_t* ev = ar.take<int32_t>(mm);
  int32_t* slot_sample = ar.take<int32_t>(mm);
  int32_t* perm_a = ar.take<int32_t>(mm);
  int32_t* perm_b = ar.take<int32_t>(mm);
  double* ew = ar.take<double>(mm);
  unsigned long long* keys_a = ar.take<unsigned long long>(mm);
  unsigned long long* keys_b = ar.take<unsigned long long>(mm);
  int32_t* vals_a = ar.take<int32_t>(mm);
  HDTree T;
  T.uf = ar.take<int32_t>(2 * mm);
  T.size = ar.take<int32_t>(2 * mm);
  T.relabel = ar.take<int32_t>(2 * mm);
  T.stack = ar.take<int32_t>(2 * mm);
  T.left = ar.take<int32_t>(mm);
  T.right = ar.take<int32_t>(mm);
  T.pclus = ar.take<int32_t>(mm);
  T.cparent = ar.take<int32_t>(mm);
  T.lab = ar.take<int32_t>(mm);
  T.flags = ar.take<int32_t>(mm);
  T.dist = ar.take<double>(mm);
  T.birth = ar.take<double>(mm);
  T.stab = ar.take<double>(mm);
  T.csum = ar.take<double>(mm);
  const size_t sort_ws_bytes = pp_sort_pairs_workspace((int64_t)mm);
  void* sort_ws = ar.take<char>(sort_ws_bytes);
  PP_REQUIRE(sort_ws != nullptr, "pp_hdbscan: workspace arithmetic");

  // uploads (pageable sources: synchronise before they go out of scope)
  PP_HIP(hipMemcpyAsync(desc, h_desc.data(), sizeof(HDDesc) * h_desc.size(), hipMemcpyHostToDevice, st));
  PP_HIP(hipMemcpyAsync(offs, h_offs.data(), 4 * (ns + 1), hipMemcpyHostToDevice, st));
  PP_HIP(hipMemcpyAsync(eoff, h_eoff.data(), 4 * (ns + 1), hipMemcpyHostToDevice, st));
  PP_HIP(hipMemcpyAsync(ok, h_ok.data(), 4 * ns, hipMemcpyHostToDevice, st));
  PP_HIP(hipMemcpyAsync(ncomp, h_ncomp.data(), 4 * ns, hipMemcpyHostToDevice, st));
  PP_HIP(hipMemcpyAsync(active, h_active.data(), 4 * ns, hipMemcpyHostToDevice, st));
  PP_HIP(hipMemcpyAsync(kths, h_kth.data(), 4 * ns, hipMemcpyHostToDevice, st));
  PP_HIP(hipMemsetAsync(nedge, 0, 4 * (ns + 1), st));
  PP_HIP(hipStreamSynchronize(st));

  if (nblk > 0 && n_edges > 0) {
    HDArgs A{x, desc, tlo, thi, dim, core2, comp, active};
    k_hd_tile_bbox<<<nblk, HD_TILE, 0, st>>>(x, desc, dim, tlo, thi);
    HDRound R{pw, pe, cw, ce, next, parent, ncomp, nedge, eu, ev, ew};
    // core distances; the neighbour rank is per sample (clamped for tiny samples, see h_kth)
    if (max_kth < 8) k_hd_core<8><<<nblk, HD_TPB, 0, st>>>(A, kths);
    else k_hd_core<32><<<nblk, HD_TPB, 0, st>>>(A, kths);
    PP_LAUNCH_CHECK();
    PP_HIP(hipMemsetAsync(cw, 0xFF, 8 * mm, st));
    PP_HIP(hipMemsetAsync(ce, 0xFF, 8 * mm, st));
    int rounds = 1;
    while ((1ll << rounds) < max_n) ++rounds;
    for (int r = 0; r < rounds; ++r) {
      k_hd_best_edge<<<nblk, HD_TPB, 0, st>>>(A, R);
      k_hd_comp_edge<<<nblk, HD_TPB, 0, st>>>(A, R);
      k_hd_link<<<nblk, HD_TPB, 0, st>>>(A, R);
      k_hd_resolve<<<nblk, HD_TPB, 0, st>>>(A, R);
      k_hd_flatten<<<nblk, HD_TPB, 0, st>>>(A, R);
      k_hd_reset<<<nblk, HD_TPB, 0, st>>>(A, R);
      k_hd_active<<<pp_blocks(ns, 256), 256, 0, st>>>(ncomp, active, ns);
      PP_LAUNCH_CHECK();
    }
    // order the edges: (a,b), then weight, then sample -- all stable
    HDSort S{eu, ev, ew, slot_sample};
    k_hd_slot_sample<<<nblk, HD_TPB, 0, st>>>(desc, slot_sample, ok, ns);
    const unsigned gb = pp_blocks(m, 256);
    k_hd_keys_uv<<<gb, 256, 0, st>>>(S, keys_a, vals_a, m, ns);
    int rc = pp_sort_pairs_u64((const uint64_t*)keys_a, (uint64_t*)keys_b, vals_a, perm_a, m, 64, sort_ws, sort_ws_bytes, st);
    if (rc != PP_OK) return rc;
    k_hd_keys_w<<<gb, 256, 0, st>>>(S, perm_a, keys_a, m, ns);
    rc = pp_sort_pairs_u64((const uint64_t*)keys_a, (uint64_t*)keys_b, perm_a, perm_b, m, 64, sort_ws, sort_ws_bytes, st);
    if (rc != PP_OK) return rc;
    k_hd_keys_s<<<gb, 256, 0, st>>>(S, perm_b, keys_a, m);
    int bits = 1;
    while ((1 << bits) <= ns) ++bits;
    rc = pp_sort_pairs_u64((const uint64_t*)keys_a, (uint64_t*)keys_b, perm_b, perm_a, m, bits, sort_ws, sort_ws_bytes, st);
    if (rc != PP_OK) return rc;
    PP_LAUNCH_CHECK();
  }
  T.offs = offs;
  T.eoff = eoff;
  T.sample_ok = ok;
  T.order = perm_a;
  T.eu = eu;
  T.ev = ev;
  T.ew = ew;
  T.labels = labels;
  T.n_clusters = n_clusters;
  T.min_cluster_size = min_cluster_size;
  T.eps = cluster_selection_epsilon;
  k_hd_tree<<<ns, HD_TREE_TPB, 0, st>>>(T);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
