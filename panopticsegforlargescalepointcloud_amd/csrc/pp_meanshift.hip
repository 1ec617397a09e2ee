// K12 flat-kernel mean shift (sklearn.cluster.MeanShift(bandwidth, bin_seeding=True) as called by
// torch_points3d/utils/meanshift_cluster.py:9-18,72-123; control flow of sklearn cluster/_mean_shift.py restated
// in SURVEY.md App. D and oracle/panoptic_oracle.c:meanshift_one).
//
//   A  bin seeds      exact hash set of (sample, round(x/bw)) tuples; one representative point per bin
//   B  search grid    uniform grid (cell = bw*(1+1e-4)) over the first min(D,3) dims, points cell-sorted
//   C  iterate        one wave per seed, all seeds concurrent: radius mean in float64, mean kept in float32,
//                     stop at |shift| <= 1e-3*bw or max_iter -- everything in-kernel, no host round trips
//   D  dedup          stable LSD radix passes give sklearn's (count, centre tuple) descending order per sample;
//                     one workgroup per sample runs the greedy suppression (barrier only on surviving centres)
//   E  labels         nearest surviving centre per point
// All samples (cylinders) of the batch are processed together.
#include "pp_common.h"
#include <vector>

#define MS_STRIDE 8  // padded floats per point / centre

struct MSParams {
  const float* x;         // [m, dim]
  const int32_t* offs;    // [ns+1] device copy of the sample offsets
  int64_t m;
  int dim, ns;
  float bwf;
  double bw2, stop;
  int min_pts, max_iter;
  float cell;
  int G, ncell;
};

__device__ inline int ms_sample_of(const int32_t* __restrict__ offs, int ns, int64_t i) {
  int lo = 0, hi = ns;  // largest s with offs[s] <= i
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (offs[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_ms_prepare(MSParams P, int32_t* sample, float* xs_unsorted_dummy) {
  (void)xs_unsorted_dummy;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.m) return;
  int s = ms_sample_of(P.offs, P.ns, i);
  bool active = (P.offs[s + 1] - P.offs[s]) > P.min_pts;
  sample[i] = active ? s : -1;
}

// ---- A: bin seeds -------------------------------------------------------------------------------
__device__ inline void ms_bin_of(const MSParams& P, int64_t i, int* k) {
  for (int d = 0; d < MS_STRIDE; ++d) k[d] = d < P.dim ? (int)rintf(P.x[i * P.dim + d] / P.bwf) : 0;
}
__global__ __launch_bounds__(256) void k_ms_bins(MSParams P, const int32_t* __restrict__ sample, int32_t* table,
                                                 int64_t cap, int32_t* is_rep, int32_t* reps_in_sample) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < P.m;
  int s = in ? sample[i] : -1;
  int k[MS_STRIDE];
  for (int d = 0; d < MS_STRIDE; ++d) k[d] = 0;
  if (s >= 0) ms_bin_of(P, i, k);
  // neighbours in the input order mostly share their bin (points of one object): only the first lane of a run of equal
  // (sample, bin) probes the table -- the others cannot be the bin's representative unless that lane is, and one
  // representative per bin is all that is needed
  bool same = (threadIdx.x & 63) != 0 && s >= 0 && __shfl_up(s, 1) == s;
  for (int d = 0; d < MS_STRIDE; ++d) same = (__shfl_up(k[d], 1) == k[d]) && same;
  if (!in) return;
  if (s < 0 || same) {
    is_rep[i] = 0;
    return;
  }
  uint64_t h = pp_mix64((uint64_t)(uint32_t)s + 0x9E3779B97F4A7C15ull);
  for (int d = 0; d < MS_STRIDE; ++d) h = pp_mix64(h ^ (uint64_t)(uint32_t)k[d]);
  uint64_t mask = (uint64_t)cap - 1, slot = h & mask;
  for (;;) {
    int cur = table[slot];
    if (cur < 0) {
      int prev = atomicCAS(&table[slot], -1, (int)i);
      if (prev < 0) {
        is_rep[i] = 1;
        atomicAdd(&reps_in_sample[s], 1);
        return;
      }
      cur = prev;
    }
    if (sample[cur] == s) {
      int kc[MS_STRIDE];
      ms_bin_of(P, cur, kc);
      bool eq = true;
      for (int d = 0; d < MS_STRIDE; ++d) eq = eq && (kc[d] == k[d]);
      if (eq) {
        is_rep[i] = 0;
        return;
      }
    }
    slot = (slot + 1) & mask;
  }
}
__global__ __launch_bounds__(256) void k_ms_seed_list(int64_t m, const int32_t* __restrict__ is_rep,
                                                      const int32_t* __restrict__ rank, int32_t* seed_point) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m && is_rep[i]) seed_point[rank[i]] = (int32_t)i;
}

// ---- B: search grid -----------------------------------------------------------------------------
__device__ inline uint64_t ms_cell_key(int s, int c0, int c1, int c2) {
  return ((uint64_t)(uint16_t)((uint32_t)s * 2654435761u >> 16) << 48) | ((uint64_t)(uint16_t)c0 << 32) |
         ((uint64_t)(uint16_t)c1 << 16) | (uint64_t)(uint16_t)c2;
}
__global__ __launch_bounds__(256) void k_ms_cells_insert(MSParams P, const int32_t* __restrict__ sample, uint64_t* keys,
                                                         int64_t cap, uint32_t* slot_of) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.m) return;
  int s = sample[i];
  int c[3] = {0, 0, 0};
  for (int d = 0; d < P.G; ++d) c[d] = (int)floorf(P.x[i * P.dim + d] / P.cell);
  // inactive samples go to a cell of their own "sample" id ns (never queried)
  slot_of[i] = (uint32_t)pp_hash_insert_slot(keys, cap, ms_cell_key(s < 0 ? P.ns : s, c[0], c[1], c[2]));
}
__global__ __launch_bounds__(256) void k_ms_cells_sorted(MSParams P, const uint32_t* __restrict__ sorted_slot,
                                                         const int32_t* __restrict__ sorted_idx,
                                                         const int32_t* __restrict__ sample, int32_t* cell_start,
                                                         int32_t* cell_end, float* xs, int32_t* ssample) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.m) return;
  uint32_t s = sorted_slot[p];
  if (p == 0 || sorted_slot[p - 1] != s) cell_start[s] = (int32_t)p;
  if (p == P.m - 1 || sorted_slot[p + 1] != s) cell_end[s] = (int32_t)(p + 1);
  int64_t i = sorted_idx[p];
  for (int d = 0; d < MS_STRIDE; ++d) xs[p * MS_STRIDE + d] = d < P.dim ? P.x[i * P.dim + d] : 0.f;
  ssample[p] = sample[i];
}

// ---- C: iterate ---------------------------------------------------------------------------------
__device__ inline double ms_wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int DIM>
__global__ __launch_bounds__(256) void k_ms_iterate(MSParams P, const float* __restrict__ xs,
                                                    const int32_t* __restrict__ ssample,
                                                    const uint64_t* __restrict__ keys,
                                                    const int32_t* __restrict__ cell_start,
                                                    const int32_t* __restrict__ cell_end, int64_t cap,
                                                    const int32_t* __restrict__ seed_point,
                                                    const int32_t* __restrict__ sample,
                                                    const int32_t* __restrict__ reps_in_sample, int64_t S, float* cen,
                                                    int32_t* cnt_out) {
  const int lane = threadIdx.x & 63;
  const int64_t sidx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (sidx >= S) return;
  const int64_t pi = seed_point[sidx];
  const int s = sample[pi];
  const bool raw = reps_in_sample[s] == (P.offs[s + 1] - P.offs[s]);  // "using data points as seeds"
  float mean[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    float xv = P.x[pi * P.dim + d];
    mean[d] = raw ? xv : (float)((int)rintf(xv / P.bwf)) * P.bwf;
  }
  int it = 0, count = 0;
  for (;;) {
    int my_start = 0, my_cnt = 0;
    if (lane < P.ncell) {
      int c[3] = {0, 0, 0};
      int l = lane;
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (d < P.G && d < DIM) {
          c[d] = (int)floorf(mean[d] / P.cell) + (l % 3 - 1);
          l /= 3;
        }
      int64_t slot = pp_hash_find_slot(keys, cap, ms_cell_key(s, c[0], c[1], c[2]));
      if (slot >= 0) {
        my_start = cell_start[slot];
        my_cnt = cell_end[slot] - my_start;
      }
    }
    double sum[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) sum[d] = 0.0;
    int c_local = 0;
    for (int cidx = 0; cidx < P.ncell; ++cidx) {
      const int st = __shfl(my_start, cidx);
      const int cn = __shfl(my_cnt, cidx);
      for (int t = lane; t < cn; t += 64) {
        const float* px = xs + (int64_t)(st + t) * MS_STRIDE;
        float4 a = *(const float4*)px;
        float4 b = *(const float4*)(px + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        double d2 = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double tdiff = (double)v[d] - (double)mean[d];
          d2 += tdiff * tdiff;
        }
        if (d2 <= P.bw2 && ssample[st + t] == s) {
          ++c_local;
#pragma unroll
          for (int d = 0; d < DIM; ++d) sum[d] += (double)v[d];
        }
      }
    }
    int c = c_local;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    count = c;
    if (c == 0) break;
    double mv = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double tot = ms_wave_sum(sum[d]);
      float nm = (float)(tot / (double)c);
      double tdiff = (double)nm - (double)mean[d];
      mv += tdiff * tdiff;
      mean[d] = nm;
    }
    if (sqrt(mv) <= P.stop || it == P.max_iter) break;
    ++it;
  }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < MS_STRIDE; ++d) cen[sidx * MS_STRIDE + d] = d < DIM ? mean[d] : 0.f;
    cnt_out[sidx] = count;
  }
}

// ---- D: ordering + greedy suppression -----------------------------------------------------------
__device__ inline uint32_t ms_ord_desc(float f) {
  uint32_t u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending order-preserving map
  return ~u;                                        // descending
}
// pass: 0..dim-1 -> centre coordinate (dim-1-pass); dim -> count; dim+1 -> sample
__global__ __launch_bounds__(256) void k_ms_sort_key(const float* __restrict__ cen, const int32_t* __restrict__ cnt,
                                                     const int32_t* __restrict__ seed_point,
                                                     const int32_t* __restrict__ sample,
                                                     const int32_t* __restrict__ perm, int64_t S, int what, int ns,
                                                     uint32_t* key) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S) return;
  int q = perm[p];
  uint32_t k;
  if (what >= 0)
    k = ms_ord_desc(cen[(int64_t)q * MS_STRIDE + what]);
  else if (what == -1)
    k = ~(uint32_t)cnt[q];
  else
    k = cnt[q] > 0 ? (uint32_t)sample[seed_point[q]] : (uint32_t)ns;  // seeds without neighbours sort last
  key[p] = k;
}
__global__ __launch_bounds__(256) void k_ms_segments(const uint32_t* __restrict__ skey, int64_t S, int ns,
                                                     int32_t* seg_start, int32_t* seg_end) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S) return;
  uint32_t s = skey[p];
  if (s >= (uint32_t)ns) return;
  if (p == 0 || skey[p - 1] != s) seg_start[s] = (int32_t)p;
  if (p == S - 1 || skey[p + 1] != s) seg_end[s] = (int32_t)(p + 1);
}

// The same order for samples with few seeds, in one launch: seeds come in point order, so those of a sample are a
// contiguous range of the seed list already; one workgroup per sample ranks each of its seeds against the others under
// the comparator of the radix passes (count descending, coordinates 0..dim-1 descending, original position) with the
// keys in LDS.  The seven stable sorts above are 25 radix passes = ~150 launches of a few microseconds of work each.
#define MS_SORT_SMALL 1024
__global__ __launch_bounds__(256) void k_ms_sort_small(const float* __restrict__ cen, const int32_t* __restrict__ cnt,
                                                       const int32_t* __restrict__ reps_in_sample, int dim,
                                                       int32_t* __restrict__ perm, int32_t* seg_start, int32_t* seg_end) {
  __shared__ uint32_t key[MS_STRIDE + 1][MS_SORT_SMALL];
  __shared__ int red[4];
  __shared__ int nvalid;
  const int s = blockIdx.x, tid = threadIdx.x;
  int part = 0;
  for (int t = tid; t < s; t += 256) part += reps_in_sample[t];
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  if (tid == 0) nvalid = 0;
  __syncthreads();
  const int r0 = red[0] + red[1] + red[2] + red[3];
  const int n = reps_in_sample[s];
  int valid = 0;
  for (int p = tid; p < n; p += 256) {
    const int c = cnt[r0 + p];
    key[0][p] = c > 0 ? ~(uint32_t)c : 0xFFFFFFFFu;  // seeds without neighbours rank last (and stay outside the segment)
    valid += c > 0 ? 1 : 0;
    for (int d = 0; d < MS_STRIDE; ++d) key[1 + d][p] = d < dim ? ms_ord_desc(cen[(int64_t)(r0 + p) * MS_STRIDE + d]) : 0u;
  }
  if (valid) atomicAdd(&nvalid, valid);
  __syncthreads();
  for (int p = tid; p < n; p += 256) {
    uint32_t mine[MS_STRIDE + 1];
    for (int d = 0; d <= MS_STRIDE; ++d) mine[d] = key[d][p];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      int before = j < p ? 1 : 0;  // equal tuples keep their order
#pragma unroll
      for (int d = MS_STRIDE; d >= 0; --d) {
        const uint32_t kj = key[d][j];
        before = kj < mine[d] ? 1 : (kj > mine[d] ? 0 : before);
      }
      rank += before;
    }
    perm[r0 + rank] = r0 + p;
  }
  if (tid == 0) {
    seg_start[s] = r0;
    seg_end[s] = r0 + nvalid;
  }
}

// one workgroup per sample; sorted centres of the sample are perm[seg_start..seg_end)
__global__ __launch_bounds__(256) void k_ms_dedup(const float* __restrict__ cen, const int32_t* __restrict__ cnt,
                                                  const int32_t* __restrict__ perm,
                                                  const int32_t* __restrict__ seg_start,
                                                  const int32_t* __restrict__ seg_end, int dim, double bw2,
                                                  int32_t* alive) {
  const int s = blockIdx.x;
  const int lo = seg_start[s], hi = seg_end[s];
  if (hi <= lo) return;
  // exact duplicates (same count, same tuple) collapse first -- dict keyed by the centre tuple
  for (int p = lo + threadIdx.x; p < hi; p += blockDim.x) {
    int a = 1;
    if (p > lo) {
      int q0 = perm[p - 1], q1 = perm[p];
      bool same = cnt[q0] == cnt[q1];
      for (int d = 0; d < dim; ++d) same = same && (cen[(int64_t)q0 * MS_STRIDE + d] == cen[(int64_t)q1 * MS_STRIDE + d]);
      if (same) a = 0;
    }
    alive[p] = a;
  }
  __syncthreads();
  volatile int32_t* va = alive;
  for (int i = lo; i < hi; ++i) {
    if (!va[i]) continue;  // uniform: alive[] only changes between barriers
    const float* ci = cen + (int64_t)perm[i] * MS_STRIDE;
    float c0[MS_STRIDE];
    for (int d = 0; d < MS_STRIDE; ++d) c0[d] = ci[d];
    for (int p = i + 1 + threadIdx.x; p < hi; p += blockDim.x) {
      if (!va[p]) continue;
      const float* cj = cen + (int64_t)perm[p] * MS_STRIDE;
      double d2 = 0.0;
      for (int d = 0; d < dim; ++d) {
        double t = (double)cj[d] - (double)c0[d];
        d2 += t * t;
      }
      if (d2 <= bw2) alive[p] = 0;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_ms_unique(const float* __restrict__ cen, const int32_t* __restrict__ perm,
                                                   const int32_t* __restrict__ alive, const int32_t* __restrict__ arank,
                                                   int64_t S, float* ucen) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S || !alive[p]) return;
  for (int d = 0; d < MS_STRIDE; ++d) ucen[(int64_t)arank[p] * MS_STRIDE + d] = cen[(int64_t)perm[p] * MS_STRIDE + d];
}
__global__ __launch_bounds__(256) void k_ms_counts(const int32_t* __restrict__ seg_start,
                                                   const int32_t* __restrict__ seg_end,
                                                   const int32_t* __restrict__ alive, const int32_t* __restrict__ arank,
                                                   int ns, int32_t* ubase, int32_t* n_clusters) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  int lo = seg_start[s], hi = seg_end[s];
  if (hi <= lo) {
    ubase[s] = 0;
    n_clusters[s] = 0;
    return;
  }
  ubase[s] = arank[lo];
  n_clusters[s] = arank[hi - 1] + alive[hi - 1] - arank[lo];
}

// ---- E: labels ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ms_labels(MSParams P, const int32_t* __restrict__ sample,
                                                   const float* __restrict__ ucen, const int32_t* __restrict__ ubase,
                                                   const int32_t* __restrict__ n_clusters, int32_t* labels,
                                                   float* centers) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.m) return;
  int s = sample[i];
  if (s < 0) {
    labels[i] = -1;
    return;
  }
  const int U = n_clusters[s];
  const float* uc = ucen + (int64_t)ubase[s] * MS_STRIDE;
  float xv[MS_STRIDE];
  for (int d = 0; d < MS_STRIDE; ++d) xv[d] = d < P.dim ? P.x[i * P.dim + d] : 0.f;
  double best = 1e300;
  int bl = 0;
  for (int c = 0; c < U; ++c) {
    double d2 = 0.0;
    for (int d = 0; d < P.dim; ++d) {
      double t = (double)xv[d] - (double)uc[(int64_t)c * MS_STRIDE + d];
      d2 += t * t;
    }
    if (d2 < best) {
      best = d2;
      bl = c;
    }
  }
  labels[i] = bl;
  // optional centre export: sample s writes its centres at rows offs[s]..offs[s]+U-1
  if (centers) {
    int64_t local = i - P.offs[s];
    if (local < U)
      for (int d = 0; d < P.dim; ++d) centers[i * P.dim + d] = uc[local * MS_STRIDE + d];
  }
}

__global__ __launch_bounds__(256) void k_ms_fill_i32(int32_t* p, int32_t v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(256) void k_ms_fill_u64(uint64_t* p, uint64_t v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(256) void k_ms_iota(int32_t* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}
static inline unsigned msfb(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }
static inline int ms_bits_for(int64_t v) {
  int b = 1;
  while ((1ll << b) <= v) ++b;
  return b;
}

extern "C" size_t pp_meanshift_workspace(int64_t m, int32_t dim, int32_t n_samples) {
  (void)dim;
  size_t mm = (size_t)std::max<int64_t>(m, 1);
  size_t cap = (size_t)pp_hash_capacity((int64_t)mm);
  size_t b = 0;
  b += 14 * pp_align(mm * 4);                       // per-point / per-seed int arrays
  b += 3 * pp_align(mm * MS_STRIDE * 4);            // xs, cen, ucen
  b += pp_align(cap * 8) + 3 * pp_align(cap * 4);   // cell hash + bin table
  b += 6 * pp_align(((size_t)n_samples + 2) * 4);
  b += pp_sort_pairs_workspace(m) + pp_scan_workspace(m) + 8192;
  return b;
}

extern "C" int pp_meanshift(const float* x, int64_t m, int32_t dim, const int64_t* sample_offsets, int32_t n_samples,
                            float bandwidth, int32_t min_points_exclusive, int32_t max_iter, int32_t* labels,
                            int32_t* n_clusters, float* centers, void* workspace, size_t workspace_bytes,
                            pp_stream_t stream) {
  PP_REQUIRE(dim >= 1 && dim <= MS_STRIDE, "pp_meanshift: dim must be in [1,8]");
  PP_REQUIRE(bandwidth > 0.f && n_samples >= 0 && sample_offsets && labels && n_clusters, "pp_meanshift: bad arguments");
  PP_REQUIRE(m < (1ll << 31), "pp_meanshift: too many points");
  if (workspace_bytes < pp_meanshift_workspace(m, dim, n_samples)) return PP_ERR_WORKSPACE;
  hipStream_t s = pp_s(stream);
  if (n_samples == 0 || m == 0) {
    if (n_samples > 0) PP_HIP(hipMemsetAsync(n_clusters, 0, sizeof(int32_t) * n_samples, s));
    return PP_OK;
  }
  PP_REQUIRE(sample_offsets[0] == 0 && sample_offsets[n_samples] == m, "pp_meanshift: sample_offsets must span [0,m]");
  PPArena ar(workspace, workspace_bytes);
  size_t mm = (size_t)m;
  int32_t* offs = ar.take<int32_t>((size_t)n_samples + 2);
  {
    int32_t* h = (int32_t*)malloc(sizeof(int32_t) * ((size_t)n_samples + 1));
    for (int i = 0; i <= n_samples; ++i) {
      h[i] = (int32_t)sample_offsets[i];
      if (i > 0 && sample_offsets[i] < sample_offsets[i - 1]) {
        free(h);
        pp_set_error("pp_meanshift: sample_offsets must be non-decreasing");
        return PP_ERR_INVALID;
      }
    }
    hipError_t e = hipMemcpyAsync(offs, h, sizeof(int32_t) * ((size_t)n_samples + 1), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // h is pageable: make sure it was consumed
    free(h);
    PP_HIP(e);
  }
  MSParams P;
  P.x = x; P.offs = offs; P.m = m; P.dim = dim; P.ns = n_samples; P.bwf = bandwidth;
  P.bw2 = (double)bandwidth * (double)bandwidth; P.stop = 1e-3 * (double)bandwidth;
  P.min_pts = min_points_exclusive; P.max_iter = max_iter;
  P.cell = bandwidth * 1.0001f; P.G = dim < 3 ? dim : 3;
  P.ncell = P.G == 1 ? 3 : (P.G == 2 ? 9 : 27);

  int32_t* sample = ar.take<int32_t>(mm);
  int32_t* is_rep = ar.take<int32_t>(mm);
  int32_t* rank = ar.take<int32_t>(mm);
  int32_t* seed_point = ar.take<int32_t>(mm);
  uint32_t* slot_of = ar.take<uint32_t>(mm);
  uint32_t* sorted_slot = ar.take<uint32_t>(mm);
  int32_t* iota = ar.take<int32_t>(mm);
  int32_t* sorted_idx = ar.take<int32_t>(mm);
  int32_t* ssample = ar.take<int32_t>(mm);
  int32_t* cnt = ar.take<int32_t>(mm);
  int32_t* perm = ar.take<int32_t>(mm);
  int32_t* perm2 = ar.take<int32_t>(mm);
  uint32_t* skey = ar.take<uint32_t>(mm);
  uint32_t* skey2 = ar.take<uint32_t>(mm);
  float* xs = ar.take<float>(mm * MS_STRIDE);
  float* cen = ar.take<float>(mm * MS_STRIDE);
  float* ucen = ar.take<float>(mm * MS_STRIDE);
  const int64_t cap = pp_hash_capacity(m);
  uint64_t* ckeys = ar.take<uint64_t>((size_t)cap);
  int32_t* cell_start = ar.take<int32_t>((size_t)cap);
  int32_t* cell_end = ar.take<int32_t>((size_t)cap);
  int32_t* bin_table = ar.take<int32_t>((size_t)cap);
  int32_t* reps_in_sample = ar.take<int32_t>((size_t)n_samples + 2);
  int32_t* seg_start = ar.take<int32_t>((size_t)n_samples + 2);
  int32_t* seg_end = ar.take<int32_t>((size_t)n_samples + 2);
  int32_t* ubase = ar.take<int32_t>((size_t)n_samples + 2);
  int32_t* misc = ar.take<int32_t>(64);
  PP_REQUIRE(misc && bin_table && ucen, "pp_meanshift: workspace carve failed");

  unsigned mb = pp_blocks(m, 256);
  hipLaunchKernelGGL(k_ms_prepare, dim3(mb), dim3(256), 0, s, P, sample, (float*)nullptr);
  // A
  hipLaunchKernelGGL(k_ms_fill_i32, dim3(msfb(cap)), dim3(256), 0, s, bin_table, -1, cap);
  PP_HIP(hipMemsetAsync(reps_in_sample, 0, sizeof(int32_t) * ((size_t)n_samples + 2), s));
  hipLaunchKernelGGL(k_ms_bins, dim3(mb), dim3(256), 0, s, P, sample, bin_table, cap, is_rep, reps_in_sample);
  PP_LAUNCH_CHECK();
  int rc = pp_exclusive_scan_i32(is_rep, rank, m, misc, ar.cur(), ar.left(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_ms_seed_list, dim3(mb), dim3(256), 0, s, m, is_rep, rank, seed_point);
  // B
  hipLaunchKernelGGL(k_ms_fill_u64, dim3(msfb(cap)), dim3(256), 0, s, ckeys, PP_EMPTY_KEY, cap);
  hipLaunchKernelGGL(k_ms_cells_insert, dim3(mb), dim3(256), 0, s, P, sample, ckeys, cap, slot_of);
  hipLaunchKernelGGL(k_ms_iota, dim3(mb), dim3(256), 0, s, iota, m);
  PP_LAUNCH_CHECK();
  rc = pp_sort_pairs_u32(slot_of, sorted_slot, iota, sorted_idx, m, ms_bits_for(cap - 1), ar.cur(), ar.left(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_ms_cells_sorted, dim3(mb), dim3(256), 0, s, P, sorted_slot, sorted_idx, sample, cell_start,
                     cell_end, xs, ssample);
  PP_LAUNCH_CHECK();
  int32_t S32 = 0;
  std::vector<int32_t> h_reps((size_t)n_samples);
  PP_HIP(hipMemcpyAsync(&S32, misc, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PP_HIP(hipMemcpyAsync(h_reps.data(), reps_in_sample, sizeof(int32_t) * (size_t)n_samples, hipMemcpyDeviceToHost, s));
  PP_HIP(hipStreamSynchronize(s));
  const int64_t S = S32;
  int32_t max_reps = 0;
  for (int32_t r : h_reps) max_reps = std::max(max_reps, r);
  PP_HIP(hipMemsetAsync(n_clusters, 0, sizeof(int32_t) * n_samples, s));
  if (S == 0) {
    hipLaunchKernelGGL(k_ms_fill_i32, dim3(msfb(m)), dim3(256), 0, s, labels, -1, m);
    PP_LAUNCH_CHECK();
    return PP_OK;
  }
  // C
  dim3 ig(pp_blocks(S * 64, 256));
#define MS_IT(D)                                                                                                  \
  hipLaunchKernelGGL((k_ms_iterate<D>), ig, dim3(256), 0, s, P, xs, ssample, ckeys, cell_start, cell_end, cap, \
                     seed_point, sample, reps_in_sample, S, cen, cnt)
  switch (dim) {
    case 1: MS_IT(1); break;
    case 2: MS_IT(2); break;
    case 3: MS_IT(3); break;
    case 4: MS_IT(4); break;
    case 5: MS_IT(5); break;
    case 6: MS_IT(6); break;
    case 7: MS_IT(7); break;
    default: MS_IT(8); break;
  }
#undef MS_IT
  PP_LAUNCH_CHECK();
  // D: stable LSD passes: coordinates dim-1..0 (descending), count (descending), sample (ascending)
  unsigned sb = pp_blocks(S, 256);
  hipLaunchKernelGGL(k_ms_iota, dim3(sb), dim3(256), 0, s, perm, S);
  int32_t* pa = perm;
  int32_t* pb = perm2;
  const bool small = max_reps <= MS_SORT_SMALL;
  if (small) {
    hipLaunchKernelGGL(k_ms_sort_small, dim3((unsigned)n_samples), dim3(256), 0, s, cen, cnt, reps_in_sample, dim, perm2,
                       seg_start, seg_end);
    PP_LAUNCH_CHECK();
    pa = perm2;
  }
  for (int pass = 0; pass < (small ? 0 : dim + 2); ++pass) {
    int what = pass < dim ? (dim - 1 - pass) : (pass == dim ? -1 : -2);
    hipLaunchKernelGGL(k_ms_sort_key, dim3(sb), dim3(256), 0, s, cen, cnt, seed_point, sample, pa, S, what, n_samples,
                       skey);
    PP_LAUNCH_CHECK();
    int bits = what == -2 ? ms_bits_for(n_samples) : 32;
    rc = pp_sort_pairs_u32(skey, skey2, pa, pb, S, bits, ar.cur(), ar.left(), s);
    if (rc) return rc;
    int32_t* t = pa; pa = pb; pb = t;
  }
  if (!small) {  // skey2 now holds the sorted sample ids
    PP_HIP(hipMemsetAsync(seg_start, 0, sizeof(int32_t) * ((size_t)n_samples + 2), s));
    PP_HIP(hipMemsetAsync(seg_end, 0, sizeof(int32_t) * ((size_t)n_samples + 2), s));
    hipLaunchKernelGGL(k_ms_segments, dim3(sb), dim3(256), 0, s, skey2, S, n_samples, seg_start, seg_end);
  }
  int32_t* alive = is_rep;  // reuse
  int32_t* arank = rank;    // reuse
  PP_HIP(hipMemsetAsync(alive, 0, sizeof(int32_t) * (size_t)S, s));
  hipLaunchKernelGGL(k_ms_dedup, dim3((unsigned)n_samples), dim3(256), 0, s, cen, cnt, pa, seg_start, seg_end, dim, P.bw2,
                     alive);
  PP_LAUNCH_CHECK();
  rc = pp_exclusive_scan_i32(alive, arank, S, nullptr, ar.cur(), ar.left(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_ms_unique, dim3(sb), dim3(256), 0, s, cen, pa, alive, arank, S, ucen);
  hipLaunchKernelGGL(k_ms_counts, dim3(pp_blocks(n_samples, 256)), dim3(256), 0, s, seg_start, seg_end, alive, arank,
                     n_samples, ubase, n_clusters);
  // E
  hipLaunchKernelGGL(k_ms_labels, dim3(mb), dim3(256), 0, s, P, sample, ucen, ubase, n_clusters, labels, centers);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
