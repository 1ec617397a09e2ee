// K6 batch-norm pieces on [n,C] features, fused MLP heads, K11 segment reductions.  All HBM-bound:
// float4 / row-contiguous accesses, one pass over the data.
// References: ME.MinkowskiBatchNorm (= BatchNorm1d on F) torch_points3d/modules/MinkowskiEngine/api_modules.py:40;
// heads torch_points3d/models/panoptic/PointGroup3heads.py:69-81,106-108 and
// torch_points3d/core/common_modules/base_modules.py:35-45; torch_scatter.scatter PointGroup3heads.py:419-452.
#include <string.h>

#include "pp_common.h"

// ---------------------------------------------------------------------------------------------
// per-channel sum / sum of squares (float64 accumulation, one atomic per wave per channel)
// lanes: c <= 64 -> 64/c rows per wave step; else one row per step, lane handles cols l, l+64, l+128, l+192
// ---------------------------------------------------------------------------------------------
template <bool TWO>
__global__ __launch_bounds__(256) void k_channel_reduce(const float* __restrict__ x, const float* __restrict__ y,
                                                        int64_t n, int c, double* __restrict__ o0,
                                                        double* __restrict__ o1) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  if (c <= 64) {
    const int rpw = 64 / c;
    const int col = lane % c;
    const int sub = lane / c;
    double s0 = 0.0, s1 = 0.0;
    if (sub < rpw) {
      for (int64_t r = wave * rpw + sub; r < n; r += nwaves * rpw) {
        double v = (double)x[r * c + col];
        if (TWO) {
          double w = (double)y[r * c + col];
          s0 += w;       // sum(dy)
          s1 += w * v;   // sum(dy*x)
        } else {
          s0 += v;
          s1 += v * v;
        }
      }
      atomicAdd(&o0[col], s0);
      atomicAdd(&o1[col], s1);
    }
  } else {
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    for (int64_t r = wave; r < n; r += nwaves) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int col = lane + 64 * u;
        if (col < c) {
          double v = (double)x[r * c + col];
          if (TWO) {
            double w = (double)y[r * c + col];
            s0[u] += w;
            s1[u] += w * v;
          } else {
            s0[u] += v;
            s1[u] += v * v;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int col = lane + 64 * u;
      if (col < c) {
        atomicAdd(&o0[col], s0[u]);
        atomicAdd(&o1[col], s1[u]);
      }
    }
  }
}

static int channel_reduce(const float* x, const float* y, int64_t n, int c, double* o0, double* o1, hipStream_t s) {
  PP_REQUIRE(c >= 1 && c <= 256, "channel reduce: c must be in [1,256]");
  PP_HIP(hipMemsetAsync(o0, 0, sizeof(double) * c, s));
  PP_HIP(hipMemsetAsync(o1, 0, sizeof(double) * c, s));
  if (n == 0) return PP_OK;
  int64_t rows_per_wave_step = c <= 64 ? 64 / c : 1;
  int64_t want_waves = (n + rows_per_wave_step * 16 - 1) / (rows_per_wave_step * 16);  // >= 16 steps per wave
  unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((want_waves + 3) / 4, 2048));
  if (y)
    hipLaunchKernelGGL(k_channel_reduce<true>, dim3(blocks), dim3(256), 0, s, x, y, n, c, o0, o1);
  else
    hipLaunchKernelGGL(k_channel_reduce<false>, dim3(blocks), dim3(256), 0, s, x, y, n, c, o0, o1);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_channel_stats(const float* x, int64_t n, int32_t c, double* sum, double* sumsq, pp_stream_t stream) {
  return channel_reduce(x, nullptr, n, c, sum, sumsq, pp_s(stream));
}
extern "C" int pp_bn_bwd_reduce(const float* x, const float* dy, int64_t n, int32_t c, double* sum_dy,
                                double* sum_dy_x, pp_stream_t stream) {
  PP_REQUIRE(dy, "pp_bn_bwd_reduce: null dy");
  return channel_reduce(x, dy, n, c, sum_dy, sum_dy_x, pp_s(stream));
}

// ---------------------------------------------------------------------------------------------
// weight gradient of a skinny Linear layer (the heads, the scorer head: cin, cout <= 32 over millions of rows):
//   dW[o][i] = sum_n dy[n][o] * x[n][i],   db[o] = sum_n dy[n][o]
// torch hands this [cout x n] x [n x cin] product to rocBLAS, whose split-K kernels for K = 325 k take 340 - 720 us per
// layer (6 layers = 2.8 ms of a 43 ms training step); it is a streaming reduction: both operands are read once.
// Block partials in float64 without atomics (tiles of 64 rows staged in LDS, fp32 products, one float64 add per tile and
// entry), then a finalize pass adds the partials in block order -> run-to-run reproducible.
// ---------------------------------------------------------------------------------------------
#define LWG_TILE 64
__global__ __launch_bounds__(256) void k_linear_wgrad_partial(const float* __restrict__ x, const float* __restrict__ dy, int64_t n,
                                                               int cin, int cout, double* __restrict__ partial) {
  __shared__ float xs[LWG_TILE][33];
  __shared__ float ds[LWG_TILE][33];
  const int tid = threadIdx.x;
  const int ne = cin * cout;  // <= 1024 entries, thread t owns t, t + 256, ...; entry e = o * cin + i
  double acc[4] = {0.0, 0.0, 0.0, 0.0}, accb = 0.0;
  const int64_t n_tiles = (n + LWG_TILE - 1) / LWG_TILE;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int64_t r0 = t * LWG_TILE;
    const int rows = (int)((n - r0) < LWG_TILE ? (n - r0) : LWG_TILE);
    __syncthreads();
    for (int e = tid; e < LWG_TILE * cin; e += 256) {
      const int r = e / cin, i = e - r * cin;
      xs[r][i] = r < rows ? x[(r0 + r) * cin + i] : 0.f;
    }
    for (int e = tid; e < LWG_TILE * cout; e += 256) {
      const int r = e / cout, o = e - r * cout;
      ds[r][o] = r < rows ? dy[(r0 + r) * cout + o] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      if (e < ne) {
        const int o = e / cin, i = e - o * cin;
        float a = 0.f;
#pragma unroll 8
        for (int r = 0; r < LWG_TILE; ++r) a = fmaf(ds[r][o], xs[r][i], a);
        acc[u] += (double)a;
      }
    }
    if (tid < cout) {
      float a = 0.f;
      for (int r = 0; r < LWG_TILE; ++r) a += ds[r][tid];
      accb += (double)a;
    }
  }
  double* out = partial + (int64_t)blockIdx.x * (ne + cout);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u;
    if (e < ne) out[e] = acc[u];
  }
  if (tid < cout) out[ne + tid] = accb;
}
// one block per entry: 256 threads split the partials (fixed assignment), LDS tree in a fixed order -> reproducible
__global__ __launch_bounds__(256) void k_linear_wgrad_finish(const double* __restrict__ partial, int blocks, int ne, int cout,
                                                              float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double red[256];
  const int e = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int b = tid; b < blocks; b += 256) s += partial[(int64_t)b * (ne + cout) + e];
  red[tid] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid == 0) {
    if (e < ne) dw[e] = (float)red[0];
    else if (db) db[e - ne] = (float)red[0];
  }
}
static int lwg_blocks(int64_t n) {
  const int64_t t = (n + LWG_TILE - 1) / LWG_TILE;
  return (int)(t < 1 ? 1 : (t > 1024 ? 1024 : t));
}
extern "C" size_t pp_linear_wgrad_workspace(int64_t n, int32_t cin, int32_t cout) {
  return pp_align((size_t)lwg_blocks(n) * (size_t)(cin * cout + cout) * sizeof(double));
}
extern "C" int pp_linear_wgrad(const float* x, const float* dy, int64_t n, int32_t cin, int32_t cout, float* dw, float* db,
                               void* ws, size_t ws_bytes, pp_stream_t stream) {
  PP_REQUIRE(dw && ws, "pp_linear_wgrad: null pointer");
  PP_REQUIRE(cin >= 1 && cin <= 32 && cout >= 1 && cout <= 32, "pp_linear_wgrad: cin, cout must be in [1,32]");
  PP_REQUIRE(n >= 0 && (n == 0 || (x && dy)), "pp_linear_wgrad: null input");
  PP_REQUIRE(ws_bytes >= pp_linear_wgrad_workspace(n, cin, cout), "pp_linear_wgrad: workspace too small");
  hipStream_t s = pp_s(stream);
  const int blocks = lwg_blocks(n);
  const int ne = cin * cout;
  hipLaunchKernelGGL(k_linear_wgrad_partial, dim3(blocks), dim3(256), 0, s, x, dy, n, cin, cout, (double*)ws);
  hipLaunchKernelGGL(k_linear_wgrad_finish, dim3(ne + cout), dim3(256), 0, s, (const double*)ws, blocks, ne, cout, dw, db);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// y = act(x*scale + shift) + residual     (c % 4 == 0: float4 path)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_affine_act4(const float4* __restrict__ x, int64_t n4, int c,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     int act, float slope, const float4* __restrict__ res,
                                                     float4* __restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    int col = (int)((i * 4) % c);
    float4 v = x[i];
    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t = e[u];
      if (scale) t *= scale[col + u];
      if (shift) t += shift[col + u];
      if (act == 1) t = fmaxf(t, 0.f);
      if (act == 2) t = t < 0.f ? t * slope : t;
      e[u] = t;
    }
    if (res) {
      float4 r = res[i];
      e[0] += r.x; e[1] += r.y; e[2] += r.z; e[3] += r.w;
    }
    y[i] = make_float4(e[0], e[1], e[2], e[3]);
  }
}
__global__ __launch_bounds__(256) void k_affine_act1(const float* __restrict__ x, int64_t n, int c,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     int act, float slope, const float* __restrict__ res,
                                                     float* __restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int col = (int)(i % c);
    float t = x[i];
    if (scale) t *= scale[col];
    if (shift) t += shift[col];
    if (act == 1) t = fmaxf(t, 0.f);
    if (act == 2) t = t < 0.f ? t * slope : t;
    if (res) t += res[i];
    y[i] = t;
  }
}
extern "C" int pp_affine_act(const float* x, int64_t n, int32_t c, const float* scale, const float* shift,
                             int32_t act, float slope, const float* residual, float* y, pp_stream_t stream) {
  PP_REQUIRE(x && y && c >= 1, "pp_affine_act: bad arguments");
  int64_t total = n * c;
  if (total == 0) return PP_OK;
  hipStream_t s = pp_s(stream);
  if (c % 4 == 0) {
    int64_t n4 = total / 4;
    unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(k_affine_act4, dim3(blocks), dim3(256), 0, s, (const float4*)x, n4, c, scale, shift, act, slope,
                       (const float4*)residual, (float4*)y);
  } else {
    unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(k_affine_act1, dim3(blocks), dim3(256), 0, s, x, total, c, scale, shift, act, slope, residual,
                       y);
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// training-mode BatchNorm1d in three launches, no atomics (bit-reproducible run to run):
//   k_bn_partial  : every block strides over rows, float64 accumulators per thread (4 rows in flight), LDS reduction
//                   over the rows a block step covers -> partial[block][2][c]
//   k_bn_finalize : 8 channels per block, 32 slices over the block partials -> statistics + per-channel coefficients
//   forward  : y  = act(x*scale + shift)                      (k_affine_act*)
//   backward : dx = a*dy' + b*x + c, dy' = dy masked by y > 0 when the ReLU was fused into the forward
// MODE 0: sum x, sum x^2;  MODE 1: sum dy, sum dy*x;  MODE 2: as 1 with the ReLU mask
// ---------------------------------------------------------------------------------------------
#define BN_MAX_BLOCKS 256

template <int VEC>
__device__ __forceinline__ void bn_ld(const float* __restrict__ p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    float4 q = *(const float4*)p;
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
    v[0] = *p;
  }
}
template <int VEC, int MODE>
__device__ __forceinline__ void bn_acc(const float (&a)[VEC], const float (&d)[VEC], const float (&m)[VEC],
                                       double (&s0)[VEC], double (&s1)[VEC]) {
#pragma unroll
  for (int u = 0; u < VEC; ++u) {
    if (MODE == 0) {
      double v = (double)a[u];
      s0[u] += v;
      s1[u] += v * v;
    } else {
      double w = (MODE == 2 && !(m[u] > 0.f)) ? 0.0 : (double)d[u];
      s0[u] += w;
      s1[u] += w * (double)a[u];
    }
  }
}

template <int VEC, int MODE>
__global__ __launch_bounds__(256) void k_bn_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                    const float* __restrict__ yr, int64_t n, int c,
                                                    double* __restrict__ partial) {
  __shared__ double sm[256 * 2 * VEC];
  const int cv = c / VEC, rps = 256 / cv;
  const int t = threadIdx.x, rin = t / cv, cc = t - rin * cv;
  double s0[VEC], s1[VEC];
#pragma unroll
  for (int u = 0; u < VEC; ++u) s0[u] = s1[u] = 0.0;
  if (rin < rps) {
    const int64_t G = (int64_t)gridDim.x * rps;
    int64_t r = (int64_t)blockIdx.x * rps + rin;
    const int64_t co = (int64_t)cc * VEC;
    for (; r + 3 * G < n; r += 4 * G) {
      float a[4][VEC], d[4][VEC], m[4][VEC];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t o = (r + q * G) * c + co;
        bn_ld<VEC>(x + o, a[q]);
        if (MODE >= 1) bn_ld<VEC>(dy + o, d[q]);
        if (MODE == 2) bn_ld<VEC>(yr + o, m[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) bn_acc<VEC, MODE>(a[q], d[q], m[q], s0, s1);
    }
    for (; r < n; r += G) {
      float a[VEC], d[VEC], m[VEC];
      const int64_t o = r * c + co;
      bn_ld<VEC>(x + o, a);
      if (MODE >= 1) bn_ld<VEC>(dy + o, d);
      if (MODE == 2) bn_ld<VEC>(yr + o, m);
      bn_acc<VEC, MODE>(a, d, m, s0, s1);
    }
  }
#pragma unroll
  for (int u = 0; u < VEC; ++u) {
    sm[t * 2 * VEC + u] = s0[u];
    sm[t * 2 * VEC + VEC + u] = s1[u];
  }
  __syncthreads();
  if (t < c) {
    const int pc = t / VEC, u = t - pc * VEC;
    double a0 = 0.0, a1 = 0.0;
    for (int q = 0; q < rps; ++q) {
      a0 += sm[(q * cv + pc) * 2 * VEC + u];
      a1 += sm[(q * cv + pc) * 2 * VEC + VEC + u];
    }
    partial[((int64_t)blockIdx.x * 2) * c + t] = a0;
    partial[((int64_t)blockIdx.x * 2 + 1) * c + t] = a1;
  }
}

struct BnFinalize {
  const double* partial;
  int nb, c, backward;
  double n, eps, momentum;
  const float *weight, *bias;
  float *running_mean, *running_var;   // forward, nullable
  int64_t* num_batches_tracked;        // forward, nullable: += 1 (nn.BatchNorm's counter, kept on the device)
  float *k0, *k1, *k2;                 // forward: scale, shift, -   backward: a, b, c
  double *save_mean, *save_rstd;       // forward: written; backward: read
  float *dweight, *dbias;              // backward, nullable
};

__global__ __launch_bounds__(256) void k_bn_finalize(BnFinalize f) {
  // 8 channels per block, 32 slices over the block partials; the (<= BN_MAX_BLOCKS / 32) loads of a thread are
  // independent and unrolled, so the kernel costs one memory latency, not one per partial block
  __shared__ double sm[2][32][8];
  const int cl = threadIdx.x & 7, j = threadIdx.x >> 3, col = blockIdx.x * 8 + cl;
  double a0 = 0.0, a1 = 0.0;
  if (col < f.c) {
    double v0[BN_MAX_BLOCKS / 32], v1[BN_MAX_BLOCKS / 32];
#pragma unroll
    for (int u = 0; u < BN_MAX_BLOCKS / 32; ++u) {
      const int b = j + 32 * u;
      v0[u] = b < f.nb ? f.partial[((int64_t)b * 2) * f.c + col] : 0.0;
      v1[u] = b < f.nb ? f.partial[((int64_t)b * 2 + 1) * f.c + col] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < BN_MAX_BLOCKS / 32; ++u) {
      a0 += v0[u];
      a1 += v1[u];
    }
  }
  sm[0][j][cl] = a0;
  sm[1][j][cl] = a1;
  __syncthreads();
  if (j != 0 || col >= f.c) return;
  if (col == 0 && !f.backward && f.num_batches_tracked) *f.num_batches_tracked += 1;
  a0 = a1 = 0.0;
  for (int q = 0; q < 32; ++q) {
    a0 += sm[0][q][cl];
    a1 += sm[1][q][cl];
  }
  const double w = f.weight ? (double)f.weight[col] : 1.0;
  if (!f.backward) {
    const double mean = a0 / f.n;
    double var = a1 / f.n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + f.eps);
    const double b = f.bias ? (double)f.bias[col] : 0.0;
    f.k0[col] = (float)(w * rstd);
    f.k1[col] = (float)(b - mean * w * rstd);
    f.save_mean[col] = mean;
    f.save_rstd[col] = rstd;
    if (f.running_mean) {
      const double unbiased = var * (f.n / (f.n > 1.0 ? f.n - 1.0 : 1.0));
      f.running_mean[col] = (float)((1.0 - f.momentum) * (double)f.running_mean[col] + f.momentum * mean);
      f.running_var[col] = (float)((1.0 - f.momentum) * (double)f.running_var[col] + f.momentum * unbiased);
    }
  } else {
    const double mean = f.save_mean[col], rstd = f.save_rstd[col];
    const double g = (a1 - mean * a0) * rstd;  // sum(dy * xhat)
    const double m2 = g / f.n;
    f.k0[col] = (float)(w * rstd);
    f.k1[col] = (float)(-w * m2 * rstd * rstd);
    f.k2[col] = (float)(w * rstd * (-a0 / f.n + mean * m2 * rstd));
    if (f.dweight) f.dweight[col] = (float)g;
    if (f.dbias) f.dbias[col] = (float)a0;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_bn_dx(const float* __restrict__ x, const float* __restrict__ dy,
                                               const float* __restrict__ yr, int64_t nv, int c,
                                               const float* __restrict__ ka, const float* __restrict__ kb,
                                               const float* __restrict__ kc, float* __restrict__ dx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nv; i += stride) {
    const int col = (int)((i * VEC) % c);
    float a[VEC], d[VEC], m[VEC], o[VEC];
    bn_ld<VEC>(x + i * VEC, a);
    bn_ld<VEC>(dy + i * VEC, d);
    if (yr) bn_ld<VEC>(yr + i * VEC, m);
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      float g = (yr && !(m[u] > 0.f)) ? 0.f : d[u];
      o[u] = ka[col + u] * g + kb[col + u] * a[u] + kc[col + u];
    }
    if constexpr (VEC == 4)
      *(float4*)(dx + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
    else
      dx[i] = o[0];
  }
}

static inline size_t bn_partial_bytes(int c) { return (size_t)BN_MAX_BLOCKS * 2 * c * sizeof(double); }

extern "C" size_t pp_bn_train_workspace(int64_t n, int32_t c) {
  (void)n;
  if (c < 1) return 0;
  return bn_partial_bytes(c) + 4 * (size_t)c * sizeof(float) + 256;
}

template <int MODE>
static int bn_partial_launch(const float* x, const float* dy, const float* yr, int64_t n, int c, double* partial,
                             hipStream_t s, int* nblocks) {
  const bool v4 = (c % 4 == 0);
  const int cv = v4 ? c / 4 : c;
  const int64_t rps = 256 / cv;
  unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + rps * 4 - 1) / (rps * 4), BN_MAX_BLOCKS));
  if (v4)
    hipLaunchKernelGGL((k_bn_partial<4, MODE>), dim3(blocks), dim3(256), 0, s, x, dy, yr, n, c, partial);
  else
    hipLaunchKernelGGL((k_bn_partial<1, MODE>), dim3(blocks), dim3(256), 0, s, x, dy, yr, n, c, partial);
  PP_LAUNCH_CHECK();
  *nblocks = (int)blocks;
  return PP_OK;
}

extern "C" int pp_bn_train_fwd(const float* x, int64_t n, int32_t c, const float* weight, const float* bias, double eps,
                               double momentum, float* running_mean, float* running_var, int32_t relu, float* y,
                               double* save_mean, double* save_rstd, int64_t* num_batches_tracked, void* ws, size_t ws_bytes,
                               pp_stream_t stream) {
  PP_REQUIRE(c >= 1 && c <= 256, "pp_bn_train_fwd: c must be in [1,256]");
  PP_REQUIRE(n >= 1, "pp_bn_train_fwd: batch statistics need at least one row");
  PP_REQUIRE(x && y && save_mean && save_rstd && ws, "pp_bn_train_fwd: null pointer");
  PP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "pp_bn_train_fwd: running_mean/var come together");
  PP_REQUIRE(ws_bytes >= pp_bn_train_workspace(n, c), "pp_bn_train_fwd: workspace too small");
  hipStream_t s = pp_s(stream);
  double* partial = (double*)ws;
  float* coef = (float*)((char*)ws + bn_partial_bytes(c));
  int nb = 0;
  int rc = bn_partial_launch<0>(x, nullptr, nullptr, n, c, partial, s, &nb);
  if (rc != PP_OK) return rc;
  BnFinalize f{};
  f.partial = partial; f.nb = nb; f.c = c; f.backward = 0;
  f.n = (double)n; f.eps = eps; f.momentum = momentum;
  f.weight = weight; f.bias = bias; f.running_mean = running_mean; f.running_var = running_var;
  f.num_batches_tracked = num_batches_tracked;
  f.k0 = coef; f.k1 = coef + c; f.k2 = nullptr; f.save_mean = save_mean; f.save_rstd = save_rstd;
  hipLaunchKernelGGL(k_bn_finalize, dim3((c + 7) / 8), dim3(256), 0, s, f);
  PP_LAUNCH_CHECK();
  return pp_affine_act(x, n, c, coef, coef + c, relu ? 1 : 0, 0.f, nullptr, y, stream);
}

extern "C" int pp_bn_train_bwd(const float* x, const float* dy, const float* y_relu, int64_t n, int32_t c,
                               const float* weight, const double* save_mean, const double* save_rstd, float* dx,
                               float* dweight, float* dbias, void* ws, size_t ws_bytes, pp_stream_t stream) {
  PP_REQUIRE(c >= 1 && c <= 256, "pp_bn_train_bwd: c must be in [1,256]");
  PP_REQUIRE(n >= 1, "pp_bn_train_bwd: batch statistics need at least one row");
  PP_REQUIRE(x && dy && dx && save_mean && save_rstd && ws, "pp_bn_train_bwd: null pointer");
  PP_REQUIRE(ws_bytes >= pp_bn_train_workspace(n, c), "pp_bn_train_bwd: workspace too small");
  hipStream_t s = pp_s(stream);
  double* partial = (double*)ws;
  float* coef = (float*)((char*)ws + bn_partial_bytes(c));
  int nb = 0;
  int rc = y_relu ? bn_partial_launch<2>(x, dy, y_relu, n, c, partial, s, &nb)
                  : bn_partial_launch<1>(x, dy, nullptr, n, c, partial, s, &nb);
  if (rc != PP_OK) return rc;
  BnFinalize f{};
  f.partial = partial; f.nb = nb; f.c = c; f.backward = 1;
  f.n = (double)n; f.weight = weight;
  f.k0 = coef; f.k1 = coef + c; f.k2 = coef + 2 * c;
  f.save_mean = const_cast<double*>(save_mean); f.save_rstd = const_cast<double*>(save_rstd);
  f.dweight = dweight; f.dbias = dbias;
  hipLaunchKernelGGL(k_bn_finalize, dim3((c + 7) / 8), dim3(256), 0, s, f);
  PP_LAUNCH_CHECK();
  const int64_t total = n * c;
  if (c % 4 == 0) {
    unsigned blocks = (unsigned)std::min<int64_t>((total / 4 + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(k_bn_dx<4>, dim3(blocks), dim3(256), 0, s, x, dy, y_relu, total / 4, c, coef, coef + c,
                       coef + 2 * c, dx);
  } else {
    unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(k_bn_dx<1>, dim3(blocks), dim3(256), 0, s, x, dy, y_relu, total, c, coef, coef + c, coef + 2 * c,
                       dx);
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// fused head: Linear(no bias) -> folded BN -> LeakyReLU(0.2) -> Linear(+bias) [-> LogSoftmax] [-> argmax]
// one thread per point; weights broadcast from LDS; 64 B in, <= 64 B out per point.
// ---------------------------------------------------------------------------------------------
template <int CIN, int CHID, int COUT_MAX>
__global__ __launch_bounds__(256) void k_head_mlp(const float* __restrict__ x, int64_t n, const float* __restrict__ w1,
                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                  const float* __restrict__ w2, const float* __restrict__ b2, int cout,
                                                  int log_softmax, float* __restrict__ y, int64_t* __restrict__ argmax) {
  __shared__ float sw1[CHID * CIN];
  __shared__ float sw2[COUT_MAX * CHID];
  __shared__ float ssc[CHID], ssh[CHID], sb2[COUT_MAX];
  for (int t = threadIdx.x; t < CHID * CIN; t += blockDim.x) sw1[t] = w1[t];
  for (int t = threadIdx.x; t < COUT_MAX * CHID; t += blockDim.x) sw2[t] = t < cout * CHID ? w2[t] : 0.f;
  for (int t = threadIdx.x; t < CHID; t += blockDim.x) {
    ssc[t] = scale[t];
    ssh[t] = shift[t];
  }
  for (int t = threadIdx.x; t < COUT_MAX; t += blockDim.x) sb2[t] = (b2 && t < cout) ? b2[t] : 0.f;
  __syncthreads();
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xi[CIN];
#pragma unroll
  for (int u = 0; u < CIN / 4; ++u) {
    float4 v = *(const float4*)(x + i * CIN + 4 * u);
    xi[4 * u] = v.x; xi[4 * u + 1] = v.y; xi[4 * u + 2] = v.z; xi[4 * u + 3] = v.w;
  }
  float h[CHID];
#pragma unroll
  for (int a = 0; a < CHID; ++a) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < CIN; ++b) s += sw1[a * CIN + b] * xi[b];
    s = s * ssc[a] + ssh[a];
    h[a] = s < 0.f ? 0.2f * s : s;
  }
  float o[COUT_MAX];
#pragma unroll
  for (int a = 0; a < COUT_MAX; ++a) {
    float s = sb2[a];
#pragma unroll
    for (int b = 0; b < CHID; ++b) s += sw2[a * CHID + b] * h[b];
    o[a] = s;
  }
  int best = 0;
  float bv = o[0];
#pragma unroll
  for (int a = 1; a < COUT_MAX; ++a)
    if (a < cout && o[a] > bv) {
      bv = o[a];
      best = a;
    }
  if (log_softmax) {
    float se = 0.f;
#pragma unroll
    for (int a = 0; a < COUT_MAX; ++a)
      if (a < cout) se += expf(o[a] - bv);
    float lse = bv + logf(se);
#pragma unroll
    for (int a = 0; a < COUT_MAX; ++a) o[a] -= lse;
  }
#pragma unroll
  for (int a = 0; a < COUT_MAX; ++a)
    if (a < cout) y[i * cout + a] = o[a];
  if (argmax) argmax[i] = best;
}

extern "C" int pp_head_mlp(const float* x, int64_t n, int32_t cin, const float* w1, int32_t chid, const float* scale,
                           const float* shift, const float* w2, const float* b2, int32_t cout, int32_t log_softmax,
                           float* y, int64_t* argmax, pp_stream_t stream) {
  PP_REQUIRE(x && w1 && scale && shift && w2 && y, "pp_head_mlp: null pointer");
  PP_REQUIRE(cout >= 1 && cout <= 16, "pp_head_mlp: cout must be in [1,16]");
  if (n == 0) return PP_OK;
  dim3 grid(pp_blocks(n, 256));
  hipStream_t s = pp_s(stream);
  if (cin == 16 && chid == 16)
    hipLaunchKernelGGL((k_head_mlp<16, 16, 16>), grid, dim3(256), 0, s, x, n, w1, scale, shift, w2, b2, cout, log_softmax,
                       y, argmax);
  else if (cin == 32 && chid == 32)
    hipLaunchKernelGGL((k_head_mlp<32, 32, 16>), grid, dim3(256), 0, s, x, n, w1, scale, shift, w2, b2, cout, log_softmax,
                       y, argmax);
  else {
    pp_set_error("pp_head_mlp: unsupported (cin,chid)=(%d,%d); built for (16,16) and (32,32)", cin, chid);
    return PP_ERR_INVALID;
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// All heads of the model in one pass over the features: the row is read once (optionally through a row index: the
// backbone's output stays in the coordinate manager's internal order and index[i] names the row of caller point i, so the
// un-permuting gather of the features disappears too), every head's two small matrices come from LDS.
// Same arithmetic per head as k_head_mlp (same summation order: bit-identical outputs).
// ---------------------------------------------------------------------------------------------
#define HEADS_MAX 3
struct HeadDesc {  // device pointers of one head (pp_head_t in the public header)
  const float* w1;
  const float* scale;
  const float* shift;
  const float* w2;
  const float* b2;
  float* y;
  int64_t* argmax;
  int32_t cout;
  int32_t log_softmax;
};
struct HeadSet {
  HeadDesc h[HEADS_MAX];
};
template <int C, int COUT_MAX>
__global__ __launch_bounds__(256) void k_heads(const float* __restrict__ x, const int64_t* __restrict__ index, int64_t n,
                                               int64_t n_src, int n_heads, HeadSet hs, int32_t* __restrict__ err) {
  __shared__ float sw1[HEADS_MAX][C * C];
  __shared__ float sw2[HEADS_MAX][COUT_MAX * C];
  __shared__ float ssc[HEADS_MAX][C], ssh[HEADS_MAX][C], sb2[HEADS_MAX][COUT_MAX];
  for (int hh = 0; hh < n_heads; ++hh) {
    const HeadDesc& d = hs.h[hh];
    for (int t = threadIdx.x; t < C * C; t += blockDim.x) sw1[hh][t] = d.w1[t];
    for (int t = threadIdx.x; t < COUT_MAX * C; t += blockDim.x) sw2[hh][t] = t < d.cout * C ? d.w2[t] : 0.f;
    for (int t = threadIdx.x; t < C; t += blockDim.x) {
      ssc[hh][t] = d.scale[t];
      ssh[hh][t] = d.shift[t];
    }
    for (int t = threadIdx.x; t < COUT_MAX; t += blockDim.x) sb2[hh][t] = (d.b2 && t < d.cout) ? d.b2[t] : 0.f;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t r = index ? index[i] : i;
  if (r < 0 || r >= n_src) {
    atomicAdd(err, 1);
    r = 0;
  }
  float xi[C];
#pragma unroll
  for (int u = 0; u < C / 4; ++u) {
    const float4 v = *(const float4*)(x + r * C + 4 * u);
    xi[4 * u] = v.x; xi[4 * u + 1] = v.y; xi[4 * u + 2] = v.z; xi[4 * u + 3] = v.w;
  }
#pragma unroll
  for (int hh = 0; hh < HEADS_MAX; ++hh) {
    if (hh >= n_heads) break;
    const HeadDesc& d = hs.h[hh];
    float h[C];
#pragma unroll
    for (int a = 0; a < C; ++a) {
      float s = 0.f;
#pragma unroll
      for (int b = 0; b < C; ++b) s += sw1[hh][a * C + b] * xi[b];
      s = s * ssc[hh][a] + ssh[hh][a];
      h[a] = s < 0.f ? 0.2f * s : s;
    }
    float o[COUT_MAX];
#pragma unroll
    for (int a = 0; a < COUT_MAX; ++a) {
      float s = sb2[hh][a];
#pragma unroll
      for (int b = 0; b < C; ++b) s += sw2[hh][a * C + b] * h[b];
      o[a] = s;
    }
    const int cout = d.cout;
    int best = 0;
    float bv = o[0];
#pragma unroll
    for (int a = 1; a < COUT_MAX; ++a)
      if (a < cout && o[a] > bv) {
        bv = o[a];
        best = a;
      }
    if (d.log_softmax) {
      float se = 0.f;
#pragma unroll
      for (int a = 0; a < COUT_MAX; ++a)
        if (a < cout) se += expf(o[a] - bv);
      const float lse = bv + logf(se);
#pragma unroll
      for (int a = 0; a < COUT_MAX; ++a) o[a] -= lse;
    }
#pragma unroll
    for (int a = 0; a < COUT_MAX; ++a)
      if (a < cout) d.y[i * cout + a] = o[a];
    if (d.argmax) d.argmax[i] = best;
  }
}

extern "C" int pp_heads(const float* x, int64_t n_src, int32_t c, const int64_t* index, int64_t n, const pp_head_t* heads,
                        int32_t n_heads, int32_t* err_flag, pp_stream_t stream) {
  PP_REQUIRE(n == 0 || (x && heads && err_flag), "pp_heads: null pointer");
  PP_REQUIRE(n_heads >= 1 && n_heads <= HEADS_MAX, "pp_heads: 1 to 3 heads");
  PP_REQUIRE(c == 16, "pp_heads: built for 16 feature channels (hidden width = channels)");
  static_assert(sizeof(pp_head_t) == sizeof(HeadDesc), "pp_head_t layout");
  if (n == 0) return PP_OK;
  HeadSet hs;
  for (int hh = 0; hh < n_heads; ++hh) {
    PP_REQUIRE(heads[hh].w1 && heads[hh].scale && heads[hh].shift && heads[hh].w2 && heads[hh].y, "pp_heads: null head tensor");
    PP_REQUIRE(heads[hh].cout >= 1 && heads[hh].cout <= 16, "pp_heads: cout must be in [1,16]");
    memcpy(&hs.h[hh], &heads[hh], sizeof(HeadDesc));
  }
  hipLaunchKernelGGL((k_heads<16, 16>), dim3(pp_blocks(n, 256)), dim3(256), 0, pp_s(stream), x, index, n, n_src, n_heads, hs, err_flag);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// row gather out[i] = src[index[i]] for [n,c] float32 with c % 4 == 0: one 16-byte chunk per thread, so a row is read
// and written as whole 16 B segments (torch's generic gather kernel reaches ~0.6 TB/s on 64 B rows; this is the
// caller-order <-> internal-order permutation of the features and the proposal-row gather of the scorer)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_rows16(const float4* __restrict__ src, const int64_t* __restrict__ index,
                                                       int64_t n_chunks, int cpr, int64_t n_src, float4* __restrict__ out,
                                                       int32_t* err) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < n_chunks; e += stride) {
    const int64_t i = e / cpr;
    const int w = (int)(e - i * cpr);
    const int64_t r = index[i];
    if (r < 0 || r >= n_src) {
      if (w == 0) atomicAdd(err, 1);
      continue;
    }
    out[e] = src[r * cpr + w];
  }
}

extern "C" int pp_gather_rows(const float* src, int64_t n_src, int32_t c, const int64_t* index, int64_t n, float* out,
                              int32_t* err_flag, pp_stream_t stream) {
  PP_REQUIRE(c >= 4 && c % 4 == 0, "pp_gather_rows: c must be a positive multiple of 4");
  PP_REQUIRE(n == 0 || (src && index && out && err_flag), "pp_gather_rows: null pointer");
  if (n == 0) return PP_OK;
  const int cpr = c / 4;
  const int64_t chunks = n * cpr;
  hipLaunchKernelGGL(k_gather_rows16, dim3((unsigned)std::min<int64_t>(pp_blocks(chunks, 256), 1 << 20)), dim3(256), 0,
                     pp_s(stream), (const float4*)src, index, chunks, cpr, n_src, (float4*)out, err_flag);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// K11 segment reductions: sum / mean via float atomics on the output row, max via an ordered-int atomicMax.
// index need not be sorted (torch_scatter semantics).  Empty segments -> 0.
// ---------------------------------------------------------------------------------------------
__device__ inline int f2ord(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ inline float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ __launch_bounds__(256) void k_seg_count(const int64_t* __restrict__ index, int64_t n, int64_t n_seg,
                                                   int32_t* cnt, int32_t* err) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t s = index[i];
  if (s < 0 || s >= n_seg) {
    atomicAdd(err, 1);
    return;
  }
  atomicAdd(&cnt[s], 1);
}
__global__ __launch_bounds__(256) void k_seg_accum(const float* __restrict__ src, const int64_t* __restrict__ index,
                                                   int64_t total, int c, int64_t n_seg, int reduce, float* out,
                                                   int* out_ord) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int64_t i = e / c;
  int j = (int)(e % c);
  int64_t s = index[i];
  if (s < 0 || s >= n_seg) return;
  if (reduce == 2)
    atomicMax(&out_ord[s * c + j], f2ord(src[e]));
  else
    atomicAdd(&out[s * c + j], src[e]);
}
// few segments (n_seg * (c + 1) words fit in LDS): block-private accumulation, then one global atomic per touched
// segment entry and block -- thousands of rows hammering a handful of output addresses is what made the
// discriminative-loss reductions slow (a14).
#define SEG_LDS_WORDS 12288
#define SEG_LDS_ELEMS_PER_THREAD 32
__global__ __launch_bounds__(256) void k_seg_accum_lds(const float* __restrict__ src, const int64_t* __restrict__ index,
                                                       int64_t total, int c, int n_seg, int reduce, float* out,
                                                       int* out_ord, int32_t* cnt, int32_t* err) {
  extern __shared__ int seg_sh[];
  int* shv = seg_sh;                // [n_seg * c] float bits (sum) or ordered ints (max)
  int* shc = seg_sh + n_seg * c;    // [n_seg]
  const int nv = n_seg * c;
  for (int t = threadIdx.x; t < nv; t += 256) shv[t] = reduce == 2 ? (int)0x80000000 : 0;
  for (int t = threadIdx.x; t < n_seg; t += 256) shc[t] = 0;
  __syncthreads();
  const int64_t e0 = (int64_t)blockIdx.x * (256 * SEG_LDS_ELEMS_PER_THREAD);
#pragma unroll 4
  for (int u = 0; u < SEG_LDS_ELEMS_PER_THREAD; ++u) {
    const int64_t e = e0 + u * 256 + threadIdx.x;
    if (e >= total) break;
    const int64_t i = e / c;
    const int j = (int)(e - i * c);
    const int64_t sg = index[i];
    if (sg < 0 || sg >= n_seg) {
      if (j == 0) atomicAdd(err, 1);
      continue;
    }
    if (reduce == 2)
      atomicMax(&shv[sg * c + j], f2ord(src[e]));
    else
      atomicAdd((float*)&shv[sg * c + j], src[e]);
    if (j == 0) atomicAdd(&shc[sg], 1);
  }
  __syncthreads();
  // flush what this block touched (judged by value: a row may straddle two blocks, so the count, taken at column 0,
  // says nothing about the other columns)
  for (int t = threadIdx.x; t < nv; t += 256) {
    const int v = shv[t];
    if (reduce == 2) {
      if (v != (int)0x80000000) atomicMax(&out_ord[t], v);
    } else if (v != 0) {
      atomicAdd(&out[t], __int_as_float(v));
    }
  }
  for (int t = threadIdx.x; t < n_seg; t += 256)
    if (shc[t]) atomicAdd(&cnt[t], shc[t]);
}

// many segments: a thread owns one column of SEG_RUN consecutive rows and keeps a running value while the segment id
// does not change, so (nearly) sorted ids -- per-proposal maxima of the scorer, whose rows are ordered by proposal --
// cost one atomic per run instead of one per element.  Unsorted ids stay correct, just with shorter runs.
#define SEG_RUN 16
__global__ __launch_bounds__(256) void k_seg_accum_runs(const float* __restrict__ src, const int64_t* __restrict__ index,
                                                        int64_t n, int c, int64_t n_seg, int reduce, float* out,
                                                        int* out_ord, int32_t* cnt, int32_t* err) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t chunk = e / c;
  const int j = (int)(e - chunk * c);
  const int64_t r0 = chunk * SEG_RUN;
  if (r0 >= n) return;
  const int64_t r1 = r0 + SEG_RUN < n ? r0 + SEG_RUN : n;
  int64_t cur = -1;
  float acc = 0.f;
  int run = 0;
  for (int64_t r = r0; r < r1; ++r) {
    const int64_t sg = index[r];
    if (sg != cur) {
      if (run) {
        if (reduce == 2) atomicMax(&out_ord[cur * c + j], f2ord(acc)); else atomicAdd(&out[cur * c + j], acc);
        if (j == 0) atomicAdd(&cnt[cur], run);
      }
      run = 0;
      cur = sg;
      if (sg < 0 || sg >= n_seg) {
        cur = -1;
        if (j == 0) atomicAdd(err, 1);
        continue;
      }
    } else if (cur < 0) {
      if (j == 0) atomicAdd(err, 1);
      continue;
    }
    const float v = src[r * c + j];
    acc = run == 0 ? v : (reduce == 2 ? fmaxf(acc, v) : acc + v);
    ++run;
  }
  if (run) {
    if (reduce == 2) atomicMax(&out_ord[cur * c + j], f2ord(acc)); else atomicAdd(&out[cur * c + j], acc);
    if (j == 0) atomicAdd(&cnt[cur], run);
  }
}

__global__ __launch_bounds__(256) void k_seg_finish(float* out, int* out_ord, const int32_t* __restrict__ cnt,
                                                    int64_t total, int c, int reduce) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int32_t k = cnt[e / c];
  if (reduce == 2)
    out[e] = k > 0 ? ord2f(out_ord[e]) : 0.f;
  else if (reduce == 1 && k > 0)
    out[e] = out[e] / (float)k;
}
__global__ __launch_bounds__(256) void k_seg_argmax(const float* __restrict__ src, const int64_t* __restrict__ index,
                                                    int64_t total, int c, int64_t n_seg, const float* __restrict__ out,
                                                    unsigned long long* arg) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int64_t i = e / c;
  int j = (int)(e % c);
  int64_t s = index[i];
  if (s < 0 || s >= n_seg) return;
  if (src[e] == out[s * c + j]) atomicMin(&arg[s * c + j], (unsigned long long)i);  // first arg-max row
}

extern "C" size_t pp_segment_reduce_workspace(int64_t n_seg) { return pp_align(sizeof(int32_t) * (size_t)(n_seg + 2)); }

static int segment_reduce_impl(const float* src, const int64_t* index, int64_t n, int32_t c, int64_t n_seg,
                               int32_t reduce, float* out, int64_t* arg, void* workspace, size_t workspace_bytes,
                               bool check, pp_stream_t stream) {
  PP_REQUIRE(out && (n == 0 || (src && index)), "pp_segment_reduce: null pointer");
  PP_REQUIRE(reduce >= 0 && reduce <= 2, "pp_segment_reduce: reduce must be 0 (sum), 1 (mean) or 2 (max)");
  hipStream_t s = pp_s(stream);
  int64_t total_out = n_seg * c;
  if (total_out == 0) return PP_OK;
  if (workspace_bytes < pp_segment_reduce_workspace(n_seg)) return PP_ERR_WORKSPACE;
  int32_t* cnt = (int32_t*)workspace;
  PP_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(n_seg + 1), s));
  if (reduce == 2) {
    // fill with ordered-int minimum
    PP_HIP(hipMemsetD32Async((hipDeviceptr_t)out, (int)0x80000000, (size_t)total_out, s));
  } else {
    PP_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)total_out, s));
  }
  if (n > 0 && n_seg * (c + 1) <= SEG_LDS_WORDS && n * c >= 65536) {
    hipLaunchKernelGGL(k_seg_accum_lds, dim3(pp_blocks(n * c, 256 * SEG_LDS_ELEMS_PER_THREAD)), dim3(256),
                       sizeof(int) * (size_t)(n_seg * (c + 1)), s, src, index, n * c, c, (int)n_seg, reduce, out,
                       (int*)out, cnt, cnt + n_seg);
  } else if (n > 0 && c >= 4) {
    const int64_t threads = ((n + SEG_RUN - 1) / SEG_RUN) * c;
    hipLaunchKernelGGL(k_seg_accum_runs, dim3(pp_blocks(threads, 256)), dim3(256), 0, s, src, index, n, c, n_seg, reduce,
                       out, (int*)out, cnt, cnt + n_seg);
  } else if (n > 0) {
    hipLaunchKernelGGL(k_seg_count, dim3(pp_blocks(n, 256)), dim3(256), 0, s, index, n, n_seg, cnt, cnt + n_seg);
    hipLaunchKernelGGL(k_seg_accum, dim3(pp_blocks(n * c, 256)), dim3(256), 0, s, src, index, n * c, c, n_seg, reduce,
                       out, (int*)out);
  }
  hipLaunchKernelGGL(k_seg_finish, dim3(pp_blocks(total_out, 256)), dim3(256), 0, s, out, (int*)out, cnt, total_out, c,
                     reduce);
  if (arg) {
    PP_HIP(hipMemsetAsync(arg, 0xFF, sizeof(int64_t) * (size_t)total_out, s));
    if (reduce == 2 && n > 0)
      hipLaunchKernelGGL(k_seg_argmax, dim3(pp_blocks(n * c, 256)), dim3(256), 0, s, src, index, n * c, c, n_seg, out,
                         (unsigned long long*)arg);
  }
  PP_LAUNCH_CHECK();
  if (!check) return PP_OK;
  int32_t bad = 0;
  PP_HIP(hipMemcpyAsync(&bad, cnt + n_seg, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PP_HIP(hipStreamSynchronize(s));
  if (bad) {
    pp_set_error("pp_segment_reduce: %d index values outside [0,%lld)", bad, (long long)n_seg);
    return PP_ERR_INVALID;
  }
  return PP_OK;
}

extern "C" int pp_segment_reduce(const float* src, const int64_t* index, int64_t n, int32_t c, int64_t n_seg,
                                 int32_t reduce, float* out, int64_t* arg, void* workspace, size_t workspace_bytes,
                                 pp_stream_t stream) {
  return segment_reduce_impl(src, index, n, c, n_seg, reduce, out, arg, workspace, workspace_bytes, true, stream);
}
extern "C" int pp_segment_reduce_unchecked(const float* src, const int64_t* index, int64_t n, int32_t c, int64_t n_seg,
                                           int32_t reduce, float* out, int64_t* arg, void* workspace,
                                           size_t workspace_bytes, pp_stream_t stream) {
  return segment_reduce_impl(src, index, n, c, n_seg, reduce, out, arg, workspace, workspace_bytes, false, stream);
}

// ---- skinny Linear layer, forward and input gradient ----------------------------------------------------------------------------
// y[r][o] = b[o] + sum_i x[r][i] * W(o, i), W(o, i) = w[o * cin + i] (y = x W^T: torch.nn.Linear's forward) or w[i * cout + o]
// (transposed = 1: y = x W, the input gradient dx = dy W of a layer whose weight is [cin_of_dx... i.e. [rows of dy's columns]).
// cin, cout <= 32 and hundreds of thousands of rows: a thread per row, the weights in LDS (read as broadcasts), a fixed summation
// order (i ascending) -- no library GEMM on the training step's path.
__global__ __launch_bounds__(256) void k_linear_rows(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                     int64_t n, int cin, int cout, int transposed, float* __restrict__ y) {
  __shared__ float wl[32 * 33];
  __shared__ float bl[32];
  for (int t = threadIdx.x; t < cin * cout; t += 256) {
    const int o = transposed ? t % cout : t / cin, i = transposed ? t / cout : t % cin;
    wl[o * 33 + i] = w[t];
  }
  if (threadIdx.x < cout) bl[threadIdx.x] = b ? b[threadIdx.x] : 0.f;
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  float xr[32];
  const float* xp = x + r * cin;
#pragma unroll
  for (int i = 0; i < 32; ++i) xr[i] = i < cin ? xp[i] : 0.f;
  float* yp = y + r * cout;
  for (int o = 0; o < cout; ++o) {
    float acc = bl[o];
#pragma unroll
    for (int i = 0; i < 32; ++i)
      if (i < cin) acc += xr[i] * wl[o * 33 + i];
    yp[o] = acc;
  }
}
extern "C" int pp_linear_rows(const float* x, const float* weight, const float* bias, int64_t n, int32_t cin, int32_t cout,
                              int32_t transposed, float* y, pp_stream_t stream) {
  PP_REQUIRE(cin >= 1 && cin <= 32 && cout >= 1 && cout <= 32, "pp_linear_rows: cin and cout must be in [1,32]");
  if (n <= 0) return PP_OK;
  PP_REQUIRE(x && weight && y, "pp_linear_rows: null pointer");
  hipLaunchKernelGGL(k_linear_rows, dim3(pp_blocks(n, 256)), dim3(256), 0, pp_s(stream), x, weight, bias, n, (int)cin, (int)cout,
                     (int)transposed, y);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- segment sum / mean without atomics (bit-reproducible run to run) -------------------------------------------------------
// The rows of segment s are rows[offs[s] .. offs[s+1]) in ascending row order (pp_group_by_key: stable).  One workgroup per
// segment: 256 / cw row lanes x cw columns (cw = min(c, 256)); row lane r adds rows r, r + R, ... in order, the lanes' partials
// are added in lane order (float64 partials, rounded once).  The summation order is a function of (segment size, c) alone.
__global__ __launch_bounds__(256) void k_seg_sum_ordered(const float* __restrict__ src, const int64_t* __restrict__ rows,
                                                         const int32_t* __restrict__ offs, int c, int mean, float* __restrict__ out) {
  __shared__ double sh[256];
  const int64_t s = blockIdx.x;
  const int lo = offs[s], hi = offs[s + 1];
  const int cw = c < 256 ? c : 256;
  const int R = 256 / cw;
  const int t = threadIdx.x, jj = t % cw, r = t / cw;
  for (int col0 = 0; col0 < c; col0 += cw) {
    const int j = col0 + jj;
    double acc = 0.0;  // float64 partials: a lane may add 10^5 rows one after the other
    if (r < R && j < c)
      for (int e = lo + r; e < hi; e += R) acc += (double)src[rows[e] * c + j];
    __syncthreads();  // (sh is read below in the previous column chunk)
    sh[t] = acc;
    __syncthreads();
    if (r == 0 && j < c) {
      double tot = sh[jj];
      for (int q = 1; q < R; ++q) tot += sh[q * cw + jj];
      out[s * c + j] = (float)((mean && hi > lo) ? tot / (double)(hi - lo) : tot);
    }
  }
}
extern "C" int pp_segment_sum_ordered(const float* src, const int64_t* rows, const int32_t* offsets, int64_t n_seg, int32_t c,
                                      int32_t mean, float* out, pp_stream_t stream) {
  PP_REQUIRE(c >= 1, "pp_segment_sum_ordered: c must be positive");
  if (n_seg <= 0) return PP_OK;
  PP_REQUIRE(offsets && out, "pp_segment_sum_ordered: null pointer");  // (src / rows may be NULL when every segment is empty)
  hipLaunchKernelGGL(k_seg_sum_ordered, dim3((unsigned)n_seg), dim3(256), 0, pp_s(stream), src, rows, offsets, (int)c, (int)mean, out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
