// K6 batch-norm pieces on [n,C] features, fused MLP heads, K11 segment reductions.  All HBM-bound:
// float4 / row-contiguous accesses, one pass over the data.
// References: ME.MinkowskiBatchNorm (= BatchNorm1d on F) torch_points3d/modules/MinkowskiEngine/api_modules.py:40;
// heads torch_points3d/models/panoptic/PointGroup3heads.py:69-81,106-108 and
// torch_points3d/core/common_modules/base_modules.py:35-45; torch_scatter.scatter PointGroup3heads.py:419-452.
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

extern "C" int pp_segment_reduce(const float* src, const int64_t* index, int64_t n, int32_t c, int64_t n_seg,
                                 int32_t reduce, float* out, int64_t* arg, void* workspace, size_t workspace_bytes,
                                 pp_stream_t stream) {
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
  if (n > 0) {
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
  int32_t bad = 0;
  PP_HIP(hipMemcpyAsync(&bad, cnt + n_seg, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PP_HIP(hipStreamSynchronize(s));
  if (bad) {
    pp_set_error("pp_segment_reduce: %d index values outside [0,%lld)", bad, (long long)n_seg);
    return PP_ERR_INVALID;
  }
  return PP_OK;
}
