// Library plumbing: error reporting, version, device scan / radix-sort wrappers (rocPRIM via hipCUB headers),
// and the HBM triad used by bench.py to confirm the roofline denominator.
#include <hipcub/hipcub.hpp>
#include <stdarg.h>

#include "pp_common.h"

static thread_local char g_err[512] = "";

void pp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pp_last_error(void) { return g_err; }
extern "C" const char* pp_version(void) { return "panoptic_hip 1 gfx950"; }

__global__ __launch_bounds__(256) void k_triad(float4* __restrict__ a, const float4* __restrict__ b,
                                               const float4* __restrict__ c, float s, int64_t n4) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 x = b[i], y = c[i];
    a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
  }
}
extern "C" int pp_triad(float* a, const float* b, const float* c, float s, int64_t n, pp_stream_t stream) {
  PP_REQUIRE(n % 4 == 0, "pp_triad: n must be a multiple of 4");
  int64_t n4 = n / 4;
  unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 256 * 64);
  if (blocks == 0) return PP_OK;
  hipLaunchKernelGGL(k_triad, dim3(blocks), dim3(256), 0, pp_s(stream), (float4*)a, (const float4*)b,
                     (const float4*)c, s, n4);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---------------------------------------------------------------------------------------------
size_t pp_scan_workspace(int64_t n) {
  size_t bytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int)std::max<int64_t>(n, 1));
  return pp_align(bytes) + 256;
}

__global__ void k_scan_total(const int32_t* in, const int32_t* out, int64_t n, int32_t* total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) total[0] = n > 0 ? out[n - 1] + in[n - 1] : 0;
}

int pp_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total, void* ws, size_t ws_bytes,
                          hipStream_t stream) {
  if (n > 0) {
    size_t bytes = ws_bytes;
    PP_HIP(hipcub::DeviceScan::ExclusiveSum(ws, bytes, in, out, (int)n, stream));
  }
  if (total) {
    hipLaunchKernelGGL(k_scan_total, dim3(1), dim3(64), 0, stream, in, out, n, total);
    PP_LAUNCH_CHECK();
  }
  return PP_OK;
}

size_t pp_sort_pairs_workspace(int64_t n) {
  size_t b64 = 0, b32 = 0;
  int nn = (int)std::max<int64_t>(n, 1);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b64, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr,
                                     (int32_t*)nullptr, nn, 0, 64);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b32, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                     (int32_t*)nullptr, nn, 0, 32);
  return pp_align(std::max(b64, b32)) + 256;
}

int pp_sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                      int64_t n, int end_bit, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n <= 0) return PP_OK;
  size_t bytes = ws_bytes;
  PP_HIP(hipcub::DeviceRadixSort::SortPairs(ws, bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, stream));
  return PP_OK;
}
int pp_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                      int64_t n, int end_bit, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n <= 0) return PP_OK;
  size_t bytes = ws_bytes;
  PP_HIP(hipcub::DeviceRadixSort::SortPairs(ws, bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, stream));
  return PP_OK;
}
