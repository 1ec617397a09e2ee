// How fast is a 16-row x 64-byte tile gather when the rows sit in LDS (a per-workgroup row cache) instead of L1 / L2?
// The thin convolution layers issue one L2 request per gathered row (profiles/r02_pmc_tcc.md) although a 256-row workgroup
// touches only 1.6 - 2.1 distinct input rows per output row: staging those once in LDS would turn 3 of 4 gathers into
// ds_read_b128.  Lane (i, q) reads 16 bytes at rows[i] * 64 + q * 16 -- the A-operand layout of k_spconv_fwd3.
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/lds_gather.hip -o /tmp/ldsg && /tmp/ldsg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ROWS>  // rows resident in LDS (64 B each)
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ rows, int iters, float* out) {
  __shared__ f32x4 cache[ROWS * 4];
  for (int t = threadIdx.x; t < ROWS * 4; t += 256) cache[t] = (f32x4){(float)t, 1.f, 2.f, 3.f};
  __syncthreads();
  const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned r[4];
  for (int t = 0; t < 4; ++t) r[t] = rows[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + t) * 16 + i] % ROWS;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned row = (r[t] + (unsigned)it * 37u) % ROWS;  // a different row every time, same distribution
      acc += cache[row * 4 + q];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

int main() {
  float* out; unsigned* rows;
  (void)hipMalloc(&out, 4);
  const int blocks = 256 * 8;
  std::vector<unsigned> h(blocks * 4 * 4 * 16);
  for (size_t t = 0; t < h.size(); ++t) h[t] = (unsigned)((t * 2654435761u) >> 8);
  (void)hipMalloc(&rows, h.size() * 4);
  (void)hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<460>, dim3(blocks), dim3(256), 0, 0, rows, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double loads = (double)blocks * 4 * iters * 4;  // wave-level gathers
    if (rep) printf("16 random rows x 64 B from a 460-row LDS cache: %.3f ms, %.1f cycles per wave gather per CU, %.1f TB/s\n", ms,
                    ms * 1e6 / (loads / 256) * 2.4, loads * 1024.0 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
