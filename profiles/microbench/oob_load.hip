// What does a buffer load cost when all of its lanes are out of range?  k_spconv_fwd3 issues the A gather of every 16-row
// tile of the wave for every offset of the wave's union; tiles that lack the offset carry the byte offset 0xFFFFFFFF in all
// lanes (hardware bounds check -> zeros).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/oob_load.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 3: as 0 without the walk (the rows stay where `offs` puts them); 0: in-range 64 B rows (16 rows per load, like the gather), 1: all lanes out of range, 2: half of the loads out of range
__global__ __launch_bounds__(256) void k(const float* src, unsigned bytes, const unsigned* offs, int iters, float* out,
                                         unsigned region_mask = 0xFFFFFFFFu, int rows_in = 16) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
  const int lane = threadIdx.x & 63;
  const unsigned q16 = (unsigned)(lane >> 4) * 16u;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned o[4];
  for (int t = 0; t < 4; ++t) o[t] = offs[(blockIdx.x * 4 + t) * 16 + (lane & 15)] | q16;
  for (int it = 0; it < iters; ++it) {
    f32x4 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      unsigned off = o[t] + (unsigned)it * 64u * 1024u;  // walk through the buffer
      if (MODE == 3) off = (((o[t] & ~63u) + (unsigned)it * 4160u) & region_mask & ~63u) | q16;  // stay inside the region
      if (MODE == 4 && ((lane & 15) >= rows_in)) off = 0xFFFFFFFFu;  // only the first rows_in of the 16 rows are in range
      if (MODE == 1 || (MODE == 2 && (t & 1))) off = 0xFFFFFFFFu;
      v[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc += v[t];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

int main() {
  const size_t bytes = 256u << 20;
  float* src; float* out; unsigned* offs;
  hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&out, 4);
  const int blocks = 256 * 16;
  std::vector<unsigned> h(blocks * 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)((i * 2654435761u) % (1u << 20)) * 64u;  // scattered 64 B rows in the first 64 MB
  hipMalloc(&offs, h.size() * 4); hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"in range (scattered 64 B rows)", "all lanes out of range", "every second load out of range"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, src, (unsigned)bytes, offs, iters, out);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, src, (unsigned)bytes, offs, iters, out);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, src, (unsigned)bytes, offs, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double loads = (double)blocks * 4 * iters * 4;  // wave-level load instructions
      if (rep) printf("%-34s %8.3f ms  %.2f ns per wave load per CU  (%.1f cycles at 2.4 GHz)\n", names[mode], ms,
                      ms * 1e6 / (loads / 256), ms * 1e6 / (loads / 256) * 2.4);
    }
  }
  // the same gather with the rows confined to a region that stays in every XCD's L2 (2 MB) or in L1 (16 KB)
  for (int region_kb : {16, 2048, 65536}) {
    for (size_t i = 0; i < h.size(); ++i) {
      const unsigned rows = (unsigned)region_kb * 1024u / 64u;
      h[i] = (unsigned)((i * 2654435761u) % rows) * 64u;
    }
    hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, src, (unsigned)bytes, offs, iters, out,
                         (unsigned)region_kb * 1024u - 1u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double loads = (double)blocks * 4 * iters * 4;
      if (rep) printf("rows inside %6d KB, no walk:       %8.3f ms  %.1f cycles per wave load per CU, %.2f TB/s of 64 B rows\n", region_kb, ms,
                      ms * 1e6 / (loads / 256) * 2.4, loads * 1024.0 / (ms * 1e-3) / 1e12);
    }
  }
  // partially filled gathers (a 16-row tile in which only some rows have the neighbour): is the cost per row or per load?
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)((i * 2654435761u) % (1u << 15)) * 64u;  // 2 MB: L2-resident
  hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int rows_in : {16, 8, 4, 2, 1}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, src, (unsigned)bytes, offs, iters, out, 0xFFFFFFFFu, rows_in);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double loads = (double)blocks * 4 * iters * 4;
      if (rep) printf("%2d of 16 rows in range (walking 64 MB): %8.3f ms  %.1f cycles per wave load per CU\n", rows_in, ms,
                      ms * 1e6 / (loads / 256) * 2.4);
    }
  }
  return 0;
}
