// Ground truth for the roofline denominator: issue rate of v_mfma_f32_16x16x4_f32 (and 32x32x2) with no memory traffic.
// build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256, 2) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float av = a + threadIdx.x, bv = b + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256, 2) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  float av = a + threadIdx.x, bv = b + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 16 MFMAs (4 chains) + NV independent VALU ops (v_fma_f32 on a private chain) per iteration: do they overlap?
template <int NV>
__global__ __launch_bounds__(256, 2) void kmix(float* out, int iters, float a, float b) {
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float av = a + threadIdx.x, bv = b + threadIdx.x;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = a * e;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int v = 0; v < NV; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], 1.0001f, bv);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
#pragma unroll
  for (int e = 0; e < 8; ++e) s += x[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 32 MFMAs per iteration in the convolution kernel's pattern: 2 row tiles x 4 column tiles x 4 k-slices, every
// instruction with its own A / B registers (8 + 16 distinct loop-invariant registers), 8 accumulator chains
__global__ __launch_bounds__(256, 2) void kpattern(float* out, int iters, float a, float b) {
  f32x4 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 A[2], B[4];
#pragma unroll
  for (int t = 0; t < 2; ++t) A[t] = (f32x4){a + t, a * 2 + threadIdx.x, a - t, a + 3.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) B[c] = (f32x4){b + c, b * 2 + threadIdx.x, b - c, b + 5.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[t][u], B[c][u], acc[t][c], 0, 0, 0);
    asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]));
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) s += acc[t][c][0] + acc[t][c][1] + acc[t][c][2] + acc[t][c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 16 MFMAs + NS scalar ALU ops per iteration
template <int NS>
__global__ __launch_bounds__(256, 2) void ksalu(float* out, int iters, float a, float b, int seed) {
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float av = a + threadIdx.x, bv = b + threadIdx.x;
  int sx = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int v = 0; v < NS; ++v) {
      sx = sx * 3 + 7;
      asm volatile("" : "+s"(sx));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
  }
  float s = sx;
#pragma unroll
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 16 MFMAs + NL 1-KiB buffer loads (L2-resident, consumed one iteration later) per iteration
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NL>
__global__ __launch_bounds__(256, 2) void kload(float* out, const float* src, int iters, float a, float b) {
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 24, 0x00020000);
  const unsigned lane16 = (threadIdx.x & 63) * 16;
  f32x4 cur[NL], nxt[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) cur[l] = (f32x4){a, b, a, b};
  unsigned so = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int l = 0; l < NL; ++l)
      nxt[l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane16, (int)(so + l * 1024), 0));
    so = (so + NL * 1024) & ((1 << 22) - 1);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[c % NL][u], cur[(c + 1) % NL][u], acc[c], 0, 0, 0);
#pragma unroll
    for (int l = 0; l < NL; ++l) cur[l] = nxt[l];
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flops_per_launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error in %s\n", name);
  printf("%-40s %8.3f ms/launch  %7.1f TFLOP/s\n", name, ms / 5, flops_per_launch / (ms / 5 * 1e-3) / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4096 * 4);
  float* src;
  hipMalloc(&src, 1 << 24);
  hipMemset(src, 0, 1 << 24);
  const int iters = 20000;
  for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2) {
    const int blocks = 256 * blocks_per_cu;
    const double waves = blocks * 4.0;
    printf("blocks per CU %d (waves per SIMD %d)\n", blocks_per_cu, blocks_per_cu);
    run("16x16x4 f32, 1 chain", [&] { k16<1><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 4.0 * 1 * 2048);
    run("16x16x4 f32, 2 chains", [&] { k16<2><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 4.0 * 2 * 2048);
    run("16x16x4 f32, 4 chains", [&] { k16<4><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 4.0 * 4 * 2048);
    run("16x16x4 f32, 8 chains", [&] { k16<8><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 4.0 * 8 * 2048);
    run("conv pattern: 32 mfma, 24 operand regs", [&] { kpattern<<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 32.0 * 2048);
    run("16 mfma + 32 valu / iter", [&] { kmix<32><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 16.0 * 2048);
    run("16 mfma + 64 valu / iter", [&] { kmix<64><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 16.0 * 2048);
    run("16 mfma + 128 valu / iter", [&] { kmix<128><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 16.0 * 2048);
    run("16 mfma + 32x2 salu / iter", [&] { ksalu<32><<<blocks, 256>>>(out, iters, 1.f, 2.f, 5); }, waves * iters * 16.0 * 2048);
    run("16 mfma + 64x2 salu / iter", [&] { ksalu<64><<<blocks, 256>>>(out, iters, 1.f, 2.f, 5); }, waves * iters * 16.0 * 2048);
    run("16 mfma + 3 buffer loads / iter", [&] { kload<3><<<blocks, 256>>>(out, src, iters, 1.f, 2.f); }, waves * iters * 16.0 * 2048);
    run("16 mfma + 6 buffer loads / iter", [&] { kload<6><<<blocks, 256>>>(out, src, iters, 1.f, 2.f); }, waves * iters * 16.0 * 2048);
    run("32x32x2 f32, 2 chains", [&] { k32<2><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 4.0 * 2 * 4096);
    run("32x32x2 f32, 4 chains", [&] { k32<4><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, waves * iters * 4.0 * 4 * 4096);
  }
  return 0;
}
