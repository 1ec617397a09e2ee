// What does a row gather cost on the CU's texture-address / L1 path, by FORM of the instruction?
//
// k_spconv_fwd3 gathers the A operand in MFMA fragment shape: one buffer_load_dwordx4 = 16 rows x 64 bytes (4 lanes per
// row), whatever the row length -- a 256-byte row (64 channels) is fetched by four instructions that each touch a quarter
// of 16 rows.  This microbenchmark prices the alternatives the LDS-staged kernel (k_spconv_fwd4) can use because the LDS
// image need not be fragment-shaped: R rows x (1024 / R) contiguous bytes per instruction, to VGPRs or straight to LDS
// (buffer_load ... lds), and 4-byte-per-lane LDS loads (4 rows x 64 bytes).  It also checks what an out-of-range lane of
// an LDS load writes (zeros are what the convolution needs for a missing neighbour).
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/gather_forms.hip -o /tmp/gf && /tmp/gf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// FORM 0: dwordx4 -> VGPR; 1: dwordx4 -> LDS; 2: dword -> LDS (RPI rows of 256 / RPI bytes per instruction)
// RPI = rows per instruction; every row is a whole row of the table (row bytes = bytes per instruction / RPI)
template <int FORM, int RPI, int U>
__global__ __launch_bounds__(256) void k_gather(const float* src, unsigned bytes, unsigned nrows, int iters, float* out) {
  __shared__ f32x4 lds[4][U][64];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr unsigned LB = FORM == 2 ? 4u : 16u;       // bytes per lane
  constexpr unsigned RB = 64u * LB / RPI;             // row bytes
  constexpr unsigned LPR = 64 / RPI;                  // lanes per row
  unsigned rsel = (unsigned)lane / LPR, inrow = ((unsigned)lane % LPR) * LB;
  // FORM 3 / 4 (round 6): the SAME 8 rows x 128 bytes per instruction, but with the lanes of a row spread over the wave the way a
  // register-level transpose into MFMA fragments wants them: row = lane & 7, 16-byte chunk = 4 (lane >> 3 & 1) + (lane >> 4)
  // (FORM 3) or lane >> 3 (FORM 4).  Does the texture path still see 8 lines, or 64 pieces?
  if constexpr (FORM == 3) { rsel = (unsigned)lane & 7u; inrow = ((((unsigned)lane >> 3) & 1u) * 4u + ((unsigned)lane >> 4)) * 16u; }
  if constexpr (FORM == 4) { rsel = (unsigned)lane & 7u; inrow = ((unsigned)lane >> 3) * 16u; }
  // FORM 5: 8 adjacent lanes = one 128-byte row, the 16-byte chunks XOR-swizzled inside the row (chunk = lane & 7 ^ s, s from the
  // row and the instruction number: the bank-conflict-free LDS image of k_spconv_x3f).  FORM 6: the pattern k_spconv_x3 uses today
  // (row = lane & 15, chunk = lane >> 4 of a 64-byte half row).  FORM 7: FORM 5 straight to LDS.
  if constexpr (FORM == 5 || FORM == 7) { rsel = (unsigned)lane >> 3; inrow = 0u; }
  if constexpr (FORM == 6) { rsel = (unsigned)lane & 15u; inrow = ((unsigned)lane >> 4) * 16u; }
  const unsigned wid = (blockIdx.x * 4 + wave) * 977u;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    unsigned off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      off[u] = ((((wid + (unsigned)(it * U + u) * 64u + rsel) * 2654435761u) >> 9) & (nrows - 1u)) * RB + inrow;  // nrows = 2^n
      if constexpr (FORM == 5 || FORM == 7) off[u] += ((((unsigned)lane & 7u) ^ ((unsigned)(u & 3) + 4u * (((unsigned)lane >> 4) & 1u))) * 16u);
    }
    if constexpr (FORM == 0 || (FORM >= 3 && FORM != 7)) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off[u], 0, 0));
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (FORM == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(&lds[wave][u][0]), 4, (int)off[u], 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(&lds[wave][u][0]), 16, (int)off[u], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if constexpr (FORM == 1 || FORM == 2 || FORM == 7) acc = lds[wave][0][lane];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

// one wave: lanes >= n_in carry an out-of-range offset; what lands in LDS?
__global__ void k_oob(const float* src, unsigned bytes, int n_in, float* out) {
  __shared__ f32x4 lds[2][64];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
  const int lane = threadIdx.x;
  lds[0][lane] = (f32x4){-7.f, -7.f, -7.f, -7.f};
  lds[1][lane] = (f32x4){-7.f, -7.f, -7.f, -7.f};
  __syncthreads();
  const unsigned off = lane < n_in ? (unsigned)(63 - lane) * 16u : 0xFFFFFFFFu;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(&lds[0][0]), 16, (int)off, 0, 0, 0);
  const unsigned off4 = lane < n_in ? (unsigned)(63 - lane) * 4u : 0xFFFFFFFFu;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(&lds[1][0]), 4, (int)off4, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < 4; ++t) out[lane * 4 + t] = lds[0][lane][t];
  out[256 + lane] = ((float*)&lds[1][0])[lane];
}

template <int FORM, int RPI, int U>
static void run(const char* name, const float* src, size_t bytes, float* out, hipEvent_t e0, hipEvent_t e1) {
  const int blocks = 256 * 8, iters = 400;
  constexpr unsigned LB = FORM == 2 ? 4u : 16u;
  constexpr unsigned RB = 64u * LB / RPI;
  for (size_t region : {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30}) {
    const unsigned nrows = (unsigned)(region / RB);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL((k_gather<FORM, RPI, U>), dim3(blocks), dim3(256), 0, 0, src, (unsigned)bytes, nrows, iters, out);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr = (double)blocks * 4 * iters * U;
    printf("%-44s U=%d region %5zu MB: %7.3f ms  %6.1f cycles per wave instruction per CU  %6.2f TB/s\n", name, U, region >> 20, ms,
           ms * 1e-3 * 2.4e9 / (instr / 256), instr * 64.0 * LB / (ms * 1e-3) / 1e12);
  }
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  float* src; float* out;
  (void)hipMalloc(&src, bytes);
  (void)hipMalloc(&out, 4096);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  (void)hipMemset(src, 0, bytes);
  (void)hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  // --- semantics of out-of-range lanes
  for (int n_in : {64, 40}) {
    hipLaunchKernelGGL(k_oob, dim3(1), dim3(64), 0, 0, src, 1024u, n_in, out);
    std::vector<float> o(320);
    (void)hipMemcpy(o.data(), out, 320 * 4, hipMemcpyDeviceToHost);
    printf("oob test n_in=%d: x4 lane0 %.0f %.0f %.0f %.0f  lane39 %.0f  lane40 %.0f %.0f  lane63 %.0f | x1 lane0 %.0f lane39 %.0f lane40 %.0f lane63 %.0f\n",
           n_in, o[0], o[1], o[2], o[3], o[39 * 4], o[40 * 4], o[40 * 4 + 1], o[63 * 4], o[256], o[256 + 39], o[256 + 40], o[256 + 63]);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const unsigned tb = (unsigned)(bytes - 1);  // buffer range < 4 GiB
  (void)tb;
  run<0, 16, 4>("x4 -> VGPR, 16 rows x  64 B (fragment)", src, bytes - 4096, out, e0, e1);
  run<0, 8, 4>("x4 -> VGPR,  8 rows x 128 B", src, bytes - 4096, out, e0, e1);
  run<3, 8, 4>("x4 -> VGPR,  8 rows x 128 B, lanes q|g2|row", src, bytes - 4096, out, e0, e1);
  run<4, 8, 4>("x4 -> VGPR,  8 rows x 128 B, lanes chunk|row", src, bytes - 4096, out, e0, e1);
  run<5, 8, 4>("x4 -> VGPR,  8 rows x 128 B, chunks xor-swizzled", src, bytes - 4096, out, e0, e1);
  run<7, 8, 4>("x4 -> LDS,   8 rows x 128 B, chunks xor-swizzled", src, bytes - 4096, out, e0, e1);
  run<6, 16, 4>("x4 -> VGPR, 16 rows x 64 B, lanes chunk|row (x3 today)", src, bytes - 4096, out, e0, e1);
  run<0, 4, 4>("x4 -> VGPR,  4 rows x 256 B", src, bytes - 4096, out, e0, e1);
  run<0, 2, 4>("x4 -> VGPR,  2 rows x 512 B", src, bytes - 4096, out, e0, e1);
  run<1, 16, 4>("x4 -> LDS,  16 rows x  64 B", src, bytes - 4096, out, e0, e1);
  run<1, 8, 4>("x4 -> LDS,   8 rows x 128 B", src, bytes - 4096, out, e0, e1);
  run<1, 4, 4>("x4 -> LDS,   4 rows x 256 B", src, bytes - 4096, out, e0, e1);
  run<1, 2, 4>("x4 -> LDS,   2 rows x 512 B", src, bytes - 4096, out, e0, e1);
  run<2, 4, 4>("x1 -> LDS,   4 rows x  64 B", src, bytes - 4096, out, e0, e1);
  run<2, 2, 4>("x1 -> LDS,   2 rows x 128 B", src, bytes - 4096, out, e0, e1);
  run<0, 16, 8>("x4 -> VGPR, 16 rows x  64 B (fragment)", src, bytes - 4096, out, e0, e1);
  run<1, 16, 8>("x4 -> LDS,  16 rows x  64 B", src, bytes - 4096, out, e0, e1);
  run<1, 4, 8>("x4 -> LDS,   4 rows x 256 B", src, bytes - 4096, out, e0, e1);
  return 0;
}
