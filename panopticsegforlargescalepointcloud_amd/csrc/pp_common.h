// Shared helpers of libpanoptic_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/panoptic_hip.h"

#define PP_WAVE 64
#define PP_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

void pp_set_error(const char* fmt, ...);

#define PP_HIP(call)                                                                  \
  do {                                                                                \
    hipError_t e__ = (call);                                                          \
    if (e__ != hipSuccess) {                                                          \
      pp_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return PP_ERR_HIP;                                                              \
    }                                                                                 \
  } while (0)

#define PP_LAUNCH_CHECK() PP_HIP(hipGetLastError())

#define PP_REQUIRE(cond, msg)                       \
  do {                                              \
    if (!(cond)) {                                  \
      pp_set_error("%s:%d %s", __FILE__, __LINE__, msg); \
      return PP_ERR_INVALID;                        \
    }                                               \
  } while (0)

static inline hipStream_t pp_s(pp_stream_t s) { return (hipStream_t)s; }
static inline unsigned pp_blocks(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }
static inline size_t pp_align(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- 64-bit coordinate key: batch 16 bits, x/y/z 16 bits each (offset 32768) ----
__host__ __device__ inline bool pp_key_ok(int b, int x, int y, int z) {
  return ((unsigned)b < 65536u) && ((unsigned)(x + 32768) < 65536u) && ((unsigned)(y + 32768) < 65536u) &&
         ((unsigned)(z + 32768) < 65536u);
}
__host__ __device__ inline uint64_t pp_key_pack(int b, int x, int y, int z) {
  return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)(x + 32768) << 32) |
         ((uint64_t)(uint16_t)(y + 32768) << 16) | (uint64_t)(uint16_t)(z + 32768);
}
__host__ __device__ inline uint64_t pp_mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}
__device__ inline int pp_floor_div(int a, int b) {
  int q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}

// Home slot of a coordinate key: the 4x4x4 voxel block (batch, x>>2, y>>2, z>>2) is hashed, the low two bits of
// x/y/z pick one of 64 consecutive slots.  The 27 probes of a kernel-map row then touch a handful of 512 B key
// groups that the Morton-ordered neighbouring rows have just pulled into L2, instead of 27 random cache lines.
__host__ __device__ inline uint64_t pp_home_slot(uint64_t key, uint64_t mask) {
  const uint64_t block = key & 0xFFFFFFFCFFFCFFFCull;
  const uint64_t low = ((key >> 32) & 3ull) << 4 | ((key >> 16) & 3ull) << 2 | (key & 3ull);
  return ((pp_mix64(block) << 6) | low) & mask;
}

// open-addressing lookup: returns slot holding `key`, or -1
__device__ inline int64_t pp_hash_find_slot(const uint64_t* __restrict__ keys, int64_t cap, uint64_t key) {
  uint64_t mask = (uint64_t)cap - 1;
  uint64_t s = pp_home_slot(key, mask);
  for (;;) {
    uint64_t k = keys[s];
    if (k == key) return (int64_t)s;
    if (k == PP_EMPTY_KEY) return -1;
    s = (s + 1) & mask;
  }
}
// insert-or-find: returns slot
__device__ inline int64_t pp_hash_insert_slot(uint64_t* keys, int64_t cap, uint64_t key) {
  uint64_t mask = (uint64_t)cap - 1;
  uint64_t s = pp_home_slot(key, mask);
  for (;;) {
    unsigned long long prev = atomicCAS((unsigned long long*)&keys[s], (unsigned long long)PP_EMPTY_KEY,
                                       (unsigned long long)key);
    if (prev == PP_EMPTY_KEY || prev == key) return (int64_t)s;
    s = (s + 1) & mask;
  }
}

// ---- internal row-order key (pp_morton_order) ------------------------------------------------------------------------
// key = batch << 48 | X(qx) | Y(qy) | Z(qz) with q = (coordinate + 32768) / unit: every axis contributes its own bits, so
// the keys of the 27 neighbours of a voxel are ORs of 9 per-axis terms.  block_bits <= 1: Z-order.  block_bits = B >= 2:
// [Z-order of q >> B][parity of q][Z-order of (q >> 1) & (2^(B-1) - 1)].  In both layouts the low 12 bits enumerate the
// positions inside a group of <= 4096 voxels (a 16^3 block for B = 0 and B = 4) -- the block index relies on that.
__host__ __device__ inline uint64_t pp_spread3_64(uint64_t x) {
  x &= 0xFFFFull;
  x = (x | (x << 32)) & 0x1F00000000FFFFull;
  x = (x | (x << 16)) & 0x1F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}
__host__ __device__ inline uint64_t pp_order_axis(uint32_t q, int axis, int block_bits) {
  if (block_bits <= 1) return pp_spread3_64(q) << axis;
  const int hb = block_bits - 1;
  const uint64_t inner = pp_spread3_64((q >> 1) & ((1u << hb) - 1u));
  const uint64_t par = (uint64_t)(q & 1u);
  const uint64_t outer = pp_spread3_64(q >> block_bits);
  return ((outer << (3 * block_bits)) | (par << (3 * hb)) | inner) << axis;
}
// inverse of pp_order_axis: one axis' q from a key
__host__ __device__ inline uint32_t pp_compact3_64(uint64_t x) {
  x &= 0x1249249249249249ull;
  x = (x | (x >> 2)) & 0x10C30C30C30C30C3ull;
  x = (x | (x >> 4)) & 0x100F00F00F00F00Full;
  x = (x | (x >> 8)) & 0x1F0000FF0000FFull;
  x = (x | (x >> 16)) & 0x1F00000000FFFFull;
  x = (x | (x >> 32)) & 0xFFFFull;
  return (uint32_t)x;
}
__host__ __device__ inline uint32_t pp_order_axis_inv(uint64_t key, int axis, int block_bits) {
  key >>= axis;
  if (block_bits <= 1) return pp_compact3_64(key);
  const int hb = block_bits - 1;
  const uint32_t inner = pp_compact3_64(key & ((1ull << (3 * hb)) - 1ull));
  const uint32_t par = (uint32_t)((key >> (3 * hb)) & 1ull);
  const uint32_t outer = pp_compact3_64(key >> (3 * block_bits));
  return (outer << block_bits) | (inner << 1) | par;
}
__host__ __device__ inline uint64_t pp_order_key(int b, int x, int y, int z, int unit_shift, int block_bits) {
  return ((uint64_t)(uint16_t)b << 48) | pp_order_axis((uint32_t)(x + 32768) >> unit_shift, 0, block_bits) |
         pp_order_axis((uint32_t)(y + 32768) >> unit_shift, 1, block_bits) |
         pp_order_axis((uint32_t)(z + 32768) >> unit_shift, 2, block_bits);
}

// internal device primitives (pp_scan.hip)
size_t pp_scan_workspace(int64_t n);
// exclusive prefix sum of int32; total (device int32[1], may be NULL) receives the grand total
int pp_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total, void* ws, size_t ws_bytes,
                          hipStream_t stream);
size_t pp_sort_pairs_workspace(int64_t n);
// stable LSD radix sort of (key,value) pairs on bits [0,end_bit); results land in keys_out/vals_out
int pp_sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                      int64_t n, int end_bit, void* ws, size_t ws_bytes, hipStream_t stream);
int pp_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const int32_t* vals_in, int32_t* vals_out,
                      int64_t n, int end_bit, void* ws, size_t ws_bytes, hipStream_t stream);

// bump allocator over a caller-provided workspace
struct PPArena {
  char* base;
  size_t size;
  size_t used;
  PPArena(void* p, size_t n) : base((char*)p), size(n), used(0) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = pp_align(count * sizeof(T));
    if (used + bytes > size) return nullptr;
    T* r = (T*)(base + used);
    used += bytes;
    return r;
  }
  size_t left() const { return size - used; }
  void* cur() const { return base + used; }
};
