// K1b/K3c: block index of a coordinate level and kernel maps looked up through it.
//
// The rows of a level are kept sorted by the order key of pp_morton_order (pp_common.h: pp_order_key).  The key's low
// 12 bits enumerate the positions inside a group of at most 4096 voxels (a 16^3 block), its upper bits name the group.
// The index stores, per occupied group ("block"),
//     start   first row of the block                                   int32
//     rec     64 x { 64 bits of the 4096-bit occupancy map in key order, row of the first voxel of that word }   (16 B each)
// plus a small open-addressing hash  block key -> block number.  The row of a voxel is then
//     rec[b][w].row + popcount(rec[b][w].bits & lower bits)             (w = word of its 12-bit code)
// i.e. ONE 16-byte load from a 1-KiB record shared by the voxels of the block, instead of a 12-byte probe of a
// row-level hash table spread over ~10 KiB per block: the 27 lookups of a kernel-map row and of its neighbours in the
// wave hit the same few records in L1/L2.  The block hash is probed only when a neighbour leaves the previous block.
// Replaces the per-row hash probing of pp_kernel_map for the coordinate manager (same map definition, same results;
// reference: MinkowskiEngine's coordinate map / kernel map, call sites api_modules.py:30-51,256-271).
#include "pp_common.h"

#define BI_WORDS 64

struct BlockIndex {
  const uint64_t* bkeys;
  const int32_t* bvals;
  int64_t cap;
  const ulonglong2* rec;  // [n_blocks * 64]: x = occupancy word, y = row of the word's first voxel
};

__device__ inline int bi_find_block(const BlockIndex& I, uint64_t blk) {
  const uint64_t mask = (uint64_t)I.cap - 1;
  uint64_t s = pp_mix64(blk) & mask;
  for (;;) {
    const uint64_t k = I.bkeys[s];
    if (k == blk) return I.bvals[s];
    if (k == PP_EMPTY_KEY) return -1;
    s = (s + 1) & mask;
  }
}

// ---- build -------------------------------------------------------------------------------------------------------------
// phase 1: block number of every row (rows sorted by key), number of blocks, duplicated rows
__global__ __launch_bounds__(256) void k_bi_flags(const int4* __restrict__ coords, int64_t n, int unit_shift, int block_bits,
                                                  int32_t* flag, int32_t* counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  const uint64_t key = pp_order_key(c.x, c.y, c.z, c.w, unit_shift, block_bits);
  int f = 1;
  if (i > 0) {
    const int4 p = coords[i - 1];
    const uint64_t pk = pp_order_key(p.x, p.y, p.z, p.w, unit_shift, block_bits);
    f = (pk >> 12) != (key >> 12);
    if (pk == key) atomicAdd(&counts[1], 1);        // duplicate coordinate
    if (pk > key) atomicAdd(&counts[2], 1);         // not sorted
  }
  if (!pp_key_ok(c.x, c.y, c.z, c.w)) atomicAdd(&counts[3], 1);
  flag[i] = f;
}
__global__ __launch_bounds__(256) void k_bi_rowblock(const int32_t* __restrict__ flag, const int32_t* __restrict__ excl,
                                                     int64_t n, int32_t* row_block) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) row_block[i] = excl[i] + flag[i] - 1;  // inclusive scan - 1
}
// phase 2
__global__ __launch_bounds__(256) void k_bi_hash_fill(uint64_t* keys, int64_t cap) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) keys[i] = PP_EMPTY_KEY;
}
__global__ __launch_bounds__(256) void k_bi_scatter(const int4* __restrict__ coords, int64_t n, int unit_shift,
                                                    int block_bits, const int32_t* __restrict__ row_block,
                                                    uint64_t* bkeys, int32_t* bvals, int64_t cap, int32_t* start,
                                                    unsigned long long* rec, uint64_t* bkey_ord) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  const uint64_t key = pp_order_key(c.x, c.y, c.z, c.w, unit_shift, block_bits);
  const int b = row_block[i];
  const int code = (int)(key & 4095ull);
  atomicOr(&rec[2 * ((size_t)b * BI_WORDS + (code >> 6))], 1ull << (code & 63));
  if (i == 0 || row_block[i - 1] != b) {  // first row of the block: owns start[] and the hash entry
    start[b] = (int32_t)i;
    const uint64_t blk = key >> 12;
    if (bkey_ord) bkey_ord[b] = blk;
    const uint64_t mask = (uint64_t)cap - 1;
    uint64_t s = pp_mix64(blk) & mask;
    for (;;) {
      const unsigned long long prev = atomicCAS((unsigned long long*)&bkeys[s], (unsigned long long)PP_EMPTY_KEY,
                                                (unsigned long long)blk);
      if (prev == PP_EMPTY_KEY) {
        bvals[s] = b;
        break;
      }
      s = (s + 1) & mask;
    }
  }
}
// one wave per block: first row of every word = start of the block + exclusive prefix of the 64 word popcounts
__global__ __launch_bounds__(256) void k_bi_prefix(unsigned long long* rec, const int32_t* __restrict__ start, int64_t nb) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nb) return;
  const int c = __popcll(rec[2 * ((size_t)b * BI_WORDS + lane)]);
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  rec[2 * ((size_t)b * BI_WORDS + lane) + 1] = (unsigned long long)(unsigned)(start[b] + incl - c);
}

extern "C" size_t pp_block_index_workspace(int64_t n) {
  const size_t m = (size_t)std::max<int64_t>(n, 1);
  return 2 * pp_align(m * 4) + pp_scan_workspace(n) + 1024;
}
extern "C" int64_t pp_block_index_capacity(int64_t n_blocks) {
  int64_t cap = 256;
  while (cap < 2 * n_blocks) cap <<= 1;
  return cap;
}
extern "C" int pp_block_index_count(const int32_t* coords_sorted, int64_t n, int32_t unit, int32_t block_bits,
                                    int32_t* row_block, int32_t* counts, void* workspace, size_t workspace_bytes,
                                    pp_stream_t stream) {
  PP_REQUIRE(row_block && counts, "pp_block_index_count: null output");
  PP_REQUIRE(unit >= 1 && (unit & (unit - 1)) == 0 && unit <= 16384, "pp_block_index_count: unit must be a power of two");
  PP_REQUIRE(block_bits >= 0 && block_bits <= 8, "pp_block_index_count: block_bits in [0,8]");
  if (workspace_bytes < pp_block_index_workspace(n)) return PP_ERR_WORKSPACE;
  hipStream_t s = pp_s(stream);
  PP_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(int32_t), s));
  if (n == 0) return PP_OK;
  int unit_shift = 0;
  while ((1 << unit_shift) < unit) ++unit_shift;
  PPArena ar(workspace, workspace_bytes);
  int32_t* flag = ar.take<int32_t>((size_t)n);
  int32_t* excl = ar.take<int32_t>((size_t)n);
  const unsigned nb = pp_blocks(n, 256);
  hipLaunchKernelGGL(k_bi_flags, dim3(nb), dim3(256), 0, s, (const int4*)coords_sorted, n, unit_shift, block_bits, flag,
                     counts);
  PP_LAUNCH_CHECK();
  int rc = pp_exclusive_scan_i32(flag, excl, n, counts, ar.cur(), ar.left(), s);  // counts[0] = number of blocks
  if (rc) return rc;
  hipLaunchKernelGGL(k_bi_rowblock, dim3(nb), dim3(256), 0, s, flag, excl, n, row_block);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_block_index_fill(const int32_t* coords_sorted, int64_t n, int32_t unit, int32_t block_bits,
                                   const int32_t* row_block, int64_t n_blocks, uint64_t* bkeys, int32_t* bvals,
                                   int64_t cap, int32_t* start, uint64_t* rec, uint64_t* bkey_ord, pp_stream_t stream) {
  PP_REQUIRE(bkeys && bvals && start && rec, "pp_block_index_fill: null output");
  PP_REQUIRE(cap >= 2 * n_blocks && (cap & (cap - 1)) == 0, "pp_block_index_fill: cap must be a power of two >= 2 n_blocks");
  hipStream_t s = pp_s(stream);
  int unit_shift = 0;
  while ((1 << unit_shift) < unit) ++unit_shift;
  hipLaunchKernelGGL(k_bi_hash_fill, dim3((unsigned)std::min<int64_t>((cap + 255) / 256, 4096)), dim3(256), 0, s, bkeys, cap);
  PP_LAUNCH_CHECK();
  if (n_blocks > 0) PP_HIP(hipMemsetAsync(rec, 0, 2 * sizeof(uint64_t) * BI_WORDS * (size_t)n_blocks, s));
  if (n == 0) return PP_OK;
  hipLaunchKernelGGL(k_bi_scatter, dim3(pp_blocks(n, 256)), dim3(256), 0, s, (const int4*)coords_sorted, n, unit_shift,
                     block_bits, row_block, bkeys, bvals, cap, start, (unsigned long long*)rec, bkey_ord);
  PP_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bi_prefix, dim3(pp_blocks(n_blocks, 4)), dim3(256), 0, s, (unsigned long long*)rec, start, n_blocks);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- next coarser level straight from the index (no row hash, no sort) -----------------------------------------------
// With the parity-block key (block_bits = 4) the coarse level (tensor stride doubled: Q = q >> 1) is a pure function of
// the fine level's bitmaps: the 8 fine blocks that share (q >> 5) form one coarse block; inside a fine block the coarse
// voxel of a position depends only on its half coordinates h = (q >> 1) & 7 -- the eight parity copies collapse (OR) --
// and lands at the coarse code  [h & 1 per axis][Z-order of (octant << 2 | h >> 1)].  Fine blocks are sorted, so the
// children of a coarse block are consecutive.  Everything below works on blocks (~300 voxels each), not on rows.
__device__ inline uint32_t bi_compact3_9(uint32_t z) {  // bits 0,3,6 -> 0,1,2
  return (z & 1u) | ((z >> 2) & 2u) | ((z >> 4) & 4u);
}
__device__ inline uint32_t bi_spread3_3(uint32_t v) {  // bits 0,1,2 -> 0,3,6
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4);
}
__device__ inline uint64_t bi_parent_key(uint64_t fkey) {  // key >> 12 of the fine block -> key >> 12 of its coarse block
  const uint64_t outer_mask = (1ull << 36) - 1ull;
  return (fkey & ~outer_mask) | ((fkey & outer_mask) >> 3);
}
// nb_dev (nullable): the number of fine blocks lives in device memory (a chained coarsening, pp_block_index_coarsen_chain: the
// previous level's count has not been read by the host); nb is then the launch's upper bound and the flags behind the real count
// are zero, so that the scans over the upper bound see nothing there
__global__ __launch_bounds__(256) void k_bic_flags(const uint64_t* __restrict__ fkey, int64_t nb, const int32_t* __restrict__ nb_dev,
                                                   int32_t* flag) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nbr = nb_dev ? (int64_t)nb_dev[0] : nb;
  if (b < nb) flag[b] = (b < nbr && (b == 0 || bi_parent_key(fkey[b]) != bi_parent_key(fkey[b - 1]))) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_bic_first_child(const int32_t* __restrict__ flag, const int32_t* __restrict__ rank,
                                                         int64_t nb, int32_t* first_child) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb && flag[b]) first_child[rank[b]] = (int32_t)b;
}
// one wave per coarse block: its bitmap from the children's bitmaps, its voxel count and key
__global__ __launch_bounds__(256) void k_bic_bits(const uint64_t* __restrict__ fkey, const uint64_t* __restrict__ frec,
                                                  int64_t nb_f, const int32_t* __restrict__ nb_dev,
                                                  const int32_t* __restrict__ first_child,
                                                  const int32_t* __restrict__ n_coarse, uint64_t* crec, int32_t* ccount,
                                                  uint64_t* ckey) {
  __shared__ unsigned long long lds[4][BI_WORDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbc = n_coarse[0];
  if (nb_dev) nb_f = nb_dev[0];
  const int64_t c = (int64_t)blockIdx.x * 4 + wave;
  if (c >= nbc) return;
  unsigned long long* w = lds[wave];
  w[lane] = 0ull;
  const int b0 = first_child[c], b1 = c + 1 < nbc ? first_child[c + 1] : (int)nb_f;
  for (int b = b0; b < b1; ++b) {
    const uint32_t oct = (uint32_t)(fkey[b] & 7ull);  // low bits of the fine block's Z-order position: x, y, z
    // lanes 0..7: word j of the octant's 512-bit occupancy = OR over the 8 parity copies
    if (lane < 8) {
      unsigned long long o = 0ull;
#pragma unroll
      for (int par = 0; par < 8; ++par) o |= frec[2 * ((size_t)b * BI_WORDS + par * 8 + lane)];
      while (o) {
        const int bit = __builtin_ctzll(o);
        o &= o - 1ull;
        const uint32_t z = (uint32_t)lane * 64u + (uint32_t)bit;  // Z-order of the half coordinates h (9 bits)
        const uint32_t hx = bi_compact3_9(z), hy = bi_compact3_9(z >> 1), hz = bi_compact3_9(z >> 2);
        const uint32_t par2 = (hx & 1u) | ((hy & 1u) << 1) | ((hz & 1u) << 2);
        const uint32_t vx = ((oct & 1u) << 2) | (hx >> 1), vy = (((oct >> 1) & 1u) << 2) | (hy >> 1),
                       vz = (((oct >> 2) & 1u) << 2) | (hz >> 1);
        const uint32_t code = (par2 << 9) | bi_spread3_3(vx) | (bi_spread3_3(vy) << 1) | (bi_spread3_3(vz) << 2);
        atomicOr(&w[code >> 6], 1ull << (code & 63u));
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned long long mine = w[lane];
  crec[2 * ((size_t)c * BI_WORDS + lane)] = mine;
  int cnt = __popcll(mine);
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (lane == 0) {
    ccount[c] = cnt;
    ckey[c] = bi_parent_key(fkey[b0]);
  }
}
// one wave per coarse block: prefix counts, hash entry, coordinate rows (decoded from block key + code)
__global__ __launch_bounds__(256) void k_bic_finish(const uint64_t* __restrict__ ckey, uint64_t* crec,
                                                    const int32_t* __restrict__ cstart, const int32_t* __restrict__ n_coarse,
                                                    int unit_shift, uint64_t* bkeys, int32_t* bvals, int64_t cap, int4* coords) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_coarse[0]) return;
  unsigned long long word = crec[2 * ((size_t)c * BI_WORDS + lane)];
  const int cnt = __popcll(word);
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  const int before = incl - cnt;
  crec[2 * ((size_t)c * BI_WORDS + lane) + 1] = (uint64_t)(unsigned)(cstart[c] + before);
  const uint64_t blk = ckey[c];
  if (lane == 0) {
    const uint64_t mask = (uint64_t)cap - 1;
    uint64_t s = pp_mix64(blk) & mask;
    for (;;) {
      const unsigned long long prev = atomicCAS((unsigned long long*)&bkeys[s], (unsigned long long)PP_EMPTY_KEY,
                                                (unsigned long long)blk);
      if (prev == PP_EMPTY_KEY) {
        bvals[s] = (int32_t)c;
        break;
      }
      s = (s + 1) & mask;
    }
  }
  int row = cstart[c] + before;
  while (word) {
    const int bit = __builtin_ctzll(word);
    word &= word - 1ull;
    const uint64_t key = (blk << 12) | (uint64_t)(lane * 64 + bit);
    const uint64_t body = key & 0xFFFFFFFFFFFFull;
    coords[row++] = make_int4((int)(key >> 48), (int)(pp_order_axis_inv(body, 0, 4) << unit_shift) - 32768,
                              (int)(pp_order_axis_inv(body, 1, 4) << unit_shift) - 32768,
                              (int)(pp_order_axis_inv(body, 2, 4) << unit_shift) - 32768);
  }
}

extern "C" size_t pp_block_index_coarsen_workspace(int64_t nb_fine) {
  const size_t m = (size_t)std::max<int64_t>(nb_fine, 1);
  return 4 * pp_align(m * 4) + pp_scan_workspace(nb_fine) + 1024;
}
// All outputs have capacity nb_fine blocks (cap = pp_block_index_capacity(nb_fine)) resp. n_fine rows; counts = {coarse
// blocks, coarse rows}.  unit_coarse = tensor stride of the NEW level.  block_bits must be 4.
static int bi_coarsen_launch(const uint64_t* f_bkey_ord, const uint64_t* f_rec, int64_t nb_fine, const int32_t* nb_dev,
                             int32_t unit_coarse, uint64_t* bkeys, int32_t* bvals, int64_t cap, int32_t* start, uint64_t* rec,
                             uint64_t* bkey_ord, int32_t* coords, int32_t* counts, void* workspace, size_t workspace_bytes,
                             hipStream_t s) {
  PP_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_bi_hash_fill, dim3((unsigned)std::min<int64_t>((cap + 255) / 256, 4096)), dim3(256), 0, s, bkeys, cap);
  PP_LAUNCH_CHECK();
  if (nb_fine == 0) return PP_OK;
  int unit_shift = 0;
  while ((1 << unit_shift) < unit_coarse) ++unit_shift;
  PPArena ar(workspace, workspace_bytes);
  int32_t* flag = ar.take<int32_t>((size_t)nb_fine);
  int32_t* rank = ar.take<int32_t>((size_t)nb_fine);
  int32_t* first_child = ar.take<int32_t>((size_t)nb_fine);
  int32_t* ccount = ar.take<int32_t>((size_t)nb_fine);
  const unsigned gb = pp_blocks(nb_fine, 256), gw = pp_blocks(nb_fine, 4);
  hipLaunchKernelGGL(k_bic_flags, dim3(gb), dim3(256), 0, s, f_bkey_ord, nb_fine, nb_dev, flag);
  PP_LAUNCH_CHECK();
  int rc = pp_exclusive_scan_i32(flag, rank, nb_fine, counts, ar.cur(), ar.left(), s);  // counts[0] = coarse blocks
  if (rc) return rc;
  hipLaunchKernelGGL(k_bic_first_child, dim3(gb), dim3(256), 0, s, flag, rank, nb_fine, first_child);
  PP_HIP(hipMemsetAsync(ccount, 0, sizeof(int32_t) * (size_t)nb_fine, s));
  hipLaunchKernelGGL(k_bic_bits, dim3(gw), dim3(256), 0, s, f_bkey_ord, f_rec, nb_fine, nb_dev, first_child, counts, rec, ccount,
                     bkey_ord);
  PP_LAUNCH_CHECK();
  rc = pp_exclusive_scan_i32(ccount, start, nb_fine, counts + 1, ar.cur(), ar.left(), s);  // counts[1] = coarse rows
  if (rc) return rc;
  hipLaunchKernelGGL(k_bic_finish, dim3(gw), dim3(256), 0, s, bkey_ord, rec, start, counts, unit_shift, bkeys, bvals, cap,
                     (int4*)coords);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_block_index_coarsen(const uint64_t* f_bkey_ord, const uint64_t* f_rec, int64_t nb_fine,
                                      int32_t unit_coarse, int32_t block_bits, uint64_t* bkeys, int32_t* bvals,
                                      int64_t cap, int32_t* start, uint64_t* rec, uint64_t* bkey_ord, int32_t* coords,
                                      int32_t* counts, void* workspace, size_t workspace_bytes, pp_stream_t stream) {
  PP_REQUIRE(f_bkey_ord && f_rec && bkeys && bvals && start && rec && bkey_ord && coords && counts,
             "pp_block_index_coarsen: null pointer");
  PP_REQUIRE(block_bits == 4, "pp_block_index_coarsen: needs the parity-block row order (block_bits = 4)");
  PP_REQUIRE(unit_coarse >= 2 && (unit_coarse & (unit_coarse - 1)) == 0, "pp_block_index_coarsen: unit must be a power of two >= 2");
  PP_REQUIRE(cap >= 2 * nb_fine && (cap & (cap - 1)) == 0, "pp_block_index_coarsen: cap must be a power of two >= 2 nb_fine");
  if (workspace_bytes < pp_block_index_coarsen_workspace(nb_fine)) return PP_ERR_WORKSPACE;
  return bi_coarsen_launch(f_bkey_ord, f_rec, nb_fine, nullptr, unit_coarse, bkeys, bvals, cap, start, rec, bkey_ord, coords,
                           counts, workspace, workspace_bytes, pp_s(stream));
}

// The level chain of a U-Net encoder in ONE call: `levels` successive coarsenings (tensor stride unit_fine -> 2 unit_fine -> ...),
// level l built from level l - 1 with the block and row counts chained in DEVICE memory -- no host read between the levels.  The
// host knows only upper bounds (a coarse level has at most as many blocks and rows as the input level): every level's outputs have
// the capacity of the input level (nb_fine blocks, n_fine rows), the launches cover nb_fine blocks and read the real count from
// counts[l - 1].  Per-level outputs are laid out one after the other: bkeys / bvals [levels][cap], start / bkey_ord [levels][nb_fine],
// rec [levels][nb_fine * 128], coords [levels][n_fine * 4], counts [levels][2] = {blocks, rows} -- ONE read of `counts` after the call
// sizes every level.  Level l's arrays equal those of `levels` single pp_block_index_coarsen calls (the hash tables aside: their
// capacity is that of the input level).  workspace: pp_block_index_coarsen_workspace(nb_fine), reused level after level.
extern "C" int pp_block_index_coarsen_chain(const uint64_t* f_bkey_ord, const uint64_t* f_rec, int64_t nb_fine, int64_t n_fine,
                                            int32_t unit_fine, int32_t block_bits, int32_t levels, uint64_t* bkeys,
                                            int32_t* bvals, int64_t cap, int32_t* start, uint64_t* rec, uint64_t* bkey_ord,
                                            int32_t* coords, int32_t* counts, void* workspace, size_t workspace_bytes,
                                            pp_stream_t stream) {
  PP_REQUIRE(f_bkey_ord && f_rec && bkeys && bvals && start && rec && bkey_ord && coords && counts,
             "pp_block_index_coarsen_chain: null pointer");
  PP_REQUIRE(block_bits == 4, "pp_block_index_coarsen_chain: needs the parity-block row order (block_bits = 4)");
  PP_REQUIRE(levels >= 1 && levels <= 14, "pp_block_index_coarsen_chain: 1 .. 14 levels");
  PP_REQUIRE(unit_fine >= 1 && (unit_fine & (unit_fine - 1)) == 0 && ((int64_t)unit_fine << levels) <= 32768,
             "pp_block_index_coarsen_chain: unit must be a power of two and the coarsest tensor stride <= 32768");
  PP_REQUIRE(cap >= 2 * nb_fine && (cap & (cap - 1)) == 0, "pp_block_index_coarsen_chain: cap must be a power of two >= 2 nb_fine");
  PP_REQUIRE(n_fine >= 0 && nb_fine >= 0, "pp_block_index_coarsen_chain: negative size");
  if (workspace_bytes < pp_block_index_coarsen_workspace(nb_fine)) return PP_ERR_WORKSPACE;
  const size_t nbm = (size_t)std::max<int64_t>(nb_fine, 1), nrm = (size_t)std::max<int64_t>(n_fine, 1);
  const uint64_t* in_key = f_bkey_ord;
  const uint64_t* in_rec = f_rec;
  for (int l = 0; l < levels; ++l) {
    uint64_t* o_key = bkey_ord + (size_t)l * nbm;
    uint64_t* o_rec = rec + (size_t)l * nbm * 2 * BI_WORDS;
    int rc = bi_coarsen_launch(in_key, in_rec, nb_fine, l ? counts + 2 * (l - 1) : nullptr, unit_fine << (l + 1),
                               bkeys + (size_t)l * (size_t)cap, bvals + (size_t)l * (size_t)cap, cap, start + (size_t)l * nbm, o_rec, o_key,
                               coords + (size_t)l * nrm * 4, counts + 2 * l, workspace, workspace_bytes, pp_s(stream));
    if (rc) return rc;
    in_key = o_key;
    in_rec = o_rec;
  }
  return PP_OK;
}

// ---- kernel map through the index ------------------------------------------------------------------------------------
// One thread per output row.  The keys of its 27 neighbours are ORs of 9 per-axis terms; the block found for the
// previous neighbour is kept in registers (a row's neighbours touch <= 8 blocks, usually 1-3).
// CUBE (block_bits <= 4: a block is a 16^3 cube of the level's lattice): along one axis the three neighbour positions
// fall into at most two blocks, so the 27 probes touch at most 2 x 2 x 2 blocks.  Those are looked up first; then the
// 16-byte records of ALL 27 probes are loaded unconditionally (index 0 for absent ones) before any of them is used.
// By PMC the one-probe-at-a-time form spent 88 % of its wave cycles in s_waitcnt on a chain of 4-5 dependent loads per
// probe (2.9 us per probe); here a row pays the lookup chain once and one batch of 27 independent loads (54 + 8 block
// starts while the occupancy words and the prefix counts were separate arrays: the kernel is bound by the number of
// scattered load instructions, 1.22 -> 0.9 ms for the 10 M rows of the bench scene's finest level).
template <bool CUBE>
__global__ __launch_bounds__(256) void k_kernel_map_bi(const int4* __restrict__ out_coords, int64_t n_out, BlockIndex I,
                                                       int unit_shift, int block_bits, int dstep,
                                                       int32_t* __restrict__ nbr, unsigned long long* n_pairs,
                                                       uint32_t* __restrict__ mask_out,
                                                       const int32_t* __restrict__ translate) {
  // grid-stride over the rows: the pair count is ONE atomic per wave at the very end (an atomic per wave and row chunk
  // on the same address serialises in L2: 154 k of them cost most of the 2 ms a 9.8 M-row map used to take)
  int found = 0;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = out_coords[o];
    uint64_t ax[3][3];
    bool ok[3][3];
    const int cc[3] = {c.y, c.z, c.w};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int v = cc[a] + (d - 1) * dstep;
        // inside the key range AND on the source level's lattice (a fine voxel probing a coarser level may sit between)
        ok[a][d] = (unsigned)(v + 32768) < 65536u && ((unsigned)(v + 32768) & ((1u << unit_shift) - 1u)) == 0u;
        ax[a][d] = pp_order_axis((uint32_t)(v + 32768) >> unit_shift, a, block_bits);
      }
    const uint64_t kb = (uint64_t)(uint16_t)c.x << 48;
    const bool bok = (unsigned)c.x < 65536u;
    uint32_t fmask = 0;
    if constexpr (CUBE) {
      uint64_t pa[3], pb[3];  // the (at most two) block parts of the key per axis, defined by the valid positions
      bool alt[3];
      int sel[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        pa[a] = pb[a] = 0;
        alt[a] = false;
        bool have = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const uint64_t part = ax[a][d] >> 12;
          sel[a][d] = 0;
          if (!ok[a][d]) continue;
          if (!have) {
            pa[a] = pb[a] = part;
            have = true;
          } else if (part != pa[a]) {
            pb[a] = part;
            alt[a] = true;
            sel[a][d] = 1;
          }
        }
      }
      int lb[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        lb[t] = -1;
        if (bok && (!(t & 1) || alt[0]) && (!(t & 2) || alt[1]) && (!(t & 4) || alt[2])) {
          const uint64_t blk = (kb >> 12) | ((t & 1) ? pb[0] : pa[0]) | ((t & 2) ? pb[1] : pa[1]) | ((t & 4) ? pb[2] : pa[2]);
          lb[t] = bi_find_block(I, blk);
        }
      }
      ulonglong2 rec[27];  // one 16-byte load per probe: occupancy word + row of its first voxel
      uint32_t meta[27];   // bit 0..5 = bit index, bit 8 = probe valid
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        int zb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) zb[t] = sel[2][dz] ? lb[4 + t] : lb[t];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int yb0 = sel[1][dy] ? zb[2] : zb[0], yb1 = sel[1][dy] ? zb[3] : zb[1];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int k = dx + 3 * dy + 9 * dz;
            const int blk_i = sel[0][dx] ? yb1 : yb0;
            const bool valid = blk_i >= 0 && ok[0][dx] && ok[1][dy] && ok[2][dz];
            const uint32_t code = (uint32_t)((ax[0][dx] | ax[1][dy] | ax[2][dz]) & 4095ull);
            const uint32_t w = valid ? (uint32_t)blk_i * BI_WORDS + (code >> 6) : 0u;  // word 0 always exists
            rec[k] = I.rec[w];
            meta[k] = (code & 63u) | (valid ? 256u : 0u);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const int bit = (int)(meta[k] & 63u);
        int32_t r = -1;
        if ((meta[k] & 256u) && ((rec[k].x >> bit) & 1ull))
          r = (int)(uint32_t)rec[k].y + __popcll(rec[k].x & ((1ull << bit) - 1ull));
        found += r >= 0 ? 1 : 0;
        fmask |= (r >= 0 ? 1u : 0u) << k;
        if (translate && r >= 0) r = translate[r];
        nbr[(int64_t)k * n_out + o] = r;
      }
    } else {
    uint64_t last_blk = ~0ull;
    int last_b = -1;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int dx = k % 3, dy = (k / 3) % 3, dz = k / 9;
      int32_t r = -1;
      if (bok && ok[0][dx] && ok[1][dy] && ok[2][dz]) {
        const uint64_t key = kb | ax[0][dx] | ax[1][dy] | ax[2][dz];
        const uint64_t blk = key >> 12;
        if (blk != last_blk) {
          last_blk = blk;
          last_b = bi_find_block(I, blk);
        }
        if (last_b >= 0) {
          const int code = (int)(key & 4095ull);
          const size_t w = (size_t)last_b * BI_WORDS + (code >> 6);
          const ulonglong2 rc = I.rec[w];
          const int bit = code & 63;
          if ((rc.x >> bit) & 1ull) r = (int)(uint32_t)rc.y + __popcll(rc.x & ((1ull << bit) - 1ull));
        }
      }
      found += r >= 0 ? 1 : 0;
      fmask |= (r >= 0 ? 1u : 0u) << k;
      if (translate && r >= 0) r = translate[r];
      nbr[(int64_t)k * n_out + o] = r;
    }
    }
    if (mask_out) mask_out[o] = fmask;
  }
  if (n_pairs) {
    for (int off = 32; off > 0; off >>= 1) found += __shfl_xor(found, off);
    if ((threadIdx.x & 63) == 0 && found) atomicAdd(n_pairs, (unsigned long long)found);
  }
}

static inline unsigned kmb_grid(int64_t n_out) { return (unsigned)std::min<int64_t>(pp_blocks(n_out, 256), 256 * 16); }

extern "C" int pp_kernel_map_bi(const int32_t* out_coords, int64_t n_out, const uint64_t* bkeys, const int32_t* bvals,
                                int64_t cap, const uint64_t* rec, int32_t unit_src, int32_t block_bits, int32_t step,
                                int32_t sign, int32_t* nbr,
                                int64_t* n_pairs, uint32_t* mask_out, const int32_t* translate, pp_stream_t stream) {
  PP_REQUIRE(out_coords || n_out == 0, "pp_kernel_map_bi: null coordinates");
  PP_REQUIRE(bkeys && bvals && rec && nbr, "pp_kernel_map_bi: null index");
  PP_REQUIRE(sign == 1 || sign == -1, "pp_kernel_map_bi: sign must be +1 or -1");
  PP_REQUIRE(unit_src >= 1 && (unit_src & (unit_src - 1)) == 0, "pp_kernel_map_bi: unit must be a power of two");
  hipStream_t s = pp_s(stream);
  if (n_pairs) PP_HIP(hipMemsetAsync(n_pairs, 0, sizeof(int64_t), s));
  if (n_out == 0) return PP_OK;
  int unit_shift = 0;
  while ((1 << unit_shift) < unit_src) ++unit_shift;
  BlockIndex I{bkeys, bvals, cap, (const ulonglong2*)rec};
  if (block_bits <= 4)
    hipLaunchKernelGGL(k_kernel_map_bi<true>, dim3(kmb_grid(n_out)), dim3(256), 0, s, (const int4*)out_coords, n_out,
                       I, unit_shift, block_bits, sign * step, nbr, (unsigned long long*)n_pairs, mask_out, translate);
  else
    hipLaunchKernelGGL(k_kernel_map_bi<false>, dim3(kmb_grid(n_out)), dim3(256), 0, s, (const int4*)out_coords, n_out,
                       I, unit_shift, block_bits, sign * step, nbr, (unsigned long long*)n_pairs, mask_out, translate);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

