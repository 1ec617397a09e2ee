/*
 * panoptic_hip.h -- C-ABI of libpanoptic_hip.so, the MI355X (gfx950) implementation of the
 * panoptic hot path of prs-eth/PanopticSegForLargeScalePointCloud.
 *
 * The reference has no FFI of its own: its hot path calls third-party native libraries through
 * Python.  Each entry point below replaces one of those call sites; the
 * "replaces:" line names the reference file:line (relative to the reference tree) whose native call
 * it stands in for.  All pointers are raw DEVICE pointers unless marked "host"; sizes are element
 * counts; every function enqueues on `stream` (a hipStream_t passed as void*) and returns a status
 * code.  No ownership is transferred, nothing is allocated behind the caller's back: functions
 * that need scratch take a workspace whose size the matching *_workspace() call reports.
 *
 * Conventions
 *   coords      int32 [n,4] rows (batch, x, y, z)  -- the layout BaseMinkowski._set_input builds
 *               (torch_points3d/applications/minkowski.py:121).
 *   hash table  open addressing, `cap` = pp_hash_capacity(n) slots: keys uint64[cap], vals int32[cap].
 *   kernel map  int32 nbr[K][n_out]  (offset-major; -1 = no neighbour); K = 27 for 3x3x3, offset index
 *               k = (dx+1) + 3(dy+1) + 9(dz+1).
 *   features    float32 row-major [n, C].
 *   weights     "ME layout" float32 [K, Cin, Cout] (MinkowskiConvolution.kernel), packed once per
 *               weight update into MFMA fragment order by pp_pack_weight.
 */
#ifndef PANOPTIC_HIP_H
#define PANOPTIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pp_stream_t; /* hipStream_t */

enum {
  PP_OK = 0,
  PP_ERR_INVALID = 1,  /* bad argument (null pointer, unsupported size) */
  PP_ERR_RANGE = 2,    /* coordinate/batch index not representable in the 64-bit key */
  PP_ERR_HIP = 3,      /* a HIP runtime call failed; see pp_last_error() */
  PP_ERR_WORKSPACE = 4, /* workspace too small */
  PP_UNSUPPORTED = 5    /* not an error: this optional fused form does not serve the shape, nothing was launched */
};

/* Library identity / diagnostics.  pp_version: "panoptic_hip <n> gfx950". */
const char* pp_version(void);
const char* pp_last_error(void);
/* Device-side triad (a[i] = b[i] + s*c[i]) used by bench.py to confirm the HBM roofline denominator. */
int pp_triad(float* a, const float* b, const float* c, float s, int64_t n, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K1  coordinate hash            replaces: ME.SparseTensor(features, coordinates) coordinate-map insert,
 *                                torch_points3d/applications/minkowski.py:121-122
 * info[0] = rows whose coordinate already existed (duplicates), info[1] = rows outside key range.
 * vals[slot] = smallest row index holding that coordinate.
 * ---------------------------------------------------------------------------------------------- */
int64_t pp_hash_capacity(int64_t n);
int pp_hash_build(const int32_t* coords, int64_t n, uint64_t* keys, int32_t* vals, int64_t cap,
                  int32_t* info /*int32[2]*/, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K2  strided output coordinates  replaces: MinkowskiConvolution(stride=2) output-map generation,
 *                                 torch_points3d/modules/MinkowskiEngine/api_modules.py:256-271
 * out_coords = unique(floor(c / ts_out) * ts_out) per batch, ordered by first appearance in `coords`.
 * On return (keys, vals) hash the OUTPUT coordinates (vals = output row), n_out[0] = #output rows,
 * fine_to_coarse[i] = output row of input row i (may be NULL).
 * ---------------------------------------------------------------------------------------------- */
size_t pp_stride_coords_workspace(int64_t n);
int pp_stride_coords(const int32_t* coords, int64_t n, int32_t ts_out, uint64_t* keys, int32_t* vals,
                     int64_t cap, int32_t* out_coords /*[n,4] capacity*/, int32_t* n_out /*int32[1]*/,
                     int32_t* fine_to_coarse, void* workspace, size_t workspace_bytes,
                     int32_t* info /*int32[2]*/, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K3  kernel map (rulebook)       replaces: ME kernel-map build inside every MinkowskiConvolution /
 *                                 MinkowskiConvolutionTranspose, api_modules.py:30-51,259-267
 * nbr[k][o] = row (in the map hashed by keys/vals) of  out_coords[o] + sign * offset_k * step,  or -1.
 * sign=+1: convolution (stride 1: step = tensor stride; stride 2: step = input tensor stride).
 * sign=-1: transposed convolution (mirrored offsets; with a coarse input map this is ME's swapped map).
 * ksize in {1,3}; ksize==1 ignores step/sign (pure coordinate lookup).
 * n_pairs (device int64[1], may be NULL) receives the number of (in, out) pairs P of the map.
 * ---------------------------------------------------------------------------------------------- */
int pp_kernel_map(const int32_t* out_coords, int64_t n_out, const uint64_t* keys, const int32_t* vals,
                  int64_t cap, int32_t ksize, int32_t step, int32_t sign, int32_t* nbr, int64_t* n_pairs,
                  pp_stream_t stream);

/* The two device-wide primitives every compaction / ordering of the library goes through (csrc/pp_scan.hip; hand-written:
 * reduce-then-scan over 4096-element tiles; stable LSD radix sort, 8 bits per pass, tiles sorted in registers by 2-bit
 * splits).  pp_exclusive_scan: out[i] = in[0] + ... + in[i-1] (int32, in place allowed), total (device, may be NULL) = sum.
 * pp_sort_pairs: stable sort of (key, int32 value) pairs by the key bits [0, end_bit); keys uint32 / uint64 (key_bytes);
 * not in place. */
size_t pp_exclusive_scan_workspace(int64_t n);
int pp_exclusive_scan(const int32_t* in, int32_t* out, int64_t n, int32_t* total, void* workspace, size_t workspace_bytes,
                      pp_stream_t stream);
/* Stream compaction and run lengths on that scan.  replaces: the torch.nonzero / unique_consecutive / cumsum / repeat_interleave
 * calls around the embedding clustering (torch_points3d/utils/meanshift_cluster.py:72-123 builds its per-sample lists with the
 * NumPy equivalents; torch_points3d/models/panoptic/PointGroup3heads.py:291-391 selects the "thing" points with a boolean mask).
 * pp_select_indices: idx[0 .. count) = the positions of the non-zero flags in ascending order (idx holds n entries).
 * pp_run_lengths: runs of equal consecutive values: run_id[i] (int32 [n], nullable) = 0-based run of element i, heads[r] = the
 * run's value, starts[r] = its first position, starts[n_runs] = n (heads n, starts n + 1 entries); n_runs int32 [1]. */
size_t pp_select_workspace(int64_t n);
int pp_select_indices(const uint8_t* flags, int64_t n, int64_t* idx, int32_t* count, void* workspace, size_t workspace_bytes,
                      pp_stream_t stream);
int pp_run_lengths(const int64_t* values, int64_t n, int32_t* run_id, int64_t* heads, int32_t* starts, int32_t* n_runs,
                   void* workspace, size_t workspace_bytes, pp_stream_t stream);
size_t pp_sort_pairs_workspace_bytes(int64_t n);
int pp_sort_pairs(const void* keys_in, void* keys_out, int32_t key_bytes, const int32_t* vals_in, int32_t* vals_out, int64_t n,
                  int32_t end_bit, void* workspace, size_t workspace_bytes, pp_stream_t stream);

/* Derived maps (no hash probes).  pp_kernel_map_transpose: out_map[k][in_map[k][o]] = o, i.e. the map of the
 * transposed strided convolution (coarse -> fine, ME's "swapped" kernel map, api_modules.py:288-311) from the
 * strided convolution's map; out_map is int32 [K][n_in], filled with -1 first.  in_order (nullable): in_map is
 * slot-ordered (pp_map_permute) and slot o stands for row in_order[o]. */
int pp_kernel_map_transpose(const int32_t* in_map, int64_t n_out, int32_t K, int64_t n_in, const int32_t* in_order,
                            int32_t* out_map, pp_stream_t stream);
/* 8-wide form of the same map for a stride-2 transposed convolution: a fine row has a coarse neighbour only through offsets
 * whose components are 0 on its even axes and +-1 on its odd ones, i.e. through <= 8 of the 27, fixed by the row's parity class
 * cls = (x odd) | (y odd) << 1 | (z odd) << 2.  map8 int32 [8][n_in]: entry j = (dx>0) | (dy>0)<<1 | (dz>0)<<2 holds
 * coarse row | cls << 28 (-1 = none); key uint32 [n_in] (nullable) = cls << 8 | presence bits -- what pp_map_order sorts by.
 * Same pairs as pp_kernel_map_transpose at 32 instead of 108 bytes per row; consumed by pp_spconv_fwd_t8 after pp_map_permute
 * (K = 8).  n_out (coarse rows) < 2^28.
 * replaces: the transposed kernel maps MinkowskiConvolutionTranspose requests, api_modules.py:259-267 (inference). */
int pp_kernel_map_transpose8(const int32_t* in_map, int64_t n_out, int64_t n_in, const int32_t* in_order, int32_t* map8,
                             uint32_t* key, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K1b/K3c  block index of a level + kernel maps through it (what the coordinate manager uses; the
 * row-level hash of pp_hash_build / pp_kernel_map gives the same maps and stays available).
 * The rows must already be in pp_morton_order(unit, block_bits) order.  Per group of <= 4096 voxels
 * (key >> 12) the index keeps the first row and 64 records {64 bits of the 4096-bit occupancy map in key
 * order, row of the first voxel of that word}; a voxel's row is record.row + popcount(record.bits below
 * its bit): one 16-byte load.  Two steps because the number of blocks sizes the arrays: count (row_block
 * int32 [n]; counts int32[4] = {blocks, duplicated rows, unsorted pairs, rows outside the key range}) then
 * fill (bkeys/bvals [cap], cap = pp_block_index_capacity; start int32 [n_blocks]; rec uint64 [n_blocks*128]:
 * word pairs (bits, row)).
 * pp_kernel_map_bi: same definition as pp_kernel_map (nbr[k][o] = row of out_coords[o] + sign*offset_k*step
 * in the indexed level, -1 if absent or off its lattice).
 * ---------------------------------------------------------------------------------------------- */
size_t pp_block_index_workspace(int64_t n);
int64_t pp_block_index_capacity(int64_t n_blocks);
int pp_block_index_count(const int32_t* coords_sorted, int64_t n, int32_t unit, int32_t block_bits, int32_t* row_block,
                         int32_t* counts /*int32[4]*/, void* workspace, size_t workspace_bytes, pp_stream_t stream);
int pp_block_index_fill(const int32_t* coords_sorted, int64_t n, int32_t unit, int32_t block_bits,
                        const int32_t* row_block, int64_t n_blocks, uint64_t* bkeys, int32_t* bvals, int64_t cap,
                        int32_t* start, uint64_t* rec /*[n_blocks*128]*/,
                        uint64_t* bkey_ord /*[n_blocks] block keys in block order, may be NULL*/, pp_stream_t stream);
/* The next coarser level (tensor stride unit_coarse = 2 x the indexed level's) computed from the index alone: with
 * the parity-block order the coarse bitmaps are bit permutations of the fine ones (replaces K2, pp_stride_coords +
 * sort + pp_block_index_* for strided convolutions, api_modules.py:256-271).  Output arrays have the capacities of
 * the fine level (nb_fine blocks, cap = pp_block_index_capacity(nb_fine), n_fine rows for coords [.,4]);
 * counts = {coarse blocks, coarse rows}.  Rows come out in the coarse level's own order. */
size_t pp_block_index_coarsen_workspace(int64_t nb_fine);
int pp_block_index_coarsen(const uint64_t* f_bkey_ord, const uint64_t* f_rec, int64_t nb_fine, int32_t unit_coarse,
                           int32_t block_bits, uint64_t* bkeys, int32_t* bvals, int64_t cap, int32_t* start,
                           uint64_t* rec, uint64_t* bkey_ord, int32_t* coords, int32_t* counts /*int32[2]*/,
                           void* workspace, size_t workspace_bytes, pp_stream_t stream);
/* The level chain of an encoder in ONE call: `levels` successive coarsenings (tensor stride unit_fine -> 2 unit_fine -> ...), the
 * block and row counts of level l - 1 consumed by level l's launches straight from DEVICE memory (no host read between levels).
 * Every level's outputs have the input level's capacity (nb_fine blocks / n_fine rows, cap = pp_block_index_capacity(nb_fine)) and lie
 * one after the other: bkeys / bvals [levels][cap], start / bkey_ord [levels][nb_fine], rec [levels][nb_fine * 128],
 * coords [levels][n_fine * 4], counts [levels][2] = {blocks, rows}: one read of `counts` sizes all levels.  Contents equal `levels`
 * calls of pp_block_index_coarsen.  workspace: pp_block_index_coarsen_workspace(nb_fine).
 * replaces: the chain of strided coordinate maps an encoder requests one after the other -- MinkowskiConvolution(stride=2) per
 * ResNetDown, modules/MinkowskiEngine/api_modules.py:256-271, reached from applications/minkowski.py:160-196 (backbone: 6 levels)
 * and models/panoptic/PointGroup3heads.py:393-454 (ScorerUnet: 2 levels). */
int pp_block_index_coarsen_chain(const uint64_t* f_bkey_ord, const uint64_t* f_rec, int64_t nb_fine, int64_t n_fine,
                                 int32_t unit_fine, int32_t block_bits, int32_t levels, uint64_t* bkeys, int32_t* bvals,
                                 int64_t cap, int32_t* start, uint64_t* rec, uint64_t* bkey_ord, int32_t* coords, int32_t* counts,
                                 void* workspace, size_t workspace_bytes, pp_stream_t stream);
int pp_kernel_map_bi(const int32_t* out_coords, int64_t n_out, const uint64_t* bkeys, const int32_t* bvals, int64_t cap,
                     const uint64_t* rec, int32_t unit_src,
                     int32_t block_bits, int32_t step, int32_t sign, int32_t* nbr /*[27][n_out]*/,
                     int64_t* n_pairs /*device, may be NULL*/, uint32_t* mask_out /*[n_out] occupied offsets, may be NULL*/,
                     const int32_t* translate /*may be NULL: found rows r are stored as translate[r] (the indexed level's
                                                physical row order, pp_level_permute)*/,
                     pp_stream_t stream);

/* Tile scheduling at map-build time (csrc/pp_maporder.hip).  The convolution executes a kernel offset for a 16-row
 * MFMA tile as soon as one of its rows has that neighbour, so every map gets a SLOT ORDER in which the rows of a tile
 * want the same offsets: inside windows of pp_map_window() consecutive rows, rows are sorted by (neighbour mask with the
 * rarest offset classes most significant, row).
 *   pp_map_mask      mask[o] = bit k set <=> nbr[k][o] >= 0            (pp_kernel_map_bi also emits it as mask_out)
 *   pp_map_order     order[s] = row taking slot s
 *   pp_map_permute   out[k][s] = T(nbr[k][order[s]]), T(v) = v < 0 ? -1 : (translate ? translate[v] : v); order NULL = identity;
 *                    window = the pp_map_window() `order` was built with (LDS-staged form) or 0 (any order)
 *   pp_level_permute coords_out[s] = coords[order[s]], inverse[order[s]] = s   (renumbering of a level: same-level maps
 *                    are stored in the level's own row order, so their convolutions need no indirection)
 * Cross-level maps stay slot-major and pp_spconv_fwd takes `order` as row_order.  Results never depend on the order. */
int32_t pp_map_window(void);
int pp_map_set_window(int32_t window /*1024, 2048, 8192 (default), 16384 or 32768*/);
int pp_map_mask(const int32_t* nbr, int32_t K, int64_t n_out, uint32_t* mask, pp_stream_t stream);
int pp_map_order(const uint32_t* mask, int64_t n, int32_t* order, pp_stream_t stream);
/* the same with an explicit window (1024, 2048, 8192, 16384 or 32768; 4096 is refused, csrc/pp_maporder.hip): the levels' own (same-level) maps take larger windows than
 * the cross-level ones, whose convolutions scatter output rows inside a window; pp_map_permute takes the window used here */
int pp_map_order_window(const uint32_t* mask, int64_t n, int32_t window, int32_t* order, pp_stream_t stream);
/* Compact form of a SAME-LEVEL map for the convolution's prologue (4 + 6 x pairs bytes per row instead of 108): the present entries
 * (neighbour rows) grouped by chunks of 32 output rows, offset-major inside a chunk; start int32 [ceil(n_out / 32) + 1] their
 * offsets (start[last] = pairs of the map), tags uint16 [pairs] = offset index << 6 | output row & 63, mask uint32 [n_out] = the
 * rows' offsets.  _count fills mask and start, _write the entries and tags (capacity: start[last], at most K * n_out). */
size_t pp_map_compact_workspace(int64_t n_out);
int pp_map_compact_count(const int32_t* nbr, int32_t K, int64_t n_out, uint32_t* mask, int32_t* start, void* workspace,
                         size_t workspace_bytes, pp_stream_t stream);
int pp_map_compact_write(const int32_t* nbr, int32_t K, int64_t n_out, const int32_t* start, uint32_t* entries, uint16_t* tags,
                         pp_stream_t stream);
int pp_map_permute(const int32_t* nbr, int32_t K, int64_t n_out, const int32_t* order, const int32_t* translate,
                   int64_t translate_rows /*entries of translate = rows of the level the map's values name; 0 without translate*/,
                   int32_t window, int32_t* out, pp_stream_t stream);
int pp_level_permute(const int32_t* coords, int64_t n, const int32_t* order, int32_t* coords_out, int32_t* inverse,
                     pp_stream_t stream);
/* caller <-> internal row permutation of the input level: perm_out[s] = perm32[order[s]] (NULL = identity for either),
 * inv_out[perm_out[s]] = s  (applications/minkowski.py:193: the output rows must come back in the caller's order) */
int pp_compose_perm(const int32_t* perm32, const int32_t* order, int64_t n, int64_t* perm_out, int64_t* inv_out,
                    pp_stream_t stream);

/* Internal row order of a coordinate level, batch-major: perm[p] = input row holding the p-th smallest key.
 * unit = tensor stride of the level (coordinates are multiples of it).
 * block_bits = 0: plain Morton (Z-) order of coords / unit.
 * block_bits = B >= 2: blocks of 2^B voxels per axis in Z-order; inside a block rows are grouped by the parity of
 *   (x, y, z) / unit, then Z-ordered.  Rows of one parity class use the same offsets of a stride-2 (transposed)
 *   convolution, so 16-row tiles of a class skip the other offsets entirely (useful MFMA work on the up-convolutions
 *   0.14 -> ~0.5), while neighbours stay a few hundred rows apart (L2-resident gathers).
 * The caller-visible row order of stride-1 tensors is unchanged (applications/minkowski.py:193).
 * info[1] = rows outside the key range. */
size_t pp_morton_order_workspace(int64_t n);
int pp_morton_order(const int32_t* coords, int64_t n, int32_t unit, int32_t block_bits, int32_t* perm,
                    int32_t* sorted_coords /*[n,4] or NULL: coords[perm], decoded from the sorted keys (no gather)*/,
                    void* workspace, size_t workspace_bytes, int32_t* info /*int32[2]*/, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K4  sparse convolution forward  replaces: ME ConvolutionForward (gather-GEMM-scatter per offset),
 *                                 api_modules.py:30-51 ; K6 folded BN + ReLU epilogue api_modules.py:40-41 ;
 *                                 K7 ME.cat fused as a second source, api_modules.py:308
 * out[o] = epilogue( sum_k  [in0 | in1][nbr[k][o]] . W_k )
 * epilogue(v) = relu?( v * scale + shift ) + residual      (scale/shift/residual may be NULL)
 * nbr == NULL with K == 1: identity map (1x1 convolution, n_out rows in == rows out).
 * transpose_w: bit 0 -> packed[k] = W_k^T (input gradients) instead of W_k (Cin x Cout); bit 1 -> offsets reversed,
 * packed[k] = W_{K-1-k}: on a same-level map nbr_mirrored[k] == nbr[K-1-k], so transposed stride-1 convolutions and
 * the input gradients of stride-1 convolutions reuse the forward map instead of a flipped copy of it.
 * n_in bounds the gathers: the fast kernel reads rows through a buffer descriptor of n_in * c0 * 4 bytes (missing
 * neighbours come back as hardware-checked zeros); inputs of 4 GiB or more per source take the slower kernel.
 * The packed buffer (pp_packed_weight_floats floats; opaque to the caller) holds three sections when cin % 16 == 0: the fp32 MFMA
 * fragments, the same fragments split EXACTLY into three bfloat16 planes (w == hi + mid + lo) and rounded to nearest-even
 * bfloat16.  Launches with >= 2 sixteen-column tiles per wave (cout >= 32; on two tiles only with >= 32 input channels; the rule itself:
 * pp_spconv_kernel_family) evaluate their fp32 products from the split planes on
 * v_mfma_f32_16x16x32_bf16 -- six bf16 products per fp32 product, fp32 accumulation, error below one fp32 rounding per
 * product (csrc/pp_spconv3.hip; environment PP_CONV_X3=0: v_mfma_f32_16x16x4_f32 everywhere); pp_spconv_fwd_bf16 uses
 * the rounded plane there.  Results are fp32 tensors in every case.
 * ---------------------------------------------------------------------------------------------- */
size_t pp_packed_weight_floats(int32_t K, int32_t cin, int32_t cout);
int pp_pack_weight(const float* weight /*[K,cin,cout]*/, int32_t K, int32_t cin, int32_t cout,
                   int32_t transpose_w, float* packed, pp_stream_t stream);
/* All layers of a model in one launch (training: every weight changes with the optimizer step).  desc int64 [n_desc][6] on the
 * device = {weight pointer, packed pointer, K, cin, cout, flags as pp_pack_weight's transpose_w}; first_block int64 [n_desc+1]
 * = running sum of ceil(pp_packed_weight_floats / 256) (first_block[n_desc] = total_blocks). */
int pp_pack_weights_batched(const int64_t* desc, const int64_t* first_block, int32_t n_desc, int64_t total_blocks,
                            pp_stream_t stream);
int pp_spconv_fwd(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in /*rows of in0 (and in1)*/,
                  const float* packed_weight, const int32_t* nbr, int32_t K, int64_t n_out, int32_t cout, const float* scale,
                  const float* shift, int32_t relu, const float* residual,
                  const int32_t* row_order /*slot order of a cross-level map (pp_map_order): nbr is slot-major and slot s writes
                                              row row_order[s]; NULL = slot s is row s*/, float* out,
                  pp_stream_t stream);
/* pp_spconv_fwd with the 1x1 shortcut ("downsample" branch: 1x1 convolution + BatchNorm of the block's input,
 * api_modules.py:9-82 ResBlock) of a residual block fused into the block's last convolution:
 *   out[o] = relu?( conv(o) * scale + shift ) + residual[o] + ( ds_in[o] . W_ds ) * ds_scale + ds_shift
 * ds_in [n_out, ds_c] fp32 (ds_c % 16 == 0), ds_packed = pp_pack_weight of the [1, ds_c, cout] kernel.  Same-level maps on
 * the unsplit pipelined kernel only: otherwise PP_UNSUPPORTED comes back and nothing has been launched. */
int pp_spconv_fwd_shortcut(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in, const float* packed_weight,
                           const int32_t* nbr, int32_t K, int64_t n_out, int32_t cout, const float* scale, const float* shift,
                           int32_t relu, const float* residual, float* out, int32_t bf16, const float* ds_in, int32_t ds_c,
                           const float* ds_packed, const float* ds_scale, const float* ds_shift, pp_stream_t stream);
/* bfloat16 compute variant (BASELINE.json configs[4], "bf16"): same arguments and fp32 tensors in memory; features and
 * weights are rounded to bfloat16 (nearest even) in registers, products accumulate in fp32 on
 * v_mfma_f32_16x16x16_bf16 -- torch.autocast(bfloat16) semantics for the convolution.  Needs cin % 16 == 0 per source,
 * K <= 28 and < 4 GiB per source (PP_ERR_INVALID otherwise; callers keep the fp32 entry for those layers). */
/* Explicit variants of the pipelined kernel (parity tests, A/B measurements): rows_per_wave in {0, 32, 64} (16-row MFMA
 * tiles per wave x 16), pipeline in {0, 1, 3, 5, 6} (1: loads of step n+1 issued before the MFMAs of step n; 3: additionally
 * the LDS read of the step after that; 5: register ring of depth 3 -- the loads of step n+2 before the MFMAs of step n,
 * inline-assembly loads with hand-counted waits, <= 4 column tiles per wave; 6: LDS-staged feature tiles -- full 128-byte
 * line gathers by buffer_load ... lds into an XOR-swizzled ring, fragments by ds_read_b128; channel counts that are
 * multiples of 32, 32 rows per wave, <= 4 column tiles), split_k in {0, 1, 2, 4, 8} (kernel offsets split over that many waves, partial
 * sums added in a fixed order; needs pp_spconv_set_scratch); 0 = the per-shape choice pp_spconv_fwd makes.  bf16 != 0
 * selects the bfloat16 compute variant.  PP_ERR_INVALID for shapes the pipelined kernel does not take. */
/* pp_spconv_fwd on the 8-wide transposed map (pp_kernel_map_transpose8 -> pp_map_permute(K = 8)); bit-identical
 * to the dense 27-wide form in the same slot order.  cin % 16 == 0, inputs < 4 GiB per source. */
int pp_spconv_fwd_t8(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in, const float* packed_weight,
                     const int32_t* nbr8, int64_t n_out, int32_t cout, const float* scale, const float* shift, int32_t relu,
                     const float* residual, const int32_t* row_order, float* out, int32_t bf16, pp_stream_t stream);
/* pp_spconv_fwd (ds_in NULL) / pp_spconv_fwd_shortcut on the compact form of a same-level map (pp_map_compact_*; K = 27, rows =
 * slots): same results bit for bit, the prologue streams the compact map instead of the dense one. */
int pp_spconv_fwd_cmap(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in, const float* packed_weight,
                       const uint32_t* cm_mask, const int32_t* cm_start, const uint32_t* cm_entries, const uint16_t* cm_tags,
                       int64_t n_out, int32_t cout, const float* scale, const float* shift, int32_t relu, const float* residual,
                       float* out, int32_t bf16, const float* ds_in, int32_t ds_c, const float* ds_packed, const float* ds_scale,
                       const float* ds_shift, pp_stream_t stream);
int pp_spconv_fwd_ex(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in, const float* packed_weight,
                     const int32_t* nbr, int32_t K, int64_t n_out, int32_t cout, const float* scale, const float* shift,
                     int32_t relu, const float* residual, const int32_t* row_order, float* out, int32_t bf16,
                     int32_t rows_per_wave, int32_t pipeline, int32_t split_k, pp_stream_t stream);
/* Optional scratch for small launches (fewer waves than SIMD slots): with a buffer registered here pp_spconv_fwd[_bf16]
 * splits the kernel offsets of such a launch over up to 8 waves per row tile and adds the partial sums in a fixed order
 * (results stay reproducible; they differ from the unsplit sum only by fp32 summation order).  The buffer belongs to
 * the caller, must outlive the launches and serves ONE stream at a time; NULL un-registers.  Needs
 * split * n_out * cout * 4 bytes, otherwise the launch runs unsplit. */
int pp_spconv_set_scratch(void* scratch, size_t bytes);
/* Which kernel the pp_spconv_fwd family runs a launch of this shape on: 1 = k_spconv_x3 (fp32 operands split exactly into three
 * bfloat16 terms, bf16 matrix pipe; >= 2 sixteen-column tiles per wave and, on <= 2 tiles, >= 32 input channels), 2 = k_spconv_x3f
 * (the same arithmetic and bits with the rows gathered as full 128-byte lines through LDS: inputs of whole 32-channel groups, from
 * one column tile per wave up; see pp_spconv_x3_full_lines), 0 = the fp32-MFMA kernels.  The dispatch's own rule (environment overrides included), exported so that a profiler attributes launch times to a
 * kernel family without mirroring it.  replaces: nothing in the reference (ME picks its kernels internally, reached from
 * modules/MinkowskiEngine/api_modules.py:30-51); measurement support for SURVEY.md 8(d). */
int pp_spconv_kernel_family(int32_t c0, int32_t c1, int64_t n_in, int32_t K, int64_t n_out, int32_t cout, int32_t shortcut);
/* The split-operand kernel gathers its rows either in MFMA fragment shape (k_spconv_x3: 16 rows x 4 pieces of 16 bytes per load) or,
 * on dense maps over whole 32-channel groups (c0 % 32 == 0), as full 128-byte lines straight into LDS (k_spconv_x3f: 8 rows x one
 * line per load, fragments read back from LDS) -- same packed weights, same summation order, bit-identical results.  mode 1 / 0
 * switches the full-line form on / off (default on; environment PP_CONV_X3F), any other value only asks; returns the setting before
 * the call.  replaces: nothing in the reference (ME's kernel choice is internal, api_modules.py:30-51); A/B and parity-test support. */
int pp_spconv_x3_full_lines(int32_t mode);
int pp_spconv_fwd_bf16(const float* in0, int32_t c0, const float* in1, int32_t c1, int64_t n_in,
                       const float* packed_weight, const int32_t* nbr, int32_t K, int64_t n_out, int32_t cout,
                       const float* scale, const float* shift, int32_t relu, const float* residual,
                       const int32_t* row_order, float* out, pp_stream_t stream);

/* K5  weight gradient             replaces: ME ConvolutionBackward (dW part), reached from
 *                                 loss.backward(), torch_points3d/models/panoptic/PointGroup3heads.py:636-639
 * dw[k] (+)= sum_o in[nbr[k][o]]^T . dout[o]          dw is float32 [K,cin,cout], zeroed by the callee.
 * in is [n_in,cin] (n_in bounds the gathers: indices in nbr are < n_in), dout is [n_out,cout]. */
int pp_spconv_bwd_weight(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout,
                         const int32_t* nbr, int32_t K, int64_t n_out, float* dw, pp_stream_t stream);
/* bfloat16 compute variant: in and dout rounded to bfloat16 in registers, fp32 accumulation and fp32 dw.
 * Needs a kernel map (nbr != NULL), cout <= 192 and in < 4 GiB. */
int pp_spconv_bwd_weight_bf16(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout,
                              const int32_t* nbr, int32_t K, int64_t n_out, float* dw, pp_stream_t stream);

/* pair-major form of K5 (what the training step uses): the pairs of every offset compacted once per kernel map, then one
 * launch per layer over the lists -- every 16-pair MFMA step is full, and a slot-ordered map needs no re-ordered dout.
 * pp_wgrad_pairs_build: pairs [<= K*n_out][2] int32 = (output row, input row), offset-major and in row order inside an
 *   offset; row_order (nullable, int32 [n_out]): row r of the map is output row row_order[r] (slot-ordered maps);
 *   tile_start int32 [K*ceil(n_out/1024) + 1]: exclusive scan of the pairs per 1024-row tile (last entry = total); the
 *   list of offset k is pairs[tile_start[k*T] .. tile_start[(k+1)*T]), T = ceil(n_out/1024).  K*n_out < 2^31.
 * pp_spconv_bwd_weight_pairs: dw [K,cin,cout] float32 (zeroed by the callee) from those lists; map_rows = the n_out the
 *   lists were built with, n_out = rows of dout; bf16 != 0: operands rounded to bfloat16 in registers.  in, dout < 4 GiB. */
size_t pp_wgrad_pairs_workspace(int32_t K, int64_t n_out);
int pp_wgrad_pairs_build(const int32_t* nbr, int32_t K, int64_t n_out, const int32_t* row_order, int32_t* pairs,
                         int32_t* tile_start, void* workspace, size_t workspace_bytes, pp_stream_t stream);
int pp_spconv_bwd_weight_pairs(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout, int64_t n_out,
                               const int32_t* pairs, const int32_t* tile_start, int32_t K, int64_t map_rows, float* dw,
                               int32_t bf16, pp_stream_t stream);
/* The same weight gradient without float atomics: every block stores its tile sum in `workspace`
 * (pp_spconv_bwd_weight_pairs_det_workspace bytes) and a second launch adds the blocks of an offset in a fixed order, so dw -- and
 * with it the loss trajectory of a training run -- is bit-identical from run to run.  replaces: the same ME backward as above
 * (torch_points3d/models/panoptic/PointGroup3heads.py:552-639 -> loss.backward()). */
size_t pp_spconv_bwd_weight_pairs_det_workspace(int32_t cin, int32_t cout, int32_t K, int64_t map_rows);
int pp_spconv_bwd_weight_pairs_det(const float* in, int32_t cin, int64_t n_in, const float* dout, int32_t cout, int64_t n_out,
                                   const int32_t* pairs, const int32_t* tile_start, int32_t K, int64_t map_rows, float* dw,
                                   int32_t bf16, void* workspace, size_t workspace_bytes, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K6  batch-norm pieces on [n,C]  replaces: ME.MinkowskiBatchNorm (= BatchNorm1d on F), api_modules.py:40,53,269
 * pp_channel_stats: sum[c], sumsq[c] in float64 (training-mode statistics; two-pass inside).
 * pp_affine_act   : y = act(x*scale + shift) + residual; act: 0 none, 1 relu, 2 leaky(slope).
 * pp_bn_bwd_reduce: per-channel sum(dy), sum(dy * x) (float64) for the BN backward.
 * ---------------------------------------------------------------------------------------------- */
int pp_channel_stats(const float* x, int64_t n, int32_t c, double* sum, double* sumsq, pp_stream_t stream);
int pp_affine_act(const float* x, int64_t n, int32_t c, const float* scale, const float* shift,
                  int32_t act, float slope, const float* residual, float* y, pp_stream_t stream);
int pp_bn_bwd_reduce(const float* x, const float* dy, int64_t n, int32_t c, double* sum_dy,
                     double* sum_dy_x, pp_stream_t stream);
/* Training-mode BatchNorm1d over [n,c] (c <= 256, n >= 1), three launches each way, no atomics (run-to-run
 * reproducible).  replaces: torch BatchNorm1d forward/backward under ME.MinkowskiBatchNorm in train mode,
 * api_modules.py:40,53,269 (model.train() path of base_model / PointGroup3heads.py:120-173).
 * fwd: batch mean / biased variance in float64 -> y = act((x-mean)*rstd*weight + bias) (relu != 0 fuses the ReLU);
 *      running_mean/var (nullable pair) updated in place: r = (1-momentum)*r + momentum*stat (unbiased variance);
 *      save_mean/save_rstd [c] float64 are kept for the backward.  weight/bias nullable (affine=False).
 *      num_batches_tracked (nullable, one int64 on the device): incremented by the same launch (nn.BatchNorm's counter).
 * bwd: y_relu non-null masks dy by (y_relu > 0) first; dx [n,c]; dweight/dbias [c] (nullable). */
size_t pp_bn_train_workspace(int64_t n, int32_t c);
int pp_bn_train_fwd(const float* x, int64_t n, int32_t c, const float* weight, const float* bias, double eps,
                    double momentum, float* running_mean, float* running_var, int32_t relu, float* y,
                    double* save_mean, double* save_rstd, int64_t* num_batches_tracked, void* ws, size_t ws_bytes,
                    pp_stream_t stream);
int pp_bn_train_bwd(const float* x, const float* dy, const float* y_relu, int64_t n, int32_t c,
                    const float* weight, const double* save_mean, const double* save_rstd, float* dx,
                    float* dweight, float* dbias, void* ws, size_t ws_bytes, pp_stream_t stream);

/* Weight / bias gradient of a skinny Linear layer y = x W^T + b (cin, cout <= 32, millions of rows): dw [cout,cin] =
 * dy^T x, db [cout] = column sums of dy (db nullable).  replaces: the torch.nn.Linear backward (rocBLAS split-K GEMM with
 * K = n) of the heads' layers in the training step, PointGroup3heads.py:69-81 / core/common_modules/base_modules.py:35-45.
 * Block partials in float64, no atomics (run-to-run reproducible). */
/* Forward / input gradient of the same skinny layers: y [n,cout] = x [n,cin] W^T + bias with weight [cout,cin] (transposed = 0:
 * torch.nn.Linear's forward) or y = x W with weight [cin,cout] read as stored (transposed = 1: dx = dy W of a layer whose weight is
 * [cols of dy, cols of dx]).  bias nullable.  cin, cout <= 32; fixed summation order.  replaces: the rocBLAS / hipBLASLt GEMMs
 * torch.nn.functional.linear and its backward dispatch for the heads' layers in the training step (PointGroup3heads.py:69-81). */
int pp_linear_rows(const float* x, const float* weight, const float* bias, int64_t n, int32_t cin, int32_t cout, int32_t transposed,
                   float* y, pp_stream_t stream);
size_t pp_linear_wgrad_workspace(int64_t n, int32_t cin, int32_t cout);
int pp_linear_wgrad(const float* x, const float* dy, int64_t n, int32_t cin, int32_t cout, float* dw, float* db,
                    void* ws, size_t ws_bytes, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Heads                           replaces: Semantic/Offset/Embed MLP heads in eval mode,
 *                                 PointGroup3heads.py:69-81,106-108 ; core/common_modules/base_modules.py:35-45
 * y = W2 . leaky_relu_0.2( (W1 . x) * scale + shift ) + b2 ; optional log-softmax; optional argmax out.
 * x [n,cin], W1 [chid,cin] (torch Linear layout), W2 [cout,chid]; cin,chid <= 32, cout <= 32.
 * ---------------------------------------------------------------------------------------------- */
int pp_head_mlp(const float* x, int64_t n, int32_t cin, const float* w1, int32_t chid, const float* scale,
                const float* shift, const float* w2, const float* b2, int32_t cout, int32_t log_softmax,
                float* y, int64_t* argmax /*may be NULL*/, pp_stream_t stream);
/* All heads of the model in one pass: x [n_src, c] is read once per output row, index (nullable int64 [n]) names the input
 * row of output row i -- the backbone's features can stay in the coordinate manager's internal row order and the heads
 * deliver their outputs in the caller's.  Per head the arithmetic of pp_head_mlp (bit-identical).  c == 16 (hidden width ==
 * c), 1 to 3 heads; err_flag (device int32) counts index entries outside [0, n_src). */
typedef struct pp_head_t {
  const float* w1;    /* [c, c]    Linear without bias */
  const float* scale; /* [c]       folded BatchNorm */
  const float* shift; /* [c] */
  const float* w2;    /* [cout, c] */
  const float* b2;    /* [cout] or NULL */
  float* y;           /* [n, cout] */
  int64_t* argmax;    /* [n] or NULL */
  int32_t cout;
  int32_t log_softmax;
} pp_head_t;
int pp_heads(const float* x, int64_t n_src, int32_t c, const int64_t* index, int64_t n, const pp_head_t* heads,
             int32_t n_heads, int32_t* err_flag, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K8+K9 region growing            replaces: torch_points_kernels.region_grow (ball_query PARTIAL_DENSE +
 *                                 sequential DFS), call sites PointGroup3heads.py:166-174,185-205,296-304,340-357
 * Exact semantics of SURVEY.md App. C incl. neighbour-list truncation at nsample (lowest indices first):
 * a point's cluster = the smallest-index point that reaches it through directed neighbour-list edges.
 * Clusters ordered by (class ascending, seed index ascending); points ascending inside a cluster.
 * counts[0] = #clusters, counts[1] = #points in clusters.  num_classes > max(labels).
 * ---------------------------------------------------------------------------------------------- */
size_t pp_region_grow_workspace(int64_t n, int32_t nsample);                    /* safe for any input */
size_t pp_region_grow_workspace_for(int64_t n, int64_t n_selected, int32_t nsample); /* when #non-ignored points is known */
int pp_region_grow(const float* pos /*[n,3]*/, const int64_t* labels, const int64_t* batch, int64_t n,
                   const int64_t* ignore_labels, int32_t n_ignore, int32_t num_classes, int32_t nsample,
                   float radius, int32_t min_cluster_size, int32_t* point_cluster /*[n]*/,
                   int32_t* cluster_offsets /*[n+1] capacity*/, int64_t* cluster_points /*[n] capacity*/,
                   int32_t* counts /*int32[2]*/, void* workspace, size_t workspace_bytes,
                   pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K12 flat-kernel mean shift      replaces: sklearn MeanShift(bandwidth, bin_seeding=True).fit(X).labels_
 *                                 via torch_points3d/utils/meanshift_cluster.py:9-18,72-123
 * Points of sample s are rows sample_offsets[s]..sample_offsets[s+1]-1 (host array).  Samples with
 * <= min_points_exclusive (reference: 3) points get label -1 and 0 clusters.
 * labels[i] = cluster id inside its sample (rank in sklearn's (count, centre) descending order).
 * centers: float32 [m, dim] capacity, rows packed per sample at the sample's point offset (optional).
 * ---------------------------------------------------------------------------------------------- */
size_t pp_meanshift_workspace(int64_t m, int32_t dim, int32_t n_samples);
int pp_meanshift(const float* x /*[m,dim]*/, int64_t m, int32_t dim, const int64_t* sample_offsets /*host*/,
                 int32_t n_samples, float bandwidth, int32_t min_points_exclusive, int32_t max_iter,
                 int32_t* labels /*[m]*/, int32_t* n_clusters /*[n_samples]*/, float* centers,
                 void* workspace, size_t workspace_bytes, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K13 HDBSCAN*                    replaces: hdbscan.HDBSCAN(min_cluster_size=15, min_samples=5,
 *                                 cluster_selection_epsilon=0.006).fit_predict(X)
 *                                 via torch_points3d/utils/hdbscan_cluster.py:8-13,117-167
 * Same sample layout as pp_meanshift (host sample_offsets; samples with <= min_points_exclusive
 * points get label -1 and 0 clusters; reference: 3 for cluster_single, 5 for cluster_loop).
 * Euclidean metric, excess-of-mass selection, allow_single_cluster=False.
 * count_self: 1 = the core distance counts the point itself among its min_samples neighbours
 * (hdbscan's Prim paths, sklearn's port); 0 = it does not (hdbscan's Boruvka path).
 * Equal-weight tree edges are ordered by (weight, min index, max index) -- see DESIGN.md.
 * labels[i] = cluster id inside the sample (ascending condensed-tree id) or -1 (noise).
 * Cost is O(n^2) distance evaluations per Boruvka round per sample (float64, exact).
 * ---------------------------------------------------------------------------------------------- */
size_t pp_hdbscan_workspace(int64_t m, int32_t n_samples);
int pp_hdbscan(const float* x /*[m,dim]*/, int64_t m, int32_t dim, const int64_t* sample_offsets /*host*/,
               int32_t n_samples, int32_t min_points_exclusive, int32_t min_cluster_size,
               int32_t min_samples, int32_t count_self, double cluster_selection_epsilon,
               int32_t* labels /*[m]*/, int32_t* n_clusters /*[n_samples]*/, void* workspace,
               size_t workspace_bytes, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * f1  voxelisation + cylinder cutting (the step before the path)
 * pp_voxelize       replaces: GridSampling3D(size, quantize_coords=True),
 *                   torch_points3d/core/data_transform/grid_transform.py:181-198 (+ torch_geometric voxel_grid /
 *                   consecutive_cluster): coords = round-half-even(pos / size); one row per occupied voxel ordered
 *                   by (batch, z, y, x); rep_index[v] = LAST input point of voxel v (the reference shuffles first);
 *                   inverse[i] = voxel row of point i; counts = {voxels, points outside the 16-bit range}.
 *                   coords int32 [n,4] / rep_index [n] are capacities.
 * pp_cylinder_pairs replaces: CylinderSampling (KDTree.query_radius, inclusive) for ALL centres at once,
 *                   transforms.py:388-441: (point, cylinder) pairs, point-major; call once with NULL outputs to
 *                   get n_pairs, then with buffers; pp_group_by_key(pair_cyl, pair_point) gives the index lists.
 * ---------------------------------------------------------------------------------------------- */
size_t pp_voxelize_workspace(int64_t n);
int pp_voxelize(const float* pos /*[n,3]*/, const int64_t* batch /*[n] or NULL*/, int64_t n, float voxel_size,
                int32_t* coords, int32_t* rep_index, int32_t* inverse /*[n]*/, int32_t* counts /*int32[2]*/,
                void* workspace, size_t workspace_bytes, pp_stream_t stream);
size_t pp_cylinder_pairs_workspace(int64_t n);
int pp_cylinder_pairs(const float* pos /*[n,3]*/, int64_t n, const float* centres_xy /*[n_cyl,2]*/, int32_t n_cyl,
                      float radius, int64_t* pair_point, int32_t* pair_cyl, int64_t capacity, int32_t* n_pairs,
                      void* workspace, size_t workspace_bytes, pp_stream_t stream);

/* Group points by a small integer key into CSR form (stable: ascending point order inside a group).
 * key[i] in [0,n_groups) or -1 (dropped).  ids[i] (int64) is what gets written (NULL -> i).
 * offsets [n_groups+1], out [n] capacity, total[0] = #kept.  Keys >= n_groups are dropped too, but they are a caller
 * error: n_out_of_range[0] (device, nullable) receives their number.  Used to turn labels into the
 * List[LongTensor] the reference APIs return (meanshift_cluster.py:102-111). */
size_t pp_group_by_key_workspace(int64_t n);
int pp_group_by_key(const int32_t* key, const int64_t* ids, int64_t n, int32_t n_groups, int32_t* offsets,
                    int64_t* out, int32_t* total, int32_t* n_out_of_range, void* workspace, size_t workspace_bytes,
                    pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K11 segment reductions          replaces: torch_scatter.scatter(src, index, dim=0, reduce=...),
 *                                 PointGroup3heads.py:419-452 ; core/losses/panoptic_losses.py:260,276
 * reduce: 0 sum, 1 mean, 2 max.  index int64 [n] in [0,n_seg).  out [n_seg,c]; empty segments -> 0.
 * arg (int64 [n_seg,c], may be NULL) receives the arg-max row for reduce=2 (needed by its backward).
 * ---------------------------------------------------------------------------------------------- */
size_t pp_segment_reduce_workspace(int64_t n_seg);
int pp_segment_reduce(const float* src, const int64_t* index, int64_t n, int32_t c, int64_t n_seg,
                      int32_t reduce, float* out, int64_t* arg, void* workspace, size_t workspace_bytes,
                      pp_stream_t stream);
/* Same, for index values the caller has validated already (e.g. the inverse ids of a unique(), or an index a checked
 * call has just accepted -- the backward of a gather): rows with an out-of-range id are skipped silently and the call
 * does not synchronise the stream. */
int pp_segment_reduce_unchecked(const float* src, const int64_t* index, int64_t n, int32_t c, int64_t n_seg,
                                int32_t reduce, float* out, int64_t* arg, void* workspace, size_t workspace_bytes,
                                pp_stream_t stream);
/* The same sums / means without atomics: rows int64 [n] = the source rows grouped by segment in ascending row order,
 * offsets int32 [n_seg + 1] (pp_group_by_key of the segment ids).  One workgroup per segment adds in an order that depends on
 * (segment size, c) only: bit-reproducible run to run (the atomic form above is not).  Empty segments -> 0. */
int pp_segment_sum_ordered(const float* src, const int64_t* rows, const int32_t* offsets, int64_t n_seg, int32_t c, int32_t mean,
                           float* out, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K10 instance IoU                replaces: torch_points_kernels.instance_iou,
 *                                 core/losses/panoptic_losses.py:37 ; metrics/panoptic_tracker_pointgroup_npm3d.py:681
 * Proposals in CSR (offsets int32 [n_prop+1], points int64).  gt_instances int64 [n] (0 = none, ids 1..k
 * per batch element), gt_offsets int32 [n_batch+1] = cumsum of per-sample GT counts (device),
 * gt_sizes int32 [total_gt] (device).  iou float32 [n_prop, total_gt] (zeroed by the callee).
 * ---------------------------------------------------------------------------------------------- */
int pp_instance_iou(const int32_t* prop_offsets, const int64_t* prop_points, int32_t n_prop,
                    const int64_t* gt_instances, const int64_t* batch, const int32_t* gt_offsets,
                    const int32_t* gt_sizes, int32_t total_gt, float* iou, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K14 proposal x proposal intersections   replaces: dense torch.mm(mask, mask^T) in
 *                                 PanopticResults.get_instances, models/panoptic/structure_3heads.py:40-60
 * inter int32 [n_prop, n_prop] (zeroed by the callee) from the point->proposal incidence;
 * n_points = size of the point index space.
 * ---------------------------------------------------------------------------------------------- */
size_t pp_proposal_intersections_workspace(int64_t total_points, int64_t n_points);
int pp_proposal_intersections(const int32_t* prop_offsets, const int64_t* prop_points, int32_t n_prop,
                              int64_t n_points, int32_t* inter, void* workspace, size_t workspace_bytes,
                              pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K14a front end of the proposal scorer (csrc/pp_proposals.hip)
 *                                 replaces: the per-proposal batch assembly of PointGroup3heads._compute_score,
 *                                 models/panoptic/PointGroup3heads.py:393-454 (every proposal = one batch element: batch index,
 *                                 coordinates and features of its points)
 * Proposals with identical point lists get identical scores; one representative per list is scored.
 * pp_proposals_unique: rep int64 [n_prop] = smallest index of a proposal with exactly the same list (order included; found by
 *   (size, two 64-bit sum hashes) and verified entry by entry, so exact), pos_of int64 [n_prop] = position of rep[p] among the
 *   kept proposals (those with rep[p] == p, in index order), uniq_offsets int32 [n_prop + 1] = CSR offsets of the kept lists
 *   (first counts[0] + 1 entries valid), counts int32 [3] (device) = {kept proposals, their entries, points outside
 *   [0, n_points)}.  No host synchronisation.
 * pp_proposals_emit: the kept lists (out_points int64 [counts[1]]), their batch index (out_batch int64, = position of the
 *   proposal) and, with coords int32 [n_points][3], the (batch, x, y, z) rows of the scorer's input (out_coords4 int32
 *   [counts[1]][4]; both NULL to skip).  Call only when counts[2] == 0.
 * ---------------------------------------------------------------------------------------------- */
size_t pp_proposals_unique_workspace(int64_t n_prop);
int pp_proposals_unique(const int32_t* prop_offsets, const int64_t* prop_points, int64_t n_prop, int64_t n_points, int64_t* rep,
                        int64_t* pos_of, int32_t* uniq_offsets, int32_t* counts, void* workspace, size_t workspace_bytes,
                        pp_stream_t stream);
int pp_proposals_emit(const int32_t* prop_offsets, const int64_t* prop_points, int64_t n_prop, const int64_t* rep,
                      const int64_t* pos_of, const int32_t* uniq_offsets, const int32_t* coords, int64_t* out_points,
                      int64_t* out_batch, int32_t* out_coords4, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K14b proposal overlaps, NMS and painting on the device (csrc/pp_nms.hip)
 *                                 replaces: PanopticResults.get_instances + non_max_suppression,
 *                                 models/panoptic/structure_3heads.py:6-71, and get_cur_ins_pre_label,
 *                                 metrics/panoptic_tracker_pointgroup_npm3d.py:326-337
 * pp_proposal_pairs: every pair (a < b) of proposals sharing at least one point, with |a n b|, from the point -> proposal
 *   incidence (sparse form of mask @ mask^T).  total_entries = prop_offsets[n_prop].  pair_* have room for
 *   pp_proposal_pairs_capacity(n_prop) triples; n_pairs (device int32) receives their number, in no particular order.
 *   prop_of_entry int32 [total_entries] = proposal of every CSR entry.  info int32[4] (device): [0] points belonging to
 *   more than 8 proposals, [1] pair-table overflow, [2] point ids outside [0, n_points), [3] (pp_nms_paint) proposals
 *   whose batch element is outside [0, n_groups) -- all must be 0.
 * pp_nms_paint: per batch element (batch[first point of the proposal]; NULL = one group) greedy NMS by descending score
 *   over the pairs with IoU > nms_threshold (float32), then size > min_cluster_points and score > min_score; the kept
 *   proposals are ranked by ascending score and painted: labels[p] = largest rank covering p, -1 = none (ids restart at 0
 *   per batch element).  counts[g] = instances of group g, rank[q] = rank or -1.  scores NULL (no ScoreNet): every proposal
 *   is an instance, ranked in proposal order.  Equal scores: visited in descending proposal id (numpy's argsort()[::-1]
 *   on small arrays); the reference's introsort leaves larger tie groups implementation-defined.
 * ---------------------------------------------------------------------------------------------- */
int64_t pp_proposal_pairs_capacity(int32_t n_prop);
size_t pp_proposal_pairs_workspace(int64_t total_entries, int64_t n_points, int32_t n_prop);
int pp_proposal_pairs(const int32_t* prop_offsets, const int64_t* prop_points, int32_t n_prop, int64_t total_entries,
                      int64_t n_points, int32_t* prop_of_entry, int32_t* pair_a, int32_t* pair_b, int32_t* pair_inter,
                      int32_t* n_pairs, int32_t* info, void* workspace, size_t workspace_bytes, pp_stream_t stream);
size_t pp_nms_paint_workspace(int32_t n_prop, int32_t n_groups, int64_t pair_capacity);
int pp_nms_paint(const int32_t* prop_offsets, const int64_t* prop_points, int32_t n_prop, int64_t total_entries,
                 int64_t n_points, const int32_t* prop_of_entry, const int32_t* pair_a, const int32_t* pair_b,
                 const int32_t* pair_inter, const int32_t* n_pairs, int64_t pair_capacity, const int64_t* batch,
                 int32_t n_groups, const float* scores, float nms_threshold, int32_t min_cluster_points, float min_score,
                 int32_t* labels, int32_t* counts, int32_t* rank, int32_t* info, void* workspace, size_t workspace_bytes,
                 pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * f2 / f3  scene assembly and final evaluation on the device (csrc/pp_eval.hip)
 * pp_histogram2d   out[a[i]][b[i]] += 1 (int64 [na][nb], zeroed by the callee): the class confusion matrix and the
 *                  instance x class tables of the final evaluation, torch_points3d/datasets/panoptic/npm3d.py:107-397,
 *                  metrics/panoptic_tracker_pointgroup_npm3d.py:711-879.  Rows with a negative label are skipped.
 *                  info int32[2] = {rows skipped, rows out of range (must be 0)}.
 * pp_pair_counts   distinct (a[i], b[i]) pairs (both >= 0, b < nb) with their multiplicities: the (prediction,
 *                  ground truth) instance contingency table of npm3d.py:232-300 in sparse form.  Outputs have `capacity`
 *                  slots, n_pairs (device int32) pairs are written in no particular order.
 *                  info int32[2] = {overflow: raise capacity, rows with b >= nb}.
 * pp_block_merge   block_merging of metrics/panoptic_tracker_pointgroup_npm3d.py:339-452 for ONE cylinder, in place on
 *                  the scene labels (int64, -1 = none): blocks must be fed in the original block order.  block_labels
 *                  int32 in [-1, n).  max_instance: device int64[1], carried from block to block.  state int32[8] (device):
 *                  {any_has, any_none, any_valid, t_num, overflow, bad ids, n_pairs, 0}; overflow / bad must be 0.
 * ---------------------------------------------------------------------------------------------- */
int pp_histogram2d(const int64_t* a, const int64_t* b, int64_t n, int32_t na, int32_t nb, int64_t* out, int32_t* info,
                   pp_stream_t stream);
size_t pp_pair_counts_workspace(int64_t capacity);
int pp_pair_counts(const int64_t* a, const int64_t* b, int64_t n, int64_t nb, int64_t capacity, int64_t* pair_a,
                   int64_t* pair_b, int64_t* count, int32_t* n_pairs, int32_t* info, void* workspace, size_t workspace_bytes,
                   pp_stream_t stream);
size_t pp_block_merge_workspace(int64_t n);
int pp_block_merge(const int64_t* origin_ids, const int32_t* block_labels, int64_t n, int64_t* scene_labels, int64_t n_scene,
                   int64_t* max_instance, int32_t* state, void* workspace, size_t workspace_bytes, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row gather                      replaces: features[perm] / features[inverse] around ME.SparseTensor (.F in caller order,
 *                                 applications/minkowski.py:193) and backbone_features[cluster] in _compute_score,
 *                                 PointGroup3heads.py:400-410
 * out[i] = src[index[i]] for float32 rows of c (multiple of 4) channels; index int64 in [0, n_src).  Out-of-range
 * indices are counted in *err_flag (device int32, zeroed by the caller) and their rows left untouched.
 * ---------------------------------------------------------------------------------------------- */
int pp_gather_rows(const float* src, int64_t n_src, int32_t c, const int64_t* index, int64_t n, float* out,
                   int32_t* err_flag, pp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * f2  exact 1-nearest neighbour     replaces: torch_geometric knn(x=ref, y=query, k=1) / knn_interpolate(k=1) of the
 *                                  full-resolution back-projection, metrics/panoptic_tracker_pointgroup_npm3d.py:564-566,
 *                                  593-618; KD-tree query of the cylinder centre label, data_transform/transforms.py:239-240
 * ref [n_ref,dim], query [n_query,dim] float32, dim in {2,3}.  idx[q] = index of the nearest reference point, dist2[q] =
 * ((dx*dx + dy*dy) + dz*dz) in float32 without FMA; ties -> smallest reference index.  cell > 0 = edge of the search
 * grid (2-4x the reference spacing).  max_dist > 0: queries with no reference point within max_dist get idx -1 and
 * dist2 +inf; max_dist <= 0: unbounded exact search.  n_ref == 0: every idx is -1.
 * ---------------------------------------------------------------------------------------------- */
size_t pp_nearest_workspace(int64_t n_ref);
int pp_nearest(const float* ref, int64_t n_ref, const float* query, int64_t n_query, int32_t dim, float cell,
               float max_dist, int64_t* idx, float* dist2, void* workspace, size_t workspace_bytes, pp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PANOPTIC_HIP_H */
