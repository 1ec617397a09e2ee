"""MinkowskiEngine-compatible module surface backed by libpanoptic_hip.so (MI355X).

Only the symbols the reference's hot path uses are provided:
  SparseTensor(features, coordinates, device) with .F .C and `+`     applications/minkowski.py:121-122, api_modules.py:79-81
  MinkowskiConvolution / MinkowskiConvolutionTranspose (.kernel)      api_modules.py:30-51,259-267
  MinkowskiBatchNorm (.bn = BatchNorm1d, settable momentum)           api_modules.py:40 ; core/schedulers/bn_schedulers.py:6-31
  MinkowskiReLU, MinkowskiNetwork, cat, utils.kaiming_normal_        api_modules.py:41,308 ; applications/minkowski.py:104-111
  (MinkowskiLinear, MinkowskiSigmoid, MinkowskiGlobalPooling, MinkowskiBroadcastMultiplication: imported by
   api_modules.py:176-186 for SE blocks that no panoptic config uses; provided as thin modules.)

Not MinkowskiEngine's object model: one CoordinateManager per input holds flat device arrays (COO rows, an
open-addressing hash per tensor stride, offset-major kernel maps) shared by every tensor derived from that input.
Use as a drop-in:   import panopticsegforlargescalepointcloud_amd.MinkowskiEngine as ME
or                  sys.modules["MinkowskiEngine"] = panopticsegforlargescalepointcloud_amd.MinkowskiEngine
"""
import enum
import math
import os
import threading

import torch
from torch import nn

from .. import ops
from . import utils  # noqa: F401


# internal row order of every coordinate level: parity-grouped blocks of 2^ORDER_BLOCK_BITS voxels (0 = plain Z-order,
# -1 = caller order with row-level hash tables)
ORDER_BLOCK_BITS = int(os.environ.get("PP_ORDER_BLOCK", "4"))
# build coarser levels from the finer level's block index (PP_COARSEN=0: hash + sort path, for A/B runs)
COARSEN_FROM_INDEX = os.environ.get("PP_COARSEN", "1") != "0"
# inference with a prefetch plan: all the coarser levels the plan asks for come from ONE library call (ops.block_index_coarsen_chain:
# level sizes chained in device memory) and one host read, instead of a call and a read per level.  Measured (round 6, two
# alternating pairs each, profiles/r06_ab_level_chain.txt): one rank's share of the 8-GPU scene (1.18 M voxels) 22.25 -> 21.60 ms per
# step, the full 9.8 M-voxel scene 114.7 -> 116.1 ms -- every level of the chain is allocated at the input level's capacity, and at
# ~10 M rows that costs more than the five host reads it saves (the builder threads hide those behind the previous batch).  So the
# chain serves inputs up to PP_LEVEL_CHAIN_MAX_ROWS rows (default 3 M: tile batches of a sharded scene, C2, the scorer's inputs
# there); PP_LEVEL_CHAIN=0 turns it off, PP_LEVEL_CHAIN_MAX_ROWS=0 lifts the bound (A/B runs).
LEVEL_CHAIN = os.environ.get("PP_LEVEL_CHAIN", "1") != "0"
LEVEL_CHAIN_MAX_ROWS = int(os.environ.get("PP_LEVEL_CHAIN_MAX_ROWS", "3000000"))
# tile scheduling at map-build time (csrc/pp_maporder.hip): every kernel map gets a slot order in which the 16 rows of
# an MFMA tile want the same offsets; a level's physical row order is the slot order of its same-level map, cross-level
# maps carry their own order (`nbr.pp_order`).  PP_MAP_ORDER=0 keeps the plain block order (A/B runs).
MAP_ORDER = os.environ.get("PP_MAP_ORDER", "1") != "0"
MAP_ORDER_MIN_ROWS = int(os.environ.get("PP_MAP_ORDER_MIN_ROWS", "50000"))  # smaller levels gain nothing from it
# rows per sort window of a level's OWN order (its same-level map; 1024, 2048, 8192, 16384 or 32768 -- 4096 is refused by the library).  Larger windows give purer 16-row tiles
# (executed / useful tile rows at tensor stride 1: 2.00 at 8192, 1.82 at 32768, profiles/r03_row_cache_model.txt) and gathers that
# leave the L2 more often: measured per setting in profiles/r04_ab_same_window.txt (time) and r04_window_traffic.txt (HBM bytes).
# 8192 at tensor stride 1 (16 channels: bound by the gathers) and 16384 at the coarser levels is the fastest; 32768 there costs
# 16 % more fetched bytes for the same time.  Cross-level maps keep pp_map_window().
# PP_SAME_WINDOW = "<tensor stride 1>,<coarser levels>" (or one number for both).
_sw = [int(v) for v in os.environ.get("PP_SAME_WINDOW", "8192,16384").split(",")]
SAME_WINDOW = (_sw[0], _sw[-1])
if os.environ.get("PP_MAP_WINDOW"):  # rows per sort window (1024 | 2048 | 4096 | 8192), A/B runs
    ops._lib.check(ops._lib.load().pp_map_set_window(int(os.environ["PP_MAP_WINDOW"])), "pp_map_set_window")
# compute dtype of the sparse convolutions: "fp32" (the reference's; parity runs) or "bf16" (BASELINE.json configs[4]):
# operands rounded to bfloat16 in registers, fp32 accumulation, fp32 tensors in memory -- what torch.autocast(bfloat16)
# does to a convolution.  `conv_autocast()` switches it for a region of code.
_CONV_BF16 = [os.environ.get("PP_CONV_DTYPE", "fp32").lower() == "bf16"]
# inference: coarser levels and their kernel maps are built on a side stream by a worker thread that replays the
# requests of the model's previous forward, while the main stream already runs the convolutions of the finer levels
MAP_PREFETCH = os.environ.get("PP_MAP_PREFETCH", "1") != "0"
# inference: the map of a stride-2 transposed convolution in its 8-wide form (a fine row has <= 8 coarse neighbours, fixed by
# its parity class): 32 instead of 108 bytes per row through transpose, mask, sort, permute and the convolution's prologue
MAP_T8 = os.environ.get("PP_MAP_T8", "1") != "0"
# strided maps are slot-ordered when the fine level has fewer than this many rows per coarse row (0 = never; A/B runs)
STRIDED_ORDER_RATIO = float(os.environ.get("PP_STRIDED_ORDER_RATIO", "1.7"))
_SIDE_STREAMS = {}
_NO_READS_MARK = os.environ.get("PP_NO_READS_MARK", "0") == "1"  # evidence runs only: CoordinateManager._reads as in round 5


_STREAM_TLS = threading.local()


def _raw_stream():
    """raw handle of the calling thread's current HIP stream (two C calls, no Python objects)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _current_stream():
    """torch.cuda.current_stream() without its cost: the public call resolves the device index through
    torch._utils._get_current_device_index -> torch.cuda.is_available() -> a device-count query, ~12 us each, and this module
    asked for it ~200 times per step (once per map / level access and per readiness event: 2.4 ms of host time per step of a
    dispatch-paced configuration).  The Stream object is cached per thread and re-made only when the raw handle changes."""
    raw = _raw_stream()
    c = getattr(_STREAM_TLS, "cur", None)
    if c is None or c[0] != raw:
        c = _STREAM_TLS.cur = (raw, torch.cuda.current_stream())
    return c[1]


def _side_stream(device):
    """the stream the coordinate levels / kernel maps are prefetched on (PP_SIDE_PRIORITY=high|default, A/B runs).
    Round 2 measured high priority 1 ms ahead (161.6 vs 162.6 ms per bench step); since the backbone's maps are built during the
    previous batch (PreparedCoordinates) the default priority gives the same step with 1 % less convolution time
    (profiles/r04_ab_side_priority.txt: 133.5 vs 133.7 ms, convolutions 106.4 vs 107.5 ms over five alternating pairs)."""
    s = _SIDE_STREAMS.get(device)
    if s is None:
        prio = -1 if os.environ.get("PP_SIDE_PRIORITY", "default") == "high" else 0
        s = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device, priority=prio)
    return s


_PREP_STREAMS = {}


def _prep_stream(device, which=0):
    """the streams the NEXT batch's coordinate manager is built on (PreparedCoordinates): 0 = its constructor (input level),
    1 = its level / map builder"""
    s = _PREP_STREAMS.get((device, which))
    if s is None:
        # (PP_PREP_PRIORITY: HIP stream priority of the preparation streams, A/B runs; larger = lower)
        s = _PREP_STREAMS[(device, which)] = torch.cuda.Stream(device=device, priority=int(os.environ.get("PP_PREP_PRIORITY", "0")))
    return s


class conv_autocast:
    """with ME.conv_autocast(): ... -- bfloat16 compute for every sparse convolution launched inside (forward, input
    and weight gradients of the layers the bf16 kernels take; the 4-channel input layer stays fp32)."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = _CONV_BF16[0]
        _CONV_BF16[0] = self.enabled
        return self

    def __exit__(self, *exc):
        _CONV_BF16[0] = self.prev
        return False


# ------------------------------------------------------------------------------------------------
# coordinate manager
# ------------------------------------------------------------------------------------------------
class _Level:
    """One coordinate level.  `coords` int32 [n,4] in the level's PHYSICAL row order (the order of every feature matrix
    of this level).  `index` (ops.BlockIndex) is built over the block / Z-order the rows are created in; `phys_of`
    (int32 [n], None = identity) maps a row of that order to its physical row -- with map ordering on, the physical
    order is the slot order of the level's same-level kernel map, kept in `same_map`.  `table`: row-level hash
    (PP_ORDER_BLOCK=-1: caller order, hash probing -- the first design, kept for A/B runs)."""
    __slots__ = ("coords", "table", "index", "n", "phys_of", "same_map")

    def __init__(self, coords, table=None, index=None, phys_of=None, same_map=None):
        self.coords = coords
        self.table = table
        self.index = index
        self.n = coords.shape[0]
        self.phys_of = phys_of
        self.same_map = same_map


# PP_COMPACT_MAPS=1 (off by default): same-level maps also get their compact form (ops.map_compact), which the convolutions'
# prologue then streams instead of the dense [27, n] map -- 4 + 6 x pairs bytes per row instead of 108.  Built, parity-tested and
# measured in round 4 (profiles/r04_ab_compact_maps.txt): bit-identical results, a third of the map bytes, and a step that is
# 12 - 19 ms SLOWER (146 - 153 against 133.5 ms; convolutions at 0.40 instead of 0.44 of the MFMA peak): the prologue is bound
# by its chain of dependent loads, not by bytes -- chunk offsets, then entries, then a second batch of entries, where the dense
# map's 28 loads per lane are independent and in flight together -- and the two compaction passes run beside the convolutions.
COMPACT_MAPS = os.environ.get("PP_COMPACT_MAPS", "0") != "0"
COMPACT_MIN_ROWS = int(os.environ.get("PP_COMPACT_MIN_ROWS", "65536"))


def _cmap_tensors(m):
    c = getattr(m, "pp_cmap", None)
    return [] if c is None else [c.mask, c.start, c.entries, c.tags]


def _with_compact(m):
    if COMPACT_MAPS and not torch.is_grad_enabled() and m.shape[0] == 27 and m.shape[1] >= COMPACT_MIN_ROWS \
            and 27 * m.shape[1] < (1 << 31):
        m.pp_cmap = ops.map_compact(m)
    return m


def _order_level(coords_m, index, ts):
    """physical order of a level from its same-level map: (coords_p, order, phys_of, finish) -- `finish()` returns the
    same-level map in physical ids.  The level (coords_p, phys_of) is usable before that last pass has run: a strided map
    onto the level only needs its row order, so the caller can build it first (the scorer's first convolution is strided
    and waits for exactly that chain)."""
    nbr_m = ops.kernel_map_bi(coords_m, index, 3, ts, 1, want_mask=True)
    order = ops.map_order(nbr_m.pp_mask, window=SAME_WINDOW[0 if ts == 1 else 1])
    del nbr_m.pp_mask
    coords_p, phys_of = ops.level_permute(coords_m, order)
    finish = lambda: ops.map_permute(nbr_m, order, translate=phys_of)  # noqa: E731
    finish.reads = (nbr_m, order, phys_of)  # (what a caller on ANOTHER stream must mark as in use there: CoordinateManager._reads)
    return coords_p, order, phys_of, finish


_IDENTITY_PAIRS = {}


def _identity_pairs(n, device):
    """pair lists (ops.wgrad_pairs) of the identity map of n rows: what a 1x1 convolution's weight gradient sums over"""
    key = (int(n), str(device))
    wp = _IDENTITY_PAIRS.get(key)
    if wp is None:
        if len(_IDENTITY_PAIRS) >= 16:
            _IDENTITY_PAIRS.clear()
        wp = _IDENTITY_PAIRS[key] = ops.wgrad_pairs(torch.arange(n, dtype=torch.int32, device=device).view(1, n), 1)
    return wp


class _PermuteRowsFn(torch.autograd.Function):
    """x[perm] for a full permutation: the gradient is one more gather (dy[inv]) instead of torch's sort-based
    index_put(accumulate=True)."""

    @staticmethod
    def forward(ctx, x, perm, inv):
        ctx.inv = inv
        return ops.gather_rows(x, perm)

    @staticmethod
    def backward(ctx, dy):
        return ops.gather_rows(dy.contiguous(), ctx.inv), None, None


class _GatherRowsFn(torch.autograd.Function):
    """x[index] with repeated rows: the gradient is an atomic row scatter-add (index_add_)."""

    @staticmethod
    def forward(ctx, x, index):
        ctx.index, ctx.n = index, x.shape[0]
        return ops.gather_rows(x, index)

    @staticmethod
    def backward(ctx, dy):
        return torch.zeros((ctx.n,) + dy.shape[1:], dtype=dy.dtype, device=dy.device).index_add_(0, ctx.index, dy), None


def _permute_rows(x, perm, inv):
    return _PermuteRowsFn.apply(x, perm, inv) if x.requires_grad else ops.gather_rows(x, perm)


class CoordinateManager:
    """Coordinate levels (tensor stride -> COO rows + hash) and kernel maps, cached for one input.

    Rows are kept in batch-major Morton (Z-) order internally (`perm`: internal row p = caller row perm[p]), so 16
    consecutive rows form a compact surface patch -- the convolution skips kernel offsets that are empty for a whole
    tile and gathered neighbours stay L2-resident.  Caller-visible order at tensor stride 1 is unchanged.

    map(ts_from, ts_to, ksize, sign)[k, o] = row at level ts_from of  coords[ts_to][o] + sign*offset_k*step,
    step = min(ts_from, ts_to).  sign=+1 is a convolution, sign=-1 a transposed convolution; the map needed
    for the input gradient of map(A->B, sign) is map(B->A, -sign).  Only map(ts,ts,+1) and map(ts,2ts,+1) are built
    with hash probes; mirrored and transposed maps are derived (flip along k / pp_kernel_map_transpose)."""

    def __init__(self, coords, reorder=True, prefetch_plan=None, side_stream=None):
        """prefetch_plan: the request log of an earlier inference pass of the same model (see `prefetch`).  Given here, the
        builder thread starts as soon as the input level's block index exists: the next coarser level (coarsening, its
        own slot order) depends on nothing else, so it is built on the side stream WHILE this constructor orders the input
        level on the caller's stream -- the first strided convolution used to wait for that chain (1.1 ms of the bench
        step in the backbone, 2.3 ms in the scorer, whose first convolution is the strided one)."""
        if coords.dtype != torch.int32:
            coords = coords.to(torch.int32)
        coords = coords.contiguous()
        self.orig_coords = coords
        self.perm = self.inv_perm = None
        self.sorted = bool(reorder) and ORDER_BLOCK_BITS >= 0
        # prefetch support: one lock around level / map construction, an event per built item (the builder's stream
        # may not be the consumer's), an optional log of the requests (the plan replayed by the next forward)
        self.levels = {}
        self.maps = {}
        self._pending_same = {}
        self._lock = threading.RLock()
        self._ready = {}
        self._log = None
        self._worker = None
        self._side = side_stream  # the builder's stream (None: the device's shared side stream)
        self._side_built = False  # anything built on the prefetch stream? (otherwise no cross-stream bookkeeping is needed)
        self._worker_err = None
        self._input_final = threading.Event()  # set when levels[1] and its same-level map are the final ones
        self._chain = {}        # tensor stride -> (BlockIndex, coords) built ahead by the level chain, not yet made a level
        self._chain_depth = {}  # tensor stride -> how many successive stride-2 levels the plan asks for from there
        if prefetch_plan and LEVEL_CHAIN and (LEVEL_CHAIN_MAX_ROWS <= 0 or coords.shape[0] <= LEVEL_CHAIN_MAX_ROWS):
            want = {it[1] for it in prefetch_plan if it[0] == "stride" and it[2] == 2}
            for ts in want:
                d = 0
                while (ts << d) in want:
                    d += 1
                self._chain_depth[ts] = d
        try:
            if self.sorted:
                perm32 = order = None
                if coords.shape[0] > 1:
                    perm32, coords = ops.morton_order(coords, 1, ORDER_BLOCK_BITS, want_sorted=True, raw=True)
                index, ndup = ops.block_index_build(coords, 1, ORDER_BLOCK_BITS)
                level = _Level(coords, index=index)
                if MAP_ORDER and ndup == 0 and coords.shape[0] >= MAP_ORDER_MIN_ROWS:
                    if prefetch_plan:
                        self.levels[1] = level  # provisional: Morton order + index, enough to derive the coarser level
                        self.prefetch(prefetch_plan, early=True)
                    coords_p, order, phys_of, finish = _order_level(coords, index, 1)
                    level = _Level(coords_p, index=index, phys_of=phys_of, same_map=_with_compact(finish()))
                if perm32 is not None:  # internal row p = caller row perm[p]
                    self.perm, self.inv_perm = ops.compose_perm(perm32, order, coords.shape[0], coords.device)
            else:
                table, ndup = ops.hash_build(coords)
                level = _Level(coords, table=table)
            if ndup:
                raise ValueError("%d duplicate coordinates: the input must hold one row per (batch, x, y, z) "
                                 "(GridSampling3D guarantees it; ME's random sub-sampling of duplicates is not reproduced)" % ndup)
            if level.same_map is not None:
                self.maps[(1, 1, 3, 1)] = level.same_map
            self.levels[1] = level
            if self._worker is not None:
                self._built(("level", 1))  # the builder's stream reads the final input level (physical order, same-level map)
        finally:
            self._input_final.set()

    def _built(self, key):
        ev = torch.cuda.Event()
        ev.record(_current_stream())
        self._ready[key] = ev

    def _use(self, key):
        ev = self._ready.get(key)
        if ev is not None:
            if ev.query():  # complete: no barrier packet in front of the consumer (each costs ~10 us of an idle queue)
                self._ready[key] = None
            else:
                _current_stream().wait_event(ev)

    def prefetch(self, plan, early=False):
        """replay `plan` (the request log of an earlier forward of the same model) on the side stream.  early (from the
        constructor): the input level is still being ordered -- only the coarsening of the input level runs before it
        is final."""
        if self._worker is not None:
            return  # started by the constructor
        dev = self.orig_coords.device
        side = self._side if self._side is not None else _side_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))

        def work():
            try:
                with torch.cuda.device(dev), torch.cuda.stream(side), torch.no_grad():
                    if early:
                        first = next((it for it in plan if it[0] == "stride" and it[1] == 1), None)
                        if first is not None:
                            self.ensure_stride(first[1], first[2])
                        self._input_final.wait()
                    for item in plan:
                        if item[0] == "stride":
                            self.ensure_stride(item[1], item[2])
                        else:
                            self.kernel_map(*item[1])
            except BaseException as e:  # surfaced by join_prefetch
                self._worker_err = e

        self._side_built = True
        self._worker = threading.Thread(target=work, name="pp-map-prefetch")
        self._worker.start()

    def join_prefetch(self):
        if self._worker is not None:
            self._worker.join()
            self._worker = None
            dev = self.orig_coords.device
            torch.cuda.current_stream(dev).wait_stream(self._side if self._side is not None else _side_stream(dev))
            if self._worker_err is not None:
                err, self._worker_err = self._worker_err, None
                raise err

    def _consumed_here(self, *tensors):
        """a tensor built on the prefetch stream belongs to that stream's allocator pool: tell the allocator that the
        calling stream reads it too, so that the block is not handed out again while this stream's kernels are pending"""
        if not self._side_built:
            return
        raw = _raw_stream()
        for t in tensors:
            if t is not None and t.is_cuda and getattr(t, "pp_seen_by", None) != raw:
                t.record_stream(_current_stream())
                t.pp_seen_by = raw

    def _reads(self, *items):
        """A build on THIS stream is about to read `items` -- levels, block indices, maps, tensors -- that another stream's build
        may have allocated (the prefetch worker and the main thread both build whatever is asked of them first; each holds the
        lock, waits for the producing launch's event, and launches on ITS stream).  The event orders the data; it does not
        keep the MEMORY: a temporary of the build that the producer stream's allocator pool owns (the Morton-order map a
        level's pending same-level map is permuted from, a reverse map, an order, a block index) goes back to that pool when
        its last reference drops, and the producer stream may hand it out again while this stream's kernels still read it --
        a map full of another kernel's output, row translations gathered at garbage indices (the memory faults of the 4096-row
        window sweeps in rounds 4 and 6: a timing the default windows rarely reach).  Marks them all for the calling stream."""
        if not self._side_built or _NO_READS_MARK:
            return
        flat = []
        for it in items:
            if it is None:
                continue
            if torch.is_tensor(it):
                flat.append(it)
                flat.extend(t for t in (getattr(it, "pp_order", None), getattr(it, "pp_mask", None), getattr(it, "pp_pairs", None),
                                        getattr(it, "pp_dense", None)) if t is not None)
                flat.extend(_cmap_tensors(it))
            elif isinstance(it, _Level):
                flat.extend(t for t in (it.coords, it.phys_of, it.same_map) if t is not None)
                if it.index is not None:
                    flat.extend(getattr(it.index, a) for a in ops.BlockIndex.__slots__ if torch.is_tensor(getattr(it.index, a, None)))
                if it.table is not None:
                    flat.extend((it.table.keys, it.table.vals))
            elif isinstance(it, (tuple, list)):
                flat.extend(t for t in it if torch.is_tensor(t))
        self._consumed_here(*flat)

    def level(self, ts):
        self._use(("level", ts))
        lv = self.levels[ts]
        self._consumed_here(lv.coords)
        return lv

    def to_internal(self, feats):
        return feats if self.perm is None else _permute_rows(feats, self.perm, self.inv_perm)

    def to_caller(self, feats):
        return feats if self.inv_perm is None else _permute_rows(feats, self.inv_perm, self.perm)

    def ensure_stride(self, ts_in, stride):
        if self._log is not None:
            self._log.append(("stride", ts_in, stride))
        ts_out = ts_in * stride
        if ts_out not in self.levels:  # (lock-free when the level exists: the worker may hold the lock for a while)
            with self._lock:
                self._ensure_stride_locked(ts_in, stride)
        self._use(("level", ts_out))
        return ts_out

    def _ensure_stride_locked(self, ts_in, stride):
        ts_out = ts_in * stride
        if ts_out not in self.levels:
            self._use(("level", ts_in))
            src = self.levels[ts_in]
            self._reads(src)
            if self.sorted and stride == 2 and ORDER_BLOCK_BITS == 4 and src.index is not None and COARSEN_FROM_INDEX:
                # the coarse level is a bit permutation of the fine level's occupancy bitmaps: no hash, no sort
                built = self._chain.pop(ts_out, None)
                if built is not None:
                    self._reads(built[1], *[getattr(built[0], a) for a in ops.BlockIndex.__slots__
                                            if torch.is_tensor(getattr(built[0], a, None))])
                if built is None:
                    depth = self._chain_depth.get(ts_in, 0)
                    if depth >= 2 and not torch.is_grad_enabled():
                        # every level the plan will ask for from here on, in one call and one host read
                        chain = ops.block_index_coarsen_chain(src.index, src.n, depth)
                        for l, item in enumerate(chain):
                            self._chain[ts_in << (l + 1)] = item
                        built = self._chain.pop(ts_out)
                    else:
                        built = ops.block_index_coarsen(src.index, src.n)
                index, out = built
                level = _Level(out, index=index)
            else:
                src_coords = src.coords
                out, table, _ = ops.stride_coords(src_coords, ts_out)
                if self.sorted:
                    # first-appearance order of the parents follows the fine level only roughly: sort the level itself
                    if out.shape[0] > 1:
                        out = ops.morton_order(out, ts_out, ORDER_BLOCK_BITS, want_sorted=True)[1]
                    index, _ = ops.block_index_build(out, ts_out, ORDER_BLOCK_BITS)
                    level = _Level(out, index=index)
                else:
                    level = _Level(out, table=table)
            if MAP_ORDER and level.index is not None and level.n >= MAP_ORDER_MIN_ROWS:
                coords_p, _, phys_of, finish = _order_level(level.coords, level.index, ts_out)
                level = _Level(coords_p, index=level.index, phys_of=phys_of)
                self._pending_same[ts_out] = finish  # run by the first request for the level's same-level map
            self._built(("level", ts_out))  # event first: lock-free readers find the item only with its event
            self.levels[ts_out] = level
        return ts_out

    def kernel_map(self, ts_from, ts_to, ksize, sign):
        if ksize == 1:
            if ts_from != ts_to:
                raise NotImplementedError("1x1x1 convolutions with stride > 1 are not used by the reference path")
            return None
        key = (ts_from, ts_to, ksize, sign)
        if self._log is not None:
            self._log.append(("map", key))
        m = self.maps.get(key)
        if m is None:
            with self._lock:
                m = self._kernel_map_locked(key)
        self._use(key)
        self._consumed_here(m, getattr(m, "pp_order", None), *_cmap_tensors(m))
        if getattr(m, "pp_t8", False) and torch.is_grad_enabled():
            # The 8-wide form is an inference-only layout ([8, n], not [27, n]): the weight gradient and the pair lists index
            # 27 * n entries.  A map built under no_grad (frozen pre-pass, prefetch worker) and then used with grad enabled
            # is expanded once to its dense twin in the same slot order; autograd only ever sees that one.
            d = getattr(m, "pp_dense", None)
            if d is None:
                with self._lock:
                    d = getattr(m, "pp_dense", None)
                    if d is None:
                        d = ops.map8_to_dense(m)
                        d.pp_order = m.pp_order
                        m.pp_dense = d
            return d
        return m

    def kernel_map_rows(self, ts_from, ts_to, ksize, sign):
        """the same map indexed by PHYSICAL output rows (un-slotted copy of a cross-level map; same-level maps are
        returned as they are) -- for consumers outside the convolution kernels (tests, reference-style gather loops)"""
        m = self.kernel_map(ts_from, ts_to, ksize, sign)
        order = getattr(m, "pp_order", None)
        if m is None or order is None:
            return m
        if getattr(m, "pp_t8", False):  # 8-wide transposed map: expand, then un-slot
            m = ops.map8_to_dense(m)
        rows = torch.empty_like(m)
        rows[:, order.long()] = m
        return rows

    def _kernel_map_locked(self, key):
        ts_from, ts_to, ksize, sign = key
        m = self.maps.get(key)
        if m is None:
            self._use(("level", ts_from))
            self._use(("level", ts_to))
            if ts_to not in self.levels:
                raise ValueError("transposed convolution onto tensor stride %d: that coordinate map was never created" % ts_to)
            rev = self.maps.get((ts_to, ts_from, ksize, -sign))
            if rev is not None:
                self._use((ts_to, ts_from, ksize, -sign))
            dst = self.levels[ts_to]
            finish = self._pending_same.pop(ts_to, None) if (ts_from == ts_to and sign == 1 and ksize == 3) else None
            self._reads(dst, self.levels.get(ts_from), rev, getattr(finish, "reads", None))
            if finish is not None:
                m = dst.same_map = _with_compact(finish())
            elif rev is not None and ts_from == ts_to:
                m = torch.flip(rev, [0]).contiguous()  # mirrored offsets: offset_k -> offset_{K-1-k}
                if hasattr(rev, "pp_pairs"):
                    m.pp_pairs = rev.pp_pairs
            elif rev is not None:
                # transposed strided map by scatter from the strided one; rows = physical rows of the finer level
                ordered = MAP_ORDER and dst.n >= MAP_ORDER_MIN_ROWS
                if (ordered and MAP_T8 and not torch.is_grad_enabled() and ksize == 3 and ts_from == 2 * ts_to
                        and rev.shape[0] == 27 and rev.shape[1] < (1 << 28)):
                    m8, key8 = ops.kernel_map_transpose8(rev, dst.n, order=getattr(rev, "pp_order", None))
                    order = ops.map_order(key8)
                    m = ops.map_permute(m8, order)
                    m.pp_order = order
                    m.pp_t8 = True
                else:
                    m = ops.kernel_map_transpose(rev, dst.n, order=getattr(rev, "pp_order", None))
                    if ordered:
                        order = ops.map_order(ops.map_mask(m))
                        m = ops.map_permute(m, order)
                        m.pp_order = order
            else:
                src = self.levels[ts_from]
                if src.index is not None:
                    # transposed maps built by lookup are slot-ordered like the scattered ones; strided maps are not: their
                    # convolutions gain 8 % from it (152 vs 165 us at 1.3 M rows), less than the sort + permute cost
                    ordered = MAP_ORDER and ts_from > ts_to and dst.n >= MAP_ORDER_MIN_ROWS
                    # ... except "dust": a coarse level with nearly as many rows as its fine level (the proposal scorer's
                    # levels: 5.3 M -> 4.9 M -> 3.3 M rows) has 1.5 - 3.6 pairs per row, i.e. one or two of 27 offsets per row --
                    # unordered, a 16-row tile then walks ~15 offsets for ~20 pairs; ordered by mask its rows share theirs
                    if (MAP_ORDER and not ordered and ts_from < ts_to and dst.n >= MAP_ORDER_MIN_ROWS
                            and src.n < STRIDED_ORDER_RATIO * dst.n):
                        ordered = True
                    if ordered:
                        m = ops.kernel_map_bi(dst.coords, src.index, ksize, min(ts_from, ts_to), sign, want_mask=True)
                        order = ops.map_order(m.pp_mask)
                        del m.pp_mask
                        m = ops.map_permute(m, order, translate=src.phys_of)
                        m.pp_order = order
                    else:
                        m = ops.kernel_map_bi(dst.coords, src.index, ksize, min(ts_from, ts_to), sign, translate=src.phys_of)
                else:
                    m = ops.kernel_map(dst.coords, src.table, ksize, min(ts_from, ts_to), sign)
            self._built(key)
            self.maps[key] = m
        return m


# ------------------------------------------------------------------------------------------------
# sparse tensor
# ------------------------------------------------------------------------------------------------
class PreparedCoordinates:
    """The coordinate manager of a batch that is not being processed yet, built by a thread of its own on a stream of its own
    while the GPU works on the previous batch: Morton order, block index, the input level's slot order and same-level map
    and -- with `prefetch_plan` -- the coarser levels and their maps, i.e. everything between "the batch is in HBM" and "the
    first convolution can start" (3.5 ms of launches and three host reads at the start of a bench step, during which the
    main stream has nothing else to run).  `SparseTensor(..., prepared=p)` takes it over if `p.matches(coordinates)`:
    the consumer's stream waits for the build, and every tensor it reads was allocated from the build stream's pool, so
    it is marked as used by the consumer's stream like the prefetched maps are (_consumed_here).
    The caller must not modify `coords` between this call and the SparseTensor that uses it.
    Inference only (the autograd path keeps to one stream)."""

    def __init__(self, coords, prefetch_plan=None):
        if coords.dtype != torch.int32:
            coords = coords.to(torch.int32)
        self.coords = coords.contiguous()
        self._key = (self.coords.data_ptr(), tuple(self.coords.shape))
        self._cm = None
        self._err = None
        dev = self.coords.device
        prep = _prep_stream(dev)
        prep.wait_stream(torch.cuda.current_stream(dev))  # `coords` may have been written by the caller's stream
        self._done = torch.cuda.Event()

        def work():
            try:
                with torch.cuda.device(dev), torch.cuda.stream(prep), torch.no_grad():
                    # (its level / map builder gets a stream of its own too: the shared side stream is busy with the maps of
                    # the batch that is being processed, whose convolutions wait for them)
                    self._cm = CoordinateManager(self.coords, prefetch_plan=prefetch_plan, side_stream=_prep_stream(dev, 1))
                    self._done.record(prep)
            except BaseException as e:  # surfaced by take()
                self._err = e

        self._thread = threading.Thread(target=work, name="pp-input-prepare")
        self._thread.start()

    def matches(self, coordinates):
        return (coordinates.dtype == torch.int32 and coordinates.is_contiguous()
                and (coordinates.data_ptr(), tuple(coordinates.shape)) == self._key)

    def take(self):
        """-> the coordinate manager, usable on the calling thread's current stream"""
        self._thread.join()
        if self._err is not None:
            raise self._err
        cm, self._cm = self._cm, None
        cur = _current_stream()
        cur.wait_event(self._done)
        cm._side_built = True  # every level / map access goes through _consumed_here from now on
        lv = cm.levels[1]
        direct = [cm.orig_coords, cm.perm, cm.inv_perm, lv.coords, lv.phys_of, lv.same_map]
        if lv.index is not None:
            direct += [t for t in vars(lv.index).values() if torch.is_tensor(t)] if hasattr(lv.index, "__dict__") else \
                [getattr(lv.index, a) for a in getattr(lv.index, "__slots__", ()) if torch.is_tensor(getattr(lv.index, a, None))]
        direct += _cmap_tensors(lv.same_map)
        cm._consumed_here(*direct)
        return cm


class GatheredRows:
    """`base[index]` that has not been gathered yet.  Passed as the features of a SparseTensor it is resolved together
    with the internal row permutation in ONE gather (base[index[perm]]) -- the proposal scorer feeds millions of
    duplicated backbone rows this way (PointGroup3heads._compute_score)."""

    def __init__(self, base, index):
        if isinstance(base, GatheredRows):  # rows of rows: one index
            base, index = base.base, base.index[index]
        self.base, self.index = base, index
        self.shape = (index.shape[0], base.shape[1])
        self.device = base.device

    def to(self, *args, **kwargs):
        return self

    # (the few tensor operations consumers outside the hot path use: they gather first)
    def cpu(self):
        return self.materialise().cpu()

    def contiguous(self):
        return self.materialise()

    def __getitem__(self, idx):
        if torch.is_tensor(idx) and idx.dtype == torch.int64 and idx.dim() == 1:
            return GatheredRows(self, idx).materialise()  # rows of rows: still one gather
        return self.materialise()[idx]

    def materialise(self, perm=None):
        index = self.index if perm is None else self.index[perm]
        return _GatherRowsFn.apply(self.base, index) if self.base.requires_grad else ops.gather_rows(self.base, index)


class SparseTensor:
    """features + coordinate manager + tensor stride.  `feats` is the internal (Morton-ordered) feature matrix every
    kernel works on; `.F` / `.C` give the caller-visible view (row i of F belongs to row i of the coordinates passed
    in, applications/minkowski.py:193)."""

    def __init__(self, features, coordinates=None, device=None, coordinate_manager=None, tensor_stride=1, prefetch_plan=None,
                 prepared=None, **kwargs):
        if coordinate_manager is None:
            if coordinates is None:
                raise ValueError("SparseTensor needs coordinates or a coordinate_manager")
            dev = torch.device(device) if device is not None else features.device
            if dev.type != "cuda":
                raise ops._lib.PanopticHipError("SparseTensor must live on a HIP device (no CPU fallback)")
            if prepared is not None and prepared.matches(coordinates):
                coordinate_manager = prepared.take()  # built ahead on the preparation stream (PreparedCoordinates)
            else:
                coordinate_manager = CoordinateManager(coordinates.to(dev), prefetch_plan=prefetch_plan)
            if isinstance(features, GatheredRows):
                features = features.materialise(coordinate_manager.perm)
            else:
                features = coordinate_manager.to_internal(features.to(dev))
        self.feats = features
        self.coordinate_manager = coordinate_manager
        self.tensor_stride = int(tensor_stride)

    @property
    def F(self):
        if self.tensor_stride == 1:
            return self.coordinate_manager.to_caller(self.feats)
        return self.feats

    @property
    def C(self):
        if self.tensor_stride == 1:
            return self.coordinate_manager.orig_coords
        return self.coordinate_manager.level(self.tensor_stride).coords

    @property
    def device(self):
        return self.feats.device

    @property
    def shape(self):
        return self.feats.shape

    def _like(self, feats):
        return SparseTensor(feats, coordinate_manager=self.coordinate_manager, tensor_stride=self.tensor_stride)

    def __add__(self, other):
        if isinstance(other, SparseTensor):
            if other.coordinate_manager is not self.coordinate_manager or other.tensor_stride != self.tensor_stride:
                raise ValueError("SparseTensor + SparseTensor needs the same coordinate map")
            return self._like(self.feats + other.feats)
        return self._like(self.feats + other)

    def __repr__(self):
        return "SparseTensor(F=%s, tensor_stride=%d)" % (tuple(self.feats.shape), self.tensor_stride)


def cat(*tensors):
    """Channel concat on the same coordinate map (ME.cat, api_modules.py:308)."""
    a = tensors[0]
    for t in tensors[1:]:
        if t.coordinate_manager is not a.coordinate_manager or t.tensor_stride != a.tensor_stride:
            raise ValueError("ME.cat: tensors must share the coordinate map")
    return a._like(torch.cat([t.feats for t in tensors], dim=1))


# ------------------------------------------------------------------------------------------------
# autograd functions (training path); eval uses the fused entry points below
# ------------------------------------------------------------------------------------------------
# weight gradients over per-offset pair lists (pp_wgrad_pairs_build + pp_spconv_bwd_weight_pairs); PP_WGRAD_PAIRS=0: over the
# dense map (pp_spconv_bwd_weight)
WGRAD_PAIRS = os.environ.get("PP_WGRAD_PAIRS", "1") != "0"


def _pack(kernel, transpose=False, kflip=False):
    """packed weights of a layer: model parameters go through the per-version cache (one launch re-packs all of them after
    an optimizer step), anything else is packed on the spot"""
    if isinstance(kernel, nn.Parameter) and os.environ.get("PP_PACK_BATCHED", "1") != "0":
        return ops.pack_weight_cached(kernel, transpose=transpose, kflip=kflip)
    return ops.pack_weight(kernel, transpose=transpose, kflip=kflip)


class _SparseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, kernel, nbr, inv_fn, n_out, K, same_level=False, kflip=False):
        """same_level: nbr is the (+1) map of a stride-1 3x3x3 layer; its mirrored twin is nbr[K-1-k], so the transposed
        layer (kflip) and every input gradient reuse nbr with the offsets of the packed weights reversed.  Cross-level
        maps are slot-ordered (nbr.pp_order: slot -> output row)."""
        feats = feats.contiguous()
        packed = _pack(kernel, kflip=kflip)
        cout = kernel.shape[-1]
        bf16 = _CONV_BF16[0]
        out = ops.spconv_fwd(feats, packed, nbr, n_out, cout, K, row_order=getattr(nbr, "pp_order", None), bf16=bf16)
        ctx.save_for_backward(feats, kernel)
        ctx.nbr, ctx.inv_fn, ctx.K, ctx.bf16 = nbr, inv_fn, K, bf16
        ctx.same_level, ctx.kflip = same_level, kflip
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, kernel = ctx.saved_tensors
        din, dw = _conv_backward(ctx, feats, kernel, dout.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return din, dw, None, None, None, None, None, None


def _conv_backward(ctx, feats, kernel, dout, need_in, need_w):
    """input and weight gradient of a sparse convolution from the context _SparseConvFn / _ConvBnActTrainFn keep"""
    din = dw = None
    if getattr(ctx.nbr, "pp_t8", False):
        raise RuntimeError("an 8-wide transposed map reached autograd: CoordinateManager.kernel_map hands out the dense "
                           "twin when grad is enabled -- the map was fetched under no_grad and used with grad")
    if need_in:
        if ctx.same_level:
            packed_t = _pack(kernel, transpose=True, kflip=not ctx.kflip)
            inv = ctx.nbr
        else:
            packed_t = _pack(kernel, transpose=True)
            inv = ctx.inv_fn()
        ops.PROFILE_TAG = "dgrad"
        try:
            din = ops.spconv_fwd(dout, packed_t, inv, feats.shape[0], feats.shape[1], ctx.K,
                                 row_order=getattr(inv, "pp_order", None), bf16=ctx.bf16)
        finally:
            ops.PROFILE_TAG = "fwd"
    if need_w:
        order = getattr(ctx.nbr, "pp_order", None)
        if (WGRAD_PAIRS and ctx.nbr is not None and dout.shape[1] <= 192  # (the kernels' widest output tile set)
                and ctx.K * ctx.nbr.shape[1] < (1 << 31)):  # tile offsets are int32; larger maps take the dense-map kernel
            # the pairs of the map compacted per offset, once per map (every layer on this map re-uses the lists);
            # a slot order is folded into the lists, dout is read in its own row order
            wp = getattr(ctx.nbr, "pp_wpairs", None)
            if wp is None:
                wp = ctx.nbr.pp_wpairs = ops.wgrad_pairs(ctx.nbr, ctx.K, row_order=order)
            dw = ops.spconv_bwd_weight_pairs(feats, dout, wp, bf16=ctx.bf16)
        elif (ctx.nbr is None and ctx.K == 1 and WGRAD_PAIRS and ops.WGRAD_DETERMINISTIC and dout.shape[1] <= 192
              and feats.shape[0] < (1 << 31)):
            # 1x1 convolution (no map): the pair-list kernel over the identity map -- its block partials are added in a fixed
            # order, the dense-map kernel below adds with float atomics (the last gradients that differed run to run)
            dw = ops.spconv_bwd_weight_pairs(feats, dout, _identity_pairs(feats.shape[0], feats.device), bf16=ctx.bf16)
        else:
            # dW[k] = sum_s in[nbr[k][s]]^T dout[order[s]]: a slot-ordered map wants the output gradient in slot order
            dout_s = dout if order is None else ops.gather_rows(dout, order.long())
            dw = ops.spconv_bwd_weight(feats, dout_s, ctx.nbr, ctx.K, bf16=ctx.bf16)
        if ctx.kflip:
            dw = dw.flip(0)
        dw = dw.reshape(kernel.shape)
    return din, dw


class _BatchNormTrainFn(torch.autograd.Function):
    """y = act((x - mean) * rstd * w + b) with batch statistics (biased variance); pp_bn_train_fwd / pp_bn_train_bwd:
    three launches each way, running statistics updated inside the forward's finalize kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu, momentum, running):
        x = x.contiguous()
        rm, rv, nbt = running if running is not None else (None, None, None)
        ctx.sync = ops.sync_bn_active()  # (the backward all-reduces exactly when the forward did)
        y, mean, rstd = ops.bn_train_fwd(x, weight, bias, eps, momentum, rm, rv, relu, nbt, use_sync=ctx.sync)
        ctx.save_for_backward(x, weight, mean, rstd, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd, y = ctx.saved_tensors
        dx, dweight, dbias = ops.bn_train_bwd(x, dy.contiguous(), y, weight, mean, rstd, use_sync=ctx.sync)
        return dx, dweight, dbias, None, None, None, None


class _ConvBnActTrainFn(torch.autograd.Function):
    """training-mode  act(BN(conv(x)))  as ONE autograd node: _SparseConvFn followed by _BatchNormTrainFn, same launches in
    the same order (results are bit-identical to the two-node form), one Python round trip per layer and direction
    instead of two -- the training step is paced by the host."""

    @staticmethod
    def forward(ctx, feats, kernel, bn_w, bn_b, nbr, inv_fn, n_out, K, same_level, kflip, eps, relu, momentum, running):
        feats = feats.contiguous()
        bf16 = _CONV_BF16[0]
        h = ops.spconv_fwd(feats, _pack(kernel, kflip=kflip), nbr, n_out, kernel.shape[-1], K,
                           row_order=getattr(nbr, "pp_order", None), bf16=bf16)
        rm, rv, nbt = running if running is not None else (None, None, None)
        ctx.sync = ops.sync_bn_active()
        y, mean, rstd = ops.bn_train_fwd(h, bn_w, bn_b, eps, momentum, rm, rv, relu, nbt, use_sync=ctx.sync)
        ctx.save_for_backward(feats, kernel, h, bn_w, mean, rstd, y if relu else None)
        ctx.nbr, ctx.inv_fn, ctx.K, ctx.bf16 = nbr, inv_fn, K, bf16
        ctx.same_level, ctx.kflip = same_level, kflip
        return y

    @staticmethod
    def backward(ctx, dy):
        feats, kernel, h, bn_w, mean, rstd, y = ctx.saved_tensors
        dh, dbw, dbb = ops.bn_train_bwd(h, dy.contiguous(), y, bn_w, mean, rstd, use_sync=ctx.sync)
        din, dw = _conv_backward(ctx, feats, kernel, dh, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return (din, dw, dbw if bn_w is not None else None, dbb if ctx.needs_input_grad[3] else None,
                None, None, None, None, None, None, None, None, None, None)


FUSE_TRAIN = os.environ.get("PP_FUSE_TRAIN", "1") != "0"


def _one_row_check(n, c):
    """batch statistics of a single row: torch's BatchNorm1d (what ME.MinkowskiBatchNorm wraps) refuses them in training.
    Under SyncBN the statistics are those of all ranks' rows and a rank must never leave the collective schedule on its own (a
    raise here would leave its peers blocked in the all-reduce): no local check, like torch's SyncBatchNorm with world size > 1."""
    if n == 1 and not ops.sync_bn_active():
        raise ValueError("Expected more than 1 value per channel when training, got input size torch.Size([1, %d])" % c)


def conv_bn_act_train(x, conv, bn, relu=True):
    """Training-mode relu?(BN(conv(x))) through _ConvBnActTrainFn; None when the pair is not the plain case (convolution
    bias, BatchNorm in eval mode or without running statistics bookkeeping to mirror): the caller then runs the modules."""
    b = bn.bn
    if not FUSE_TRAIN or conv.bias is not None or not b.training or not torch.is_grad_enabled():
        return None
    ts_out, nbr, inv_fn = conv.out_stride_and_map(x)
    cm = x.coordinate_manager
    running, mom = None, 0.0
    if b.track_running_stats:
        mom = b.momentum if b.momentum is not None else 1.0 / float(b.num_batches_tracked + 1)
        running = (b.running_mean, b.running_var, b.num_batches_tracked)
        bn._folded = None
    same_level = conv.stride == 1 and conv.kernel_volume > 1
    _one_row_check(cm.level(ts_out).n, b.num_features)
    feats = _ConvBnActTrainFn.apply(x.feats, conv.kernel, b.weight, b.bias, nbr, inv_fn, cm.level(ts_out).n, conv.kernel_volume,
                                    same_level, conv.mirrored, b.eps, relu, mom, running)
    return SparseTensor(feats, coordinate_manager=cm, tensor_stride=ts_out)


class _AffineFn(torch.autograd.Function):
    """y = act(x*scale + shift) with constant scale/shift (eval-mode BN, ReLU)."""

    @staticmethod
    def forward(ctx, x, scale, shift, act):
        y = ops.affine_act(x.contiguous(), scale, shift, act=act)
        ctx.save_for_backward(scale, y if act == 1 else None)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        scale, y = ctx.saved_tensors
        if ctx.act == 1:
            dy = torch.ops.aten.threshold_backward(dy, y, 0)  # dy where y > 0 else 0, one launch
        if scale is not None:
            dy = ops.affine_act(dy.contiguous(), scale, None)
        return dy, None, None, None


# ------------------------------------------------------------------------------------------------
# modules
# ------------------------------------------------------------------------------------------------
class RegionType(enum.Enum):
    """kernel shapes ME knows; the reference's modules/MinkowskiEngine/common.py:53-62 builds lookup tables from these
    at import time.  Only HYPER_CUBE kernels (what every panoptic config uses) are implemented."""
    HYPER_CUBE = 0
    HYPER_CROSS = 1
    CUSTOM = 2


class KernelGenerator:
    """ME.KernelGenerator as the reference's conv helpers (common.py:117-190) construct it: a record of the kernel
    geometry, accepted by the convolution constructors through `kernel_generator=`."""

    def __init__(self, kernel_size=-1, stride=1, dilation=1, region_type=RegionType.HYPER_CUBE, axis_types=None,
                 dimension=-1, **kwargs):
        self.kernel_size, self.kernel_stride, self.kernel_dilation = kernel_size, stride, dilation
        self.region_type, self.axis_types, self.dimension = region_type, axis_types, dimension


class MinkowskiNetwork(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.D = D


def _to_int(v, name):
    if isinstance(v, (list, tuple)):
        if len(set(int(x) for x in v)) != 1:
            raise NotImplementedError("anisotropic %s is not used by the reference path" % name)
        v = v[0]
    return int(v)


class _ConvBase(nn.Module):
    TRANSPOSED = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, kernel_generator=None,
                 dimension=-1, **kwargs):
        super().__init__()
        if dimension not in (3, -1):
            raise NotImplementedError("only 3-D sparse convolutions are implemented")
        if kernel_generator is not None:
            if kernel_generator.region_type is not RegionType.HYPER_CUBE:
                raise NotImplementedError("only HYPER_CUBE kernels are implemented (what the panoptic configs use)")
            if kernel_size == -1:
                kernel_size = kernel_generator.kernel_size
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _to_int(kernel_size, "kernel_size")
        self.stride = _to_int(stride, "stride")
        self.dilation = _to_int(dilation, "dilation")
        if self.kernel_size not in (1, 3):
            raise NotImplementedError("kernel_size must be 1 or 3 (what the panoptic configs use)")
        if self.dilation != 1:
            raise NotImplementedError("dilation != 1 is not used by the reference path")
        self.kernel_volume = self.kernel_size ** 3
        shape = (self.kernel_volume, in_channels, out_channels) if self.kernel_volume > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None
        self.reset_parameters()
        self._packed = None

    def reset_parameters(self):
        # ME default: uniform(-stdv, stdv), stdv = 1/sqrt(in_channels * kernel_volume)
        stdv = 1.0 / math.sqrt(self.in_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    @property
    def mirrored(self):
        """transposed 3x3x3 convolution with stride 1: same-level map with mirrored offsets = the forward map with the
        offsets of the weights reversed (no flipped copy of the map)"""
        return self.TRANSPOSED and self.stride == 1 and self.kernel_volume > 1

    def packed(self):
        k = self.kernel
        tag = (k._version, ops.param_epoch(), k.data_ptr(), k.device)
        if self._packed is None or self._packed[0] != tag:
            self._packed = (tag, ops.pack_weight(k, kflip=self.mirrored))
        return self._packed[1]

    def out_stride_and_map(self, x):
        """-> (ts_out, nbr forward map, callable giving the input-gradient map)"""
        cm = x.coordinate_manager
        ts_in = x.tensor_stride
        sign = -1 if self.TRANSPOSED else 1
        if self.stride == 1:
            ts_out = ts_in
        elif self.TRANSPOSED:
            if ts_in % self.stride:
                raise ValueError("transposed convolution: tensor stride %d not divisible by stride %d" % (ts_in, self.stride))
            ts_out = ts_in // self.stride
        else:
            ts_out = cm.ensure_stride(ts_in, self.stride)
        ks = self.kernel_size
        if self.mirrored:
            sign = 1  # with packed(kflip) / _SparseConvFn(kflip=True)
        nbr = cm.kernel_map(ts_in, ts_out, ks, sign)
        return ts_out, nbr, (lambda: cm.kernel_map(ts_out, ts_in, ks, -sign))

    def forward(self, x):
        ts_out, nbr, inv_fn = self.out_stride_and_map(x)
        cm = x.coordinate_manager
        n_out = cm.level(ts_out).n
        if not torch.is_grad_enabled():  # inference: cached packed weights, no autograd bookkeeping
            feats = ops.spconv_fwd(x.feats.contiguous(), self.packed(), nbr, n_out, self.out_channels, self.kernel_volume,
                                   row_order=getattr(nbr, "pp_order", None), bf16=_CONV_BF16[0])
            if self.bias is not None:
                feats = feats + self.bias
            return SparseTensor(feats, coordinate_manager=cm, tensor_stride=ts_out)
        same_level = self.stride == 1 and self.kernel_volume > 1
        feats = _SparseConvFn.apply(x.feats, self.kernel, nbr, inv_fn, n_out, self.kernel_volume, same_level, self.mirrored)
        if self.bias is not None:
            feats = feats + self.bias
        return SparseTensor(feats, coordinate_manager=x.coordinate_manager, tensor_stride=ts_out)

    def extra_repr(self):
        return "in=%d, out=%d, kernel_size=%d, stride=%d" % (self.in_channels, self.out_channels, self.kernel_size, self.stride)


class MinkowskiConvolution(_ConvBase):
    TRANSPOSED = False


class MinkowskiConvolutionTranspose(_ConvBase):
    TRANSPOSED = True


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)
        self._folded = None
        self._fold_fast = None

    # bn_schedulers.py:6-31 sets `.momentum` on the module
    @property
    def momentum(self):
        return self.bn.momentum

    @momentum.setter
    def momentum(self, v):
        self.bn.momentum = v

    def folded(self):
        """(scale, shift) of the eval-mode affine map, cached until a parameter/buffer changes."""
        if self._fold_fast is not None and not torch.is_grad_enabled():
            return self._fold_fast  # inference fast path: validated once per eval()/load_state_dict (see train())
        bn = self.bn
        p, b = bn._parameters, bn._buffers  # direct dict access: nn.Module.__getattr__ costs ~1 us per attribute
        w, bias, mean, var = p["weight"], p["bias"], b["running_mean"], b["running_var"]
        tag = (w._version, bias._version, mean._version, var._version, ops.param_epoch(), w.data_ptr())
        if self._folded is None or self._folded[0] != tag:
            with torch.no_grad():
                scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                shift = bn.bias - bn.running_mean * scale
            self._folded = (tag, scale.contiguous(), shift.contiguous())
        if not self.training:
            self._fold_fast = (self._folded[1], self._folded[2])
        return self._folded[1], self._folded[2]

    def train(self, mode=True):
        self._fold_fast = None  # re-validate the folded affine after any switch of mode (weights may have been trained)
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._fold_fast = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .float(): parameters are replaced
        self._fold_fast = None
        self._folded = None
        return super()._apply(fn, *args, **kwargs)

    def features_forward(self, feats, relu=False):
        bn = self.bn
        if self.training or not bn.track_running_stats:
            running = None
            mom = 0.0
            if bn.track_running_stats:
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                # updated in place by the finalize kernel (the counter too: one launch less per layer)
                running = (bn.running_mean, bn.running_var, bn.num_batches_tracked)
                self._folded = None  # ... which does not bump the tensors' version counters
            if self.training:
                _one_row_check(feats.shape[0], bn.num_features)
            return _BatchNormTrainFn.apply(feats, bn.weight, bn.bias, bn.eps, relu, mom, running)
        scale, shift = self.folded()
        return _AffineFn.apply(feats, scale, shift, 1 if relu else 0)

    def forward(self, x):
        return x._like(self.features_forward(x.feats))

    def __repr__(self):
        return "MinkowskiBatchNorm(%s)" % self.bn.extra_repr()


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return x._like(_AffineFn.apply(x.feats, None, None, 1))


class MinkowskiSigmoid(nn.Module):
    def forward(self, x):
        return x._like(torch.sigmoid(x.feats))


class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x):
        return x._like(self.linear(x.feats))


class MinkowskiGlobalPooling(nn.Module):
    """Average of the features of each batch element (one row per batch element, tensor stride kept)."""

    def forward(self, x):
        b = x.coordinate_manager.level(x.tensor_stride).coords[:, 0].long()
        nb = int(b.max().item()) + 1 if b.numel() else 0
        if torch.is_grad_enabled() and x.feats.requires_grad:
            from ..torch_scatter import scatter  # differentiable segment mean (the raw op below is not)
            pooled = scatter(x.feats, b, dim=0, reduce="mean", dim_size=nb)
        else:
            pooled = ops.segment_reduce(x.feats.contiguous(), b, nb, "mean")
        out = SparseTensor.__new__(SparseTensor)
        out.feats, out.coordinate_manager, out.tensor_stride = pooled, x.coordinate_manager, -1
        return out


class MinkowskiBroadcastMultiplication(nn.Module):
    def forward(self, x, pooled):
        b = x.coordinate_manager.level(x.tensor_stride).coords[:, 0].long()
        return x._like(x.feats * pooled.feats[b])


# ------------------------------------------------------------------------------------------------
# fused eval entry point used by the build-owned ResBlock / ResNetDown / ResNetUp
# ------------------------------------------------------------------------------------------------
def conv_bn_act(x, conv, bn, relu=True, residual=None, skip=None, shortcut=None):
    """Eval-mode  relu?(BN(conv(cat(x, skip)))) + residual  as ONE kernel launch (folded BN in the epilogue,
    ME.cat fused as a second source).  x/skip/residual are SparseTensors on compatible maps.
    shortcut = (SparseTensor s, 1x1 conv, bn): `+ BN(conv1x1(s))` computed by the same launch (the downsample branch of a
    residual block); returns None when the library does not serve the shape that way (nothing launched)."""
    ts_out, nbr, _ = conv.out_stride_and_map(x)
    cm = x.coordinate_manager
    n_out = cm.level(ts_out).n
    scale = shift = None
    if bn is not None:
        scale, shift = bn.folded()
    in1 = None
    if skip is not None:
        if skip.coordinate_manager is not cm or skip.tensor_stride != x.tensor_stride:
            raise ValueError("fused cat: tensors must share the coordinate map")
        in1 = skip.feats
    res = None
    if residual is not None:
        if residual.tensor_stride != ts_out:
            raise ValueError("residual on a different tensor stride")
        res = residual.feats
    c0 = x.feats.shape[1]
    c1 = 0 if in1 is None else in1.shape[1]
    sc = None
    if shortcut is not None:
        s_t, s_conv, s_bn = shortcut
        if getattr(nbr, "pp_order", None) is not None or s_t.tensor_stride != ts_out or s_conv.kernel_volume != 1:
            return None
        s_scale, s_shift = s_bn.folded()
        sc = (s_t.feats, s_conv.packed(), s_scale, s_shift)
    feats = ops.spconv_fwd(x.feats, conv.packed(), nbr, n_out, conv.out_channels, conv.kernel_volume, in1=in1,
                           scale=scale, shift=shift, relu=relu, residual=res, row_order=getattr(nbr, "pp_order", None),
                           bf16=_CONV_BF16[0], shortcut=sc)
    if feats is None:
        return None
    if conv.bias is not None:
        raise NotImplementedError("fused path assumes bias=False (every conv of the reference network)")
    return SparseTensor(feats, coordinate_manager=cm, tensor_stride=ts_out)
