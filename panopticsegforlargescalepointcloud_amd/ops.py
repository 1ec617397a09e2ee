"""Torch-tensor front-end of the C-ABI (device pointers + the current HIP stream).

PyTorch is plumbing here (allocation, streams); every op below is one or a few launches of the hand-written
gfx950 kernels in csrc/.  All tensors must live on a HIP device; there is no CPU path.
"""
import ctypes as C
import os
import threading
import weakref

import torch

from . import _lib

_REDUCE = {"sum": 0, "add": 0, "mean": 1, "max": 2}


class _TimingEvent:
    """HIP event for timing only, created with hipEventDisableSystemFence: recording a default event makes the queue write
    back and invalidate its caches at system scope (what torch.cuda.Event does; ~6.6 us per record on the launch stream,
    1.2 ms per bench step for the two records around each of its 91 convolutions); a timing event does not need that."""
    _FLAGS = 0x20000000  # hipEventDisableSystemFence (hip_runtime_api.h)

    def __init__(self):
        rt = _lib.load()  # (the HIP runtime's symbols resolve through the library that links it)
        self.h = C.c_void_p()
        rt.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        rt.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        rt.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        rt.hipEventDestroy.argtypes = [C.c_void_p]
        if rt.hipEventCreateWithFlags(C.byref(self.h), self._FLAGS) != 0:
            raise _lib.PanopticHipError("hipEventCreateWithFlags failed")
        self._rt = rt

    def record(self):
        if self._rt.hipEventRecord(self.h, _stream()) != 0:
            raise _lib.PanopticHipError("hipEventRecord failed")

    def elapsed_time(self, end):
        ms = C.c_float()
        rc = self._rt.hipEventElapsedTime(C.byref(ms), self.h, end.h)
        if rc != 0:
            raise _lib.PanopticHipError("hipEventElapsedTime failed (%d)" % rc)
        return float(ms.value)

    def __del__(self):
        try:
            self._rt.hipEventDestroy(self.h)
        except Exception:
            pass


class LaunchProfiler:
    """Optional per-launch timing of the dominant kernel (pp_spconv_fwd) with HIP events recorded on the stream the
    kernel is launched on (bench.py uses it for the roofline object; off by default)."""

    def __init__(self, reserve=0):
        self.records = []  # (start, end, n_in, n_out, cin, cout, K, nbr tensor or None, has_residual, fused shortcut channels)
        self.map_k = []    # per record: rows of the kernel map the launch streams (27, or 8 for an 8-wide transposed map)
        self.fams = []     # per record: kernel family the launch ran on ("x3" / "x3f" / "fwd3"), asked from the library at launch time
        self.tags = []     # per record: TAG at launch time ("fwd" / "dgrad": the input gradient runs on the forward kernel)
        self.records_w = []  # weight-gradient launches (pp_spconv_bwd_weight): (start, end, n_in, n_out, cin, cout, K, pairs)
        # creating a timing event costs ~12 us of host time, recording one ~3 us: the pairs a run needs are created up front
        # (bench.py: launches per step x steps, counted during the warm-up) so that a timed launch only pays the records
        self._pool = [_TimingEvent() for _ in range(2 * int(reserve))]

    def events(self):
        pool = self._pool
        if len(pool) >= 2:
            return pool.pop(), pool.pop()
        return _TimingEvent(), _TimingEvent()

    def _pair_counts(self):
        """number of map pairs of every record, with ONE host read for all the device counters"""
        dev = [r[7] for r in self.records if r[7] is not None]
        vals = iter(torch.stack([d.reshape(()).to(torch.int64) for d in dev]).tolist()) if dev else iter(())
        return [r[3] if r[7] is None else int(next(vals)) for r in self.records]

    def summarize_wgrad(self, elem_bytes=4.0):
        """time, algorithmic bytes and flops of the weight-gradient launches: dW[k] = sum over pairs of in^T dout reads both
        feature matrices once and the map once, writes K x cin x cout floats; 2 flops per pair and channel pair"""
        torch.cuda.synchronize()
        dev = [r[7] for r in self.records_w if r[7] is not None]
        vals = iter(torch.stack([d.reshape(()).to(torch.int64) for d in dev]).tolist()) if dev else iter(())
        ms = b = fl = 0.0
        for (e0, e1, n_in, n_out, cin, cout, K, pairs) in self.records_w:
            P = n_out if pairs is None else int(next(vals))
            ms += e0.elapsed_time(e1)
            b += elem_bytes * (n_in * cin + n_out * cout) + 4.0 * K * cin * cout + 8.0 * P
            fl += 2.0 * P * cin * cout
        return {"launches": len(self.records_w), "ms": ms, "bytes": b, "flops": fl}

    @staticmethod
    def kernel_family(cin, cout, K, n_in=1 << 20, n_out=1 << 20, c1=0, shortcut=False):
        """which kernel the library runs a convolution of this shape on: "x3" = the split-operand kernel, "fwd3" = the fp32-MFMA
        kernels.  Asks the library (pp_spconv_kernel_family: the dispatch's own rule, its cached environment overrides included)
        instead of mirroring it.  cin = channels of both sources together, c1 = those of the second (ME.cat fused)."""
        fam = _lib.load().pp_spconv_kernel_family(int(cin) - int(c1), int(c1), int(n_in), int(K), int(n_out), int(cout), int(bool(shortcut)))
        return {1: "x3", 2: "x3f"}.get(fam, "fwd3")  # ("x3f": the split-operand arithmetic with full-line gathers through LDS)

    def summarize(self, tag=None):
        torch.cuda.synchronize()
        tot_ms = tot_bytes = tot_flops = map_bytes = 0.0
        counts = self._pair_counts()
        n_used = 0
        by_family = {}
        tags = self.tags if len(self.tags) == len(self.records) else [None] * len(self.records)
        map_k = self.map_k if len(self.map_k) == len(self.records) else [None] * len(self.records)
        fams = self.fams if len(self.fams) == len(self.records) else [None] * len(self.records)
        for (e0, e1, n_in, n_out, cin, cout, K, pairs, has_res, ds_c), P, tg, mk, fam in zip(self.records, counts, tags, map_k, fams):
            if tag is not None and tg != tag:
                continue
            n_used += 1
            mb = 0.0
            if mk == -1:
                mb += 4.125 * n_out + 6.0 * P - 4.0 * K * n_out  # compact map: masks, chunk offsets, 6 bytes per present entry
            elif mk is not None and K > 1:
                mb += 4.0 * (mk - K) * n_out  # (corrects the dense estimate below for 8-wide maps)
            mb += 4.0 * K * n_out if K > 1 else 0.0  # what the kernel actually streams: the dense [K, n_out] map
            map_bytes += mb
            # features read once + written once, weights once, one (in,out) int32 pair per map entry;
            # a fused 1x1 shortcut (ds_c input channels) adds its input rows, its weights and its flops
            b = 4.0 * (n_in * cin + n_out * cout) + 4.0 * K * cin * cout + 8.0 * P + (4.0 * n_out * cout if has_res else 0.0)
            b += 4.0 * n_out * ds_c + 4.0 * ds_c * cout
            tot_bytes += b
            fl = 2.0 * P * cin * cout + 2.0 * n_out * ds_c * cout
            tot_flops += fl
            ms = e0.elapsed_time(e1)
            tot_ms += ms
            if fam is None:
                fam = self.kernel_family(cin, cout, K, n_in, n_out, 0, ds_c > 0)
            f = by_family.setdefault(fam, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0, "map_bytes": 0.0})
            f["launches"] += 1
            f["ms"] += ms
            f["bytes"] += b
            f["flops"] += fl
            f["map_bytes"] += mb
        return {"launches": n_used, "ms": tot_ms, "bytes": tot_bytes, "flops": tot_flops, "map_bytes": map_bytes, "by_family": by_family}

    def table(self, steps=1, hbm_peak=8.0e12, mfma_peak=157.3e12, bf16_peak=2.5e15):
        """Markdown table: launches grouped by shape, per step -- time, algorithmic GB and GFLOP, fraction of the HBM roof, of the fp32
        MFMA peak (algorithmic flops; the split-operand kernel can exceed 1 there) and of the pipe the launch RUNS on (x3: six bf16
        products per fp32 product against the dense bf16 peak)."""
        torch.cuda.synchronize()
        groups = {}
        fams = self.fams if len(self.fams) == len(self.records) else [None] * len(self.records)
        for (e0, e1, n_in, n_out, cin, cout, K, pairs, has_res, ds_c), P, fam in zip(self.records, self._pair_counts(), fams):
            b = 4.0 * (n_in * cin + n_out * cout) + 4.0 * K * cin * cout + 8.0 * P + (4.0 * n_out * cout if has_res else 0.0)
            b += 4.0 * n_out * ds_c + 4.0 * ds_c * cout
            if fam is None:
                fam = self.kernel_family(cin, cout, K, n_in, n_out, 0, ds_c > 0)
            g = groups.setdefault((n_in, n_out, "%d+%d" % (cin, ds_c) if ds_c else cin, cout, K, P, fam), [0, 0.0, 0.0, 0.0])
            g[0] += 1
            g[1] += e0.elapsed_time(e1)
            g[2] += b
            g[3] += 2.0 * P * cin * cout + 2.0 * n_out * ds_c * cout
        lines = ["| rows in | rows out | Cin | Cout | K | pairs/row | kernel | launches/step | ms/step | us/launch | alg GB/s | alg TFLOP/s | "
                 "frac HBM | frac fp32 MFMA | frac pipe |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
        for (n_in, n_out, cin, cout, K, P, fam), (cnt, ms, b, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
            sec = ms / 1e3
            pipe = 6.0 * fl / bf16_peak if fam in ("x3", "x3f") else fl / mfma_peak
            lines.append("| %d | %d | %s | %d | %d | %.2f | %s | %.1f | %.2f | %.0f | %.0f | %.1f | %.3f | %.3f | %.3f |" % (
                n_in, n_out, cin, cout, K, P / max(n_out, 1), fam, cnt / steps, ms / steps, ms / cnt * 1e3, b / sec / 1e9,
                fl / sec / 1e12, b / sec / hbm_peak, fl / sec / mfma_peak, pipe / sec))
        return "\n".join(lines) + "\n"


PROFILER = None
PROFILE_TAG = "fwd"  # what the next pp_spconv_fwd launches are (the autograd backward sets "dgrad" around its launch)



class _CounterPool(threading.local):
    """Small zero-initialised device counters (pair counts, overflow flags, sizes ...) handed out as slices of one zeroed block
    per thread -- every `torch.zeros(2)` was a fill launch of its own (~80 per bench step on the two builder threads and the
    main thread; the small configurations are paced by their dispatches).  A block is never reused: when it is exhausted a
    fresh one is allocated, so a counter stays valid for as long as its tensor lives.  One block per thread AND stream (the
    main thread also launches on a side stream, e.g. the overlap-pair pass): the fill of a block is ordered on the stream whose
    kernels use its counters, and the block comes from that stream's allocator pool."""

    def __init__(self):
        self.blocks = {}

    def take(self, n, dtype, device):
        key = (dtype, device, _stream().value or 0)
        blk = self.blocks.get(key)
        if blk is None or blk[1] + n > blk[0].numel():
            blk = self.blocks[key] = [torch.zeros(2048, dtype=dtype, device=device), 0]
        out = blk[0][blk[1]: blk[1] + n]
        blk[1] += (n + 3) & ~3  # 16-byte granules: neighbours never share a 64-bit word that a kernel updates atomically
        return out


_COUNTERS = _CounterPool()


def _zeros(n, dtype, device):
    return _COUNTERS.take(n, dtype, device)


def _pairs_of(nbr):
    """device scalar holding the number of pairs of a kernel map (None = identity map); never keeps the map alive"""
    if nbr is None:
        return None
    p = getattr(nbr, "pp_pairs", None)
    return p if p is not None else (nbr >= 0).sum()


_PAIR_COUNTERS = []  # (weakref to a device pair counter, raw stream it was written on): counters nobody has read yet
_PAIR_LOCK = threading.Lock()  # (builder threads register counters while the main thread reads them)


def _register_pairs(p):
    with _PAIR_LOCK:
        if len(_PAIR_COUNTERS) >= 512:  # (inference never reads them: drop the dead ones now and then)
            _PAIR_COUNTERS[:] = [(r, st) for r, st in _PAIR_COUNTERS if r() is not None][-256:]
        _PAIR_COUNTERS.append((weakref.ref(p), _stream().value or 0))
    return p


def _pairs_host(p):
    """int value of a device pair counter.  The first request reads ALL counters written on the calling stream that are still
    unread in one transfer: the training step asked for them one map at a time from inside the backward pass (19 synchronisations
    per step, each with the GPU running dry behind it)."""
    v = getattr(p, "pp_host", None)
    if v is not None:
        return v
    cur = _stream().value or 0
    batch, keep = [p], []
    with _PAIR_LOCK:
        for ref, st in _PAIR_COUNTERS:
            t = ref()
            if t is None or t is p or getattr(t, "pp_host", None) is not None:
                continue
            # (only counters written on the calling stream: nothing orders this read behind another stream's kernels)
            if st == cur and t.device == p.device:
                batch.append(t)
            else:
                keep.append((ref, st))
        _PAIR_COUNTERS[:] = keep
    vals = torch.stack([t.reshape(()).to(torch.int64) for t in batch]).tolist()
    for t, val in zip(batch, vals):
        t.pp_host = int(val)
    return p.pp_host


def _stream():
    # raw handle of torch's current HIP stream; the C-level getters skip ~25 us of Python per call
    # (torch.cuda.current_stream() re-runs lazy-init checks and builds a Stream object every time)
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:  # private API moved: fall back to the public one
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need(t, dtype, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.PanopticHipError("%s must be a HIP device tensor (there is no CPU fallback)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


_WS_CACHE = {}


def _ws(nbytes, device, tag=None):
    """Scratch buffer.  Large tagged workspaces are kept (grow-only) per device AND stream: re-allocating multi-GB blocks
    every call makes the caching allocator split / re-hipMalloc them (measured: +270 ms per region_grow call); the stream
    is part of the key because worker threads launch the same tagged ops on side streams (map prefetch, mean shift next to
    region growing) and nothing orders two streams' use of one scratch block."""
    nbytes = max(int(nbytes), 256)
    if tag is None or nbytes < (64 << 20):
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    key = (tag, str(device), _stream().value)
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < nbytes:
        with torch.cuda.device(device):
            _WS_CACHE[key] = buf = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=device)
    return buf


def version():
    return _lib.load().pp_version().decode()


def spconv_x3_full_lines(mode=-1):
    """Switch the split-operand kernel's full-line row gathers (k_spconv_x3f) on (1) / off (0), or only ask (-1); returns the
    setting before the call.  Results are bit-identical either way (tests/test_hip_ops.py)."""
    return int(_lib.load().pp_spconv_x3_full_lines(int(mode)))


def triad(a, b, c, s):
    lib = _lib.load()
    _lib.check(lib.pp_triad(_ptr(a), _ptr(b), _ptr(c), s, a.numel(), _stream()), "pp_triad")


# ------------------------------------------------------------------------------------------ coordinates
class HashTable:
    """Open-addressing coordinate hash living in HBM (keys uint64 as int64 storage, vals int32)."""

    def __init__(self, n, device):
        lib = _lib.load()
        self.cap = int(lib.pp_hash_capacity(int(n)))
        self.keys = torch.empty(self.cap, dtype=torch.int64, device=device)
        self.vals = torch.empty(self.cap, dtype=torch.int32, device=device)


def hash_build(coords):
    """coords int32 [n,4] -> (HashTable, n_duplicates, n_out_of_range)."""
    lib = _lib.load()
    coords = _need(coords, torch.int32, "coords")
    n = coords.shape[0]
    table = HashTable(n, coords.device)
    info = _zeros(2, torch.int32, coords.device)
    _lib.check(lib.pp_hash_build(_ptr(coords), n, _ptr(table.keys), _ptr(table.vals), table.cap, _ptr(info), _stream()),
               "pp_hash_build")
    ndup, nrange = info.tolist()
    if nrange:
        raise _lib.PanopticHipError("%d coordinates outside the packable range (batch < 65536, |xyz| < 32768)" % nrange)
    return table, ndup


def stride_coords(coords, ts_out):
    """-> (out_coords [n_out,4] int32 in first-appearance order, HashTable of out_coords, fine_to_coarse [n])."""
    lib = _lib.load()
    coords = _need(coords, torch.int32, "coords")
    n = coords.shape[0]
    dev = coords.device
    table = HashTable(n, dev)
    out = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
    n_out = _zeros(1, torch.int32, dev)
    f2c = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    info = _zeros(2, torch.int32, dev)
    wsb = lib.pp_stride_coords_workspace(n)
    ws = _ws(wsb, dev)
    _lib.check(lib.pp_stride_coords(_ptr(coords), n, int(ts_out), _ptr(table.keys), _ptr(table.vals), table.cap, _ptr(out),
                                    _ptr(n_out), _ptr(f2c), _ptr(ws), wsb, _ptr(info), _stream()), "pp_stride_coords")
    k = int(n_out.item())
    if int(info[1].item()):
        raise _lib.PanopticHipError("coordinates outside the packable range")
    return out[:k], table, f2c[:n]


def kernel_map(out_coords, table, ksize, step, sign):
    """nbr int32 [K, n_out]: row in `table`'s map of out_coords + sign*offset*step (or -1)."""
    lib = _lib.load()
    out_coords = _need(out_coords, torch.int32, "out_coords")
    n_out = out_coords.shape[0]
    K = ksize ** 3
    nbr = torch.empty((K, n_out), dtype=torch.int32, device=out_coords.device)
    pairs = _zeros(1, torch.int64, out_coords.device)
    _lib.check(lib.pp_kernel_map(_ptr(out_coords), n_out, _ptr(table.keys), _ptr(table.vals), table.cap, ksize, int(step),
                                 int(sign), _ptr(nbr), _ptr(pairs), _stream()), "pp_kernel_map")
    nbr.pp_pairs = _register_pairs(pairs)  # device scalar: number of (in, out) pairs (flops / density bookkeeping, no sync here)
    return nbr


class BlockIndex:
    """Block index of a coordinate level (csrc/pp_blockindex.hip): per group of <= 4096 voxels the first row and 64
    records (64 bits of the occupancy map, row of the word's first voxel), plus a hash block key -> block number."""

    __slots__ = ("unit", "block_bits", "n_blocks", "cap", "bkeys", "bvals", "start", "rec", "bkey_ord")


def block_index_build(coords_sorted, unit, block_bits):
    """coords_sorted int32 [n,4] in morton_order(unit, block_bits) order -> (BlockIndex, n_duplicate_rows)."""
    lib = _lib.load()
    coords = _need(coords_sorted, torch.int32, "coords_sorted")
    n = coords.shape[0]
    dev = coords.device
    row_block = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    counts = _zeros(4, torch.int32, dev)
    wsb = lib.pp_block_index_workspace(n)
    ws = _ws(wsb, dev, tag="block_index")
    _lib.check(lib.pp_block_index_count(_ptr(coords), n, int(unit), int(block_bits), _ptr(row_block), _ptr(counts), _ptr(ws),
                                        wsb, _stream()), "pp_block_index_count")
    nb, ndup, unsorted, oor = [int(v) for v in counts.tolist()]
    if unsorted:
        raise _lib.PanopticHipError("block_index_build: rows are not in morton_order(unit=%d, block_bits=%d) order" % (unit, block_bits))
    if oor:
        raise _lib.PanopticHipError("%d coordinates outside the 16-bit key range" % oor)
    bi = BlockIndex()
    bi.unit, bi.block_bits, bi.n_blocks = int(unit), int(block_bits), nb
    bi.cap = int(lib.pp_block_index_capacity(nb))
    bi.bkeys = torch.empty(bi.cap, dtype=torch.int64, device=dev)
    bi.bvals = torch.empty(bi.cap, dtype=torch.int32, device=dev)
    bi.start = torch.empty(max(nb, 1), dtype=torch.int32, device=dev)
    bi.rec = torch.empty(max(nb, 1) * 128, dtype=torch.int64, device=dev)
    bi.bkey_ord = torch.empty(max(nb, 1), dtype=torch.int64, device=dev)
    if ndup == 0:
        _lib.check(lib.pp_block_index_fill(_ptr(coords), n, int(unit), int(block_bits), _ptr(row_block), nb, _ptr(bi.bkeys),
                                           _ptr(bi.bvals), bi.cap, _ptr(bi.start), _ptr(bi.rec),
                                           _ptr(bi.bkey_ord), _stream()), "pp_block_index_fill")
    return bi, ndup


def block_index_coarsen(fine, n_fine_rows):
    """Next coarser level (tensor stride doubled) from a BlockIndex alone: (BlockIndex, coords int32 [n_coarse,4])."""
    lib = _lib.load()
    dev = fine.rec.device
    nbf = fine.n_blocks
    bi = BlockIndex()
    bi.unit, bi.block_bits = fine.unit * 2, fine.block_bits
    bi.cap = int(lib.pp_block_index_capacity(nbf))
    bi.bkeys = torch.empty(bi.cap, dtype=torch.int64, device=dev)
    bi.bvals = torch.empty(bi.cap, dtype=torch.int32, device=dev)
    bi.start = torch.empty(max(nbf, 1), dtype=torch.int32, device=dev)
    bi.rec = torch.empty(max(nbf, 1) * 128, dtype=torch.int64, device=dev)
    bi.bkey_ord = torch.empty(max(nbf, 1), dtype=torch.int64, device=dev)
    coords = torch.empty((max(int(n_fine_rows), 1), 4), dtype=torch.int32, device=dev)
    counts = _zeros(2, torch.int32, dev)
    wsb = lib.pp_block_index_coarsen_workspace(nbf)
    ws = _ws(wsb, dev)
    _lib.check(lib.pp_block_index_coarsen(_ptr(fine.bkey_ord), _ptr(fine.rec), nbf, bi.unit, bi.block_bits, _ptr(bi.bkeys),
                                          _ptr(bi.bvals), bi.cap, _ptr(bi.start), _ptr(bi.rec), _ptr(bi.bkey_ord),
                                          _ptr(coords), _ptr(counts), _ptr(ws), wsb, _stream()), "pp_block_index_coarsen")
    nbc, nc = [int(v) for v in counts.tolist()]
    bi.n_blocks = nbc
    return bi, coords[:nc]


def block_index_coarsen_chain(fine, n_fine_rows, levels):
    """`levels` successive coarsenings of a BlockIndex in one library call (csrc/pp_blockindex.hip: the counts of a level feed the
    next level's launches from device memory) and ONE host read for all the level sizes: [(BlockIndex, coords int32 [n_l, 4]), ...],
    element l at tensor stride fine.unit << (l + 1).  Every level is a view of arrays with the input level's capacity (the level
    sizes are not known when they are allocated); levels that turn out much smaller than that get compact copies of their coordinate
    rows so that the large blocks go back to the allocator."""
    lib = _lib.load()
    dev = fine.rec.device
    nbf, n0, L = int(fine.n_blocks), int(n_fine_rows), int(levels)
    nbm, nrm = max(nbf, 1), max(n0, 1)
    cap = int(lib.pp_block_index_capacity(nbf))
    bkeys = torch.empty((L, cap), dtype=torch.int64, device=dev)
    bvals = torch.empty((L, cap), dtype=torch.int32, device=dev)
    start = torch.empty((L, nbm), dtype=torch.int32, device=dev)
    rec = torch.empty((L, nbm * 128), dtype=torch.int64, device=dev)
    bkey_ord = torch.empty((L, nbm), dtype=torch.int64, device=dev)
    coords_all = torch.empty((L, nrm, 4), dtype=torch.int32, device=dev)
    counts = _zeros(2 * L, torch.int32, dev)
    wsb = lib.pp_block_index_coarsen_workspace(nbf)
    ws = _ws(wsb, dev)
    _lib.check(lib.pp_block_index_coarsen_chain(_ptr(fine.bkey_ord), _ptr(fine.rec), nbf, n0, fine.unit, fine.block_bits, L,
                                                _ptr(bkeys), _ptr(bvals), cap, _ptr(start), _ptr(rec), _ptr(bkey_ord),
                                                _ptr(coords_all), _ptr(counts), _ptr(ws), wsb, _stream()),
               "pp_block_index_coarsen_chain")
    vals = counts.tolist()  # the chain's one host read
    out = []
    for l in range(L):
        nbc, nc = int(vals[2 * l]), int(vals[2 * l + 1])
        bi = BlockIndex()
        bi.unit, bi.block_bits, bi.n_blocks, bi.cap = fine.unit << (l + 1), fine.block_bits, nbc, cap
        bi.bkeys, bi.bvals, bi.start, bi.rec, bi.bkey_ord = bkeys[l], bvals[l], start[l], rec[l], bkey_ord[l]
        # the coordinate rows of ALL levels sit in one block of levels x n_fine rows: they are copied out level by level (the sum of
        # the level sizes, a fraction of the block) and the block goes back to the allocator
        out.append((bi, coords_all[l, :nc].clone()))
    return out


def kernel_map_bi(out_coords, index, ksize, step, sign, want_mask=False, translate=None):
    """nbr int32 [27, n_out] through a BlockIndex: row in the indexed level of out_coords + sign*offset*step (or -1).
    translate (int32 [rows of the indexed level], optional): found rows r are stored as translate[r]."""
    if ksize != 3:
        raise NotImplementedError("kernel_map_bi: 3x3x3 kernels only")
    lib = _lib.load()
    out_coords = _need(out_coords, torch.int32, "out_coords")
    n_out = out_coords.shape[0]
    nbr = torch.empty((27, n_out), dtype=torch.int32, device=out_coords.device)
    pairs = _zeros(1, torch.int64, out_coords.device)
    mask = torch.empty(max(n_out, 1), dtype=torch.int32, device=out_coords.device) if want_mask else None
    _lib.check(lib.pp_kernel_map_bi(_ptr(out_coords), n_out, _ptr(index.bkeys), _ptr(index.bvals), index.cap,
                                    _ptr(index.rec), index.unit, index.block_bits, int(step), int(sign),
                                    _ptr(nbr), _ptr(pairs), _ptr(mask), _ptr(_need(translate, torch.int32, "translate")),
                                    _stream()), "pp_kernel_map_bi")
    nbr.pp_pairs = _register_pairs(pairs)
    if want_mask:
        nbr.pp_mask = mask  # int32 [n_out]: bit k set <=> offset k occupied
    return nbr


def exclusive_scan(x, want_total=False):
    """exclusive prefix sum of an int32 vector (csrc/pp_scan.hip); want_total: (scan, total int32 [1])"""
    lib = _lib.load()
    x = _need(x, torch.int32, "x")
    n = x.shape[0]
    out = torch.empty_like(x)
    total = _zeros(1, torch.int32, x.device) if want_total else None
    wsb = lib.pp_exclusive_scan_workspace(n)
    ws = _ws(wsb, x.device, tag="scan")
    _lib.check(lib.pp_exclusive_scan(_ptr(x), _ptr(out), n, _ptr(total), _ptr(ws), wsb, _stream()), "pp_exclusive_scan")
    return (out, total) if want_total else out


def select_indices(flags):
    """positions of the non-zero entries of a bool / uint8 vector, ascending (int64) -- torch.nonzero(flags).view(-1) on the
    library's own scan (one host read: the count)"""
    lib = _lib.load()
    if flags.dtype == torch.bool:
        flags = flags.view(torch.uint8)
    flags = _need(flags, torch.uint8, "flags")
    n = flags.shape[0]
    idx = torch.empty(max(n, 1), dtype=torch.int64, device=flags.device)
    count = _zeros(1, torch.int32, flags.device)
    wsb = lib.pp_select_workspace(n)
    ws = _ws(wsb, flags.device, tag="select")
    _lib.check(lib.pp_select_indices(_ptr(flags), n, _ptr(idx), _ptr(count), _ptr(ws), wsb, _stream()), "pp_select_indices")
    return idx[: int(count.item())]


def run_lengths(values, want_run_id=True):
    """runs of equal consecutive values of an int64 vector: (heads int64 [n] , starts int32 [n + 1], run_id int32 [n] or None,
    n_runs int32 [1] on the device) -- only the first n_runs (+ 1) entries of heads / starts are meaningful; no host read"""
    lib = _lib.load()
    values = _need(values, torch.int64, "values")
    n = values.shape[0]
    dev = values.device
    heads = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    starts = torch.empty(n + 1, dtype=torch.int32, device=dev)
    run_id = torch.empty(max(n, 1), dtype=torch.int32, device=dev) if want_run_id else None
    n_runs = _zeros(1, torch.int32, dev)
    wsb = lib.pp_select_workspace(n)
    ws = _ws(wsb, dev, tag="select")
    _lib.check(lib.pp_run_lengths(_ptr(values), n, _ptr(run_id), _ptr(heads), _ptr(starts), _ptr(n_runs), _ptr(ws), wsb, _stream()),
               "pp_run_lengths")
    return heads, starts, (run_id[:n] if run_id is not None else None), n_runs


def sort_pairs(keys, vals, end_bit=None):
    """stable sort of (key, value) pairs by the low end_bit bits of the keys (int32 / int64 tensors read as unsigned; values
    int32): (sorted keys, sorted values)"""
    lib = _lib.load()
    if keys.dtype not in (torch.int32, torch.int64):
        raise TypeError("sort_pairs: keys must be int32 or int64")
    keys = _need(keys, keys.dtype, "keys")
    vals = _need(vals, torch.int32, "vals")
    n = keys.shape[0]
    kb = keys.element_size()
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    wsb = lib.pp_sort_pairs_workspace_bytes(n)
    ws = _ws(wsb, keys.device, tag="sort")
    _lib.check(lib.pp_sort_pairs(_ptr(keys), _ptr(ko), kb, _ptr(vals), _ptr(vo), n, 8 * kb if end_bit is None else int(end_bit),
                                 _ptr(ws), wsb, _stream()), "pp_sort_pairs")
    return ko, vo


def map_mask(nbr):
    """uint32-as-int32 [n_out]: bit k set <=> nbr[k][o] >= 0"""
    lib = _lib.load()
    K, n_out = nbr.shape
    mask = torch.empty(max(n_out, 1), dtype=torch.int32, device=nbr.device)
    _lib.check(lib.pp_map_mask(_ptr(nbr), K, n_out, _ptr(mask), _stream()), "pp_map_mask")
    return mask[:n_out]


def map_order(mask, window=None):
    """slot order of a kernel map (csrc/pp_maporder.hip): order[s] = output row taking slot s.  Rows are sorted by
    (neighbour mask, row) inside windows of `window` (default pp_map_window()) consecutive rows."""
    lib = _lib.load()
    mask = _need(mask, torch.int32, "mask")
    n = mask.shape[0]
    order = torch.empty(max(n, 1), dtype=torch.int32, device=mask.device)
    if window is None:
        window = int(lib.pp_map_window())
        _lib.check(lib.pp_map_order(_ptr(mask), n, _ptr(order), _stream()), "pp_map_order")
    else:
        _lib.check(lib.pp_map_order_window(_ptr(mask), n, int(window), _ptr(order), _stream()), "pp_map_order_window")
    order = order[:n]
    order.pp_window = int(window)  # map_permute stages window slices in LDS
    return order


class CompactMap:
    """compact form of a same-level kernel map (csrc/pp_maporder.hip): mask int32 [n], start int32 [ceil(n / 32) + 1], entries
    int32 / tags int16 [capacity] (the first start[-1] valid)"""
    __slots__ = ("mask", "start", "entries", "tags", "n")

    def __init__(self, mask, start, entries, tags, n):
        self.mask, self.start, self.entries, self.tags, self.n = mask, start, entries, tags, n


def map_compact(nbr):
    """CompactMap of a dense same-level map [27, n] (row = slot).  No host read: the entry arrays have the map's capacity
    (they are written and read only up to the number of pairs)."""
    lib = _lib.load()
    nbr = _need(nbr, torch.int32, "nbr")
    K, n = nbr.shape
    dev = nbr.device
    chunks = (n + 31) // 32
    mask = torch.empty(n, dtype=torch.int32, device=dev)
    start = torch.empty(chunks + 1, dtype=torch.int32, device=dev)
    wsb = lib.pp_map_compact_workspace(n)
    ws = _ws(wsb, dev)
    _lib.check(lib.pp_map_compact_count(_ptr(nbr), K, n, _ptr(mask), _ptr(start), _ptr(ws), wsb, _stream()), "pp_map_compact_count")
    entries = torch.empty(K * n, dtype=torch.int32, device=dev)
    tags = torch.empty(K * n, dtype=torch.int16, device=dev)
    _lib.check(lib.pp_map_compact_write(_ptr(nbr), K, n, _ptr(start), _ptr(entries), _ptr(tags), _stream()), "pp_map_compact_write")
    return CompactMap(mask, start, entries, tags, n)


def map_permute(nbr, order=None, translate=None):
    """out[k][s] = T(nbr[k][order[s]]) with T(v) = -1 if v < 0 else (translate[v] if translate is given else v)"""
    lib = _lib.load()
    nbr = _need(nbr, torch.int32, "nbr")
    K, n_out = nbr.shape
    out = torch.empty_like(nbr)
    _lib.check(lib.pp_map_permute(_ptr(nbr), K, n_out, _ptr(_need(order, torch.int32, "order")),
                                  _ptr(_need(translate, torch.int32, "translate")), 0 if translate is None else translate.shape[0],
                                  int(getattr(order, "pp_window", 0)), _ptr(out), _stream()), "pp_map_permute")
    if hasattr(nbr, "pp_pairs"):
        out.pp_pairs = nbr.pp_pairs
    return out


def level_permute(coords, order):
    """(coords[order], inverse) with inverse[order[s]] = s"""
    lib = _lib.load()
    coords = _need(coords, torch.int32, "coords")
    order = _need(order, torch.int32, "order")
    n = coords.shape[0]
    out = torch.empty_like(coords)
    inv = torch.empty(max(n, 1), dtype=torch.int32, device=coords.device)
    _lib.check(lib.pp_level_permute(_ptr(coords), n, _ptr(order), _ptr(out), _ptr(inv), _stream()), "pp_level_permute")
    return out, inv[:n]


def kernel_map_transpose(nbr, n_in, order=None):
    """map of the transposed strided conv from the strided conv's map: out[k][nbr[k][o]] = o (order[o] when the map
    is slot-ordered).  The result is indexed by physical rows of the finer level."""
    lib = _lib.load()
    K, n_out = nbr.shape
    out = torch.empty((K, n_in), dtype=torch.int32, device=nbr.device)
    _lib.check(lib.pp_kernel_map_transpose(_ptr(nbr), n_out, K, int(n_in), _ptr(_need(order, torch.int32, "order")), _ptr(out),
                                           _stream()), "pp_kernel_map_transpose")
    if hasattr(nbr, "pp_pairs"):
        out.pp_pairs = nbr.pp_pairs  # same pairs, roles swapped
    return out


def kernel_map_transpose8(nbr, n_in, order=None, want_key=True):
    """8-wide form of kernel_map_transpose for a stride-2 transposed convolution (csrc/pp_coords.hip): (map8 int32 [8, n_in]
    with entries coarse row | parity class of the fine row << 28, key int32 [n_in] = class << 8 | presence bits or None)."""
    lib = _lib.load()
    K, n_out = nbr.shape
    if K != 27:
        raise ValueError("kernel_map_transpose8: 3x3x3 maps only")
    dev = nbr.device
    map8 = torch.empty((8, n_in), dtype=torch.int32, device=dev)
    key = torch.empty(max(n_in, 1), dtype=torch.int32, device=dev) if want_key else None
    _lib.check(lib.pp_kernel_map_transpose8(_ptr(nbr), n_out, int(n_in), _ptr(_need(order, torch.int32, "order")), _ptr(map8),
                                            _ptr(key), _stream()), "pp_kernel_map_transpose8")
    if hasattr(nbr, "pp_pairs"):
        map8.pp_pairs = nbr.pp_pairs
    return map8, (key[:n_in] if want_key else None)


def map8_to_dense(map8):
    """the dense [27, n] map (same column order) an 8-wide transposed map stands for (tests / consumers outside the convolution)"""
    n = map8.shape[1]
    present = map8 >= 0
    cls = torch.where(present, (map8.long() >> 28) & 7, torch.zeros_like(map8, dtype=torch.int64)).amax(0)
    dense = torch.full((27, n), -1, dtype=torch.int32, device=map8.device)
    cols = torch.arange(n, device=map8.device)
    for j in range(8):
        d = [torch.where((cls >> a) & 1 == 1, torch.full_like(cls, 2 if (j >> a) & 1 else 0), torch.ones_like(cls)) for a in range(3)]
        k = d[0] + 3 * d[1] + 9 * d[2]
        sel = present[j]
        dense[k[sel], cols[sel]] = map8[j][sel] & 0x0FFFFFFF
    return dense


def compose_perm(perm32, order, n, device):
    """(perm, inv) int64: perm[s] = perm32[order[s]] (None = identity for either), inv[perm[s]] = s"""
    lib = _lib.load()
    perm = torch.empty(max(n, 1), dtype=torch.int64, device=device)
    inv = torch.empty(max(n, 1), dtype=torch.int64, device=device)
    _lib.check(lib.pp_compose_perm(_ptr(_need(perm32, torch.int32, "perm32")), _ptr(_need(order, torch.int32, "order")), n, _ptr(perm),
                                   _ptr(inv), _stream()), "pp_compose_perm")
    return perm[:n], inv[:n]


def morton_order(coords, unit=1, block_bits=0, want_sorted=False, raw=False):
    """perm (int64 [n]): rows of `coords` in batch-major Z-order (block_bits=0) or parity-grouped block order.
    want_sorted: also return coords[perm], decoded from the sorted keys instead of gathered."""
    lib = _lib.load()
    coords = _need(coords, torch.int32, "coords")
    n = coords.shape[0]
    dev = coords.device
    perm = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    info = _zeros(2, torch.int32, dev)
    wsb = lib.pp_morton_order_workspace(n)
    ws = _ws(wsb, dev)
    srt = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev) if want_sorted else None
    _lib.check(lib.pp_morton_order(_ptr(coords), n, int(unit), int(block_bits), _ptr(perm), _ptr(srt), _ptr(ws), wsb, _ptr(info),
                                   _stream()), "pp_morton_order")
    if raw:  # int32 permutation (compose_perm turns it into the int64 pair the feature gathers use)
        return (perm[:n], srt[:n]) if want_sorted else perm[:n]
    if want_sorted:
        return perm[:n].long(), srt[:n]
    return perm[:n].long()


# ------------------------------------------------------------------------------------------ convolution
def pack_weight(weight, transpose=False, kflip=False):
    """ME-layout kernel [K,Cin,Cout] (or [Cin,Cout]) -> MFMA fragment order.  transpose=True packs W_k^T; kflip=True
    packs the offsets in reverse order (mirrored same-level map without flipping the map)."""
    lib = _lib.load()
    w = _need(weight.detach(), torch.float32, "weight")
    if w.dim() == 2:
        w = w.unsqueeze(0)
    K, cin, cout = w.shape
    if transpose:
        cin, cout = cout, cin
    packed = torch.empty(lib.pp_packed_weight_floats(K, cin, cout), dtype=torch.float32, device=w.device)
    _lib.check(lib.pp_pack_weight(_ptr(w), K, cin, cout, int(bool(transpose)) | (2 if kflip else 0), _ptr(packed),
                                  _stream()), "pp_pack_weight")
    return packed


# Derived copies of parameters (packed weights, folded BatchNorm) are keyed by the parameters' `_version`.  Not every writer
# bumps it: torch's fused optimizers (Adam(fused=True), ...) update the parameters in place WITHOUT touching the version
# counter (measured on torch 2.10: foreach -> version + 1, fused -> unchanged), and a stale packed copy trains silently
# on the old weights.  So every optimizer step also advances an epoch that is part of every such key.
_PARAM_EPOCH = [0]


def param_epoch():
    return _PARAM_EPOCH[0]


def parameters_changed():
    """call after writing parameters through a path that does not bump tensor versions (`.data`, custom fused kernels)"""
    _PARAM_EPOCH[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook
    register_optimizer_step_post_hook(lambda *a, **k: parameters_changed())
except ImportError:  # very old torch: versions only
    pass


class _PackedWeights:
    """Packed copies of a model's convolution weights, refreshed in ONE launch per weight version (training: the optimizer
    step changes every weight, and packing each of the ~190 (layer, orientation) pairs of a step as its own 4 us launch
    costs 2 ms of host time per step).  An entry is (parameter, flags) -> packed tensor; the first request of an entry packs
    it alone and registers it; when a registered entry is requested with a new version of its parameter (or after an optimizer step, see above), ALL registered
    entries are re-packed by pp_pack_weights_batched (parameters whose weights did not change are simply packed again)."""

    def __init__(self):
        self.entries = {}      # (id(param), flags) -> [weakref(param), flags, packed, tag, K, cin, cout]
        self.tables = {}       # device -> (desc, first_block, n, total_blocks, source pointers), dropped when its entries change

    @staticmethod
    def _tag(param):
        # data_ptr: `p.data = ...`, vector_to_parameters, EMA / SWA swaps and to_empty replace a parameter's storage without
        # touching its version or the optimizer epoch -- the packed copy (and the pointer table below) would go stale
        return (param._version, _PARAM_EPOCH[0], param.data_ptr())

    def get(self, param, transpose, kflip):
        import weakref
        flags = int(bool(transpose)) | (2 if kflip else 0)
        key = (id(param), flags)
        ent = self.entries.get(key)
        if ent is not None and ent[0]() is param and ent[2].device == param.device:
            if ent[3] != self._tag(param):
                self._repack_all(param.device)
            return ent[2]
        packed = pack_weight(param, transpose=transpose, kflip=kflip)
        w = param.detach()
        K, a, b = (1,) + tuple(w.shape) if w.dim() == 2 else tuple(w.shape)
        cin, cout = (b, a) if transpose else (a, b)
        self.entries[key] = [weakref.ref(param), flags, packed, self._tag(param), K, cin, cout]
        self.tables.pop(param.device, None)
        return packed

    def _repack_all(self, device):
        lib = _lib.load()
        dead = [k for k, e in self.entries.items() if e[0]() is None]
        if dead:  # only dead parameters leave; entries of OTHER devices stay (one process may drive several GPUs)
            for k in dead:
                del self.entries[k]
            self.tables.clear()
        live = [e for e in self.entries.values() if e[2].device == device]
        ptrs = tuple(e[0]().data_ptr() for e in live)
        table = self.tables.get(device)
        if table is None or table[4] != ptrs:
            rows, first, blocks = [], [0], 0
            for e, src in zip(live, ptrs):
                rows.append([src, e[2].data_ptr(), e[4], e[5], e[6], e[1]])
                blocks += (e[2].numel() + 255) // 256
                first.append(blocks)
            table = self.tables[device] = (torch.tensor(rows, dtype=torch.int64).to(device),
                                           torch.tensor(first, dtype=torch.int64).to(device), len(rows), blocks, ptrs)
        desc, first, n, blocks, _ = table
        _lib.check(lib.pp_pack_weights_batched(_ptr(desc), _ptr(first), n, blocks, _stream()), "pp_pack_weights_batched")
        for e in live:
            e[3] = self._tag(e[0]())


PACKED_WEIGHTS = _PackedWeights()


def pack_weight_cached(param, transpose=False, kflip=False):
    """pack_weight for a model PARAMETER: packed once per weight version, all registered parameters in one launch"""
    return PACKED_WEIGHTS.get(param, transpose, kflip)


_CONV_SCRATCH = {"key": None, "bufs": {}}
CONV_SCRATCH_BYTES = int(os.environ.get("PP_CONV_SCRATCH_MB", "64")) << 20


def _conv_scratch(lib, device):
    """registers the split-K scratch of small convolution launches (pp_spconv_set_scratch) for the stream the next convolution
    is launched on: the library holds ONE pointer and reads it at launch time, the buffers are owned here, one per (device,
    stream) -- two streams' convolutions (the next batch's backbone beside this batch's grouping, scene.TileRunner) must not
    add their partial sums in the same block"""
    key = (device, _stream().value)
    if _CONV_SCRATCH["key"] != key:
        buf = _CONV_SCRATCH["bufs"].get(key)
        if buf is None and CONV_SCRATCH_BYTES > 0:
            buf = _CONV_SCRATCH["bufs"][key] = torch.empty(CONV_SCRATCH_BYTES, dtype=torch.uint8, device=device)
        _lib.check(lib.pp_spconv_set_scratch(_ptr(buf), CONV_SCRATCH_BYTES if buf is not None else 0), "pp_spconv_set_scratch")
        _CONV_SCRATCH["key"] = key


def bf16_conv_supported(c0, c1, K, nbr_given=True):
    """layers the bfloat16 entries take (the 4-channel input layer and 1x1 shortcuts without a map stay fp32)"""
    return c0 % 16 == 0 and c1 % 16 == 0 and (c1 == 0 or c1 == c0) and K <= 28


# A/B measurements only: PP_AB_T4="max_ntw,min_rows" overrides the library's rows-per-wave choice (64 rows when the launch
# has <= max_ntw column tiles per wave and >= min_rows rows) through pp_spconv_fwd_ex
_AB_T4 = tuple(int(v) for v in os.environ["PP_AB_T4"].split(",")) if os.environ.get("PP_AB_T4") else None


_CONV_LOCK = threading.Lock()


def spconv_fwd(*args, **kwargs):
    """pp_spconv_fwd and its variants (see _spconv_fwd).  The library reads ONE split-K scratch pointer at launch time and this
    module registers the buffer of the launching stream right before the launch: the pair is atomic, because two host threads
    may launch convolutions on two streams (scene.TileRunner's backbone ahead in its own thread)."""
    with _CONV_LOCK:
        return _spconv_fwd(*args, **kwargs)


def _spconv_fwd(in0, packed, nbr, n_out, cout, K, in1=None, scale=None, shift=None, relu=False, residual=None, out=None,
                row_order=None, bf16=False, variant=None, shortcut=None):
    """variant = (rows_per_wave, pipeline, split_k): an explicit variant of the pipelined kernel through
    pp_spconv_fwd_ex (tests / A-B runs); None = the library's per-shape choice.  row_order: slot order of a
    cross-level map (nbr is slot-major then).  shortcut = (x [n_out, c], packed 1x1 weights, scale, shift): the 1x1
    shortcut of a residual block fused in (pp_spconv_fwd_shortcut); returns None when the library does not serve the
    shape that way -- nothing has been launched then and the caller runs the shortcut as its own convolution."""
    lib = _lib.load()
    in0 = _need(in0, torch.float32, "in0")
    in1 = _need(in1, torch.float32, "in1")
    c0 = in0.shape[1]
    c1 = 0 if in1 is None else in1.shape[1]
    if in1 is not None and in1.shape[0] != in0.shape[0]:
        raise ValueError("in0 and in1 must have the same number of rows")
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=in0.device)
    scale = _need(scale, torch.float32, "scale")
    shift = _need(shift, torch.float32, "shift")
    residual = _need(residual, torch.float32, "residual")
    row_order = _need(row_order, torch.int32, "row_order")
    prof = PROFILER
    if prof is not None:
        e0, e1 = prof.events()
        e0.record()
    _conv_scratch(lib, in0.device)
    args = (_ptr(in0), c0, _ptr(in1), c1, in0.shape[0], _ptr(packed), _ptr(nbr), K, n_out, cout, _ptr(scale),
            _ptr(shift), int(bool(relu)), _ptr(residual), _ptr(row_order), _ptr(out))
    use_bf16 = bool(bf16) and bf16_conv_supported(c0, c1, K)
    if variant is None and _AB_T4 is not None and c0 % 16 == 0 and c1 % 16 == 0 and K == 27:
        nt = (cout + 15) // 16
        groups = (nt + 3) // 4
        ntw = (nt + groups - 1) // groups
        variant = (64 if (ntw <= _AB_T4[0] and n_out >= _AB_T4[1]) else 32, 0, 0)
    t8 = bool(getattr(nbr, "pp_t8", False))
    cmap = None
    if variant is None and row_order is None and K == 27 and not t8 and n_out == in0.shape[0]:
        # (the compact form is read by the pipelined kernel only: 16-channel steps from one source or two equal ones, or the input layer)
        if ((c0 % 16 == 0 and c1 in (0, c0)) or (c0 == 4 and c1 == 0)) and in0.shape[0] * c0 * 4 < 4294967000:
            cmap = getattr(nbr, "pp_cmap", None)
        # ... and not by the full-line form of the split kernel (k_spconv_x3f: dense and 8-wide maps).  A launch that kernel serves
        # keeps its dense map: which kernel family evaluates a layer must not depend on the map's form (one-column-tile layers
        # are on the split kernel only in that form), and results never do
        if cmap is not None and lib.pp_spconv_kernel_family(c0, c1, in0.shape[0], K, n_out, cout, int(shortcut is not None)) == 2:
            cmap = None
    if cmap is not None:
        # same-level map in its compact form (map_compact): 4 + 6 x pairs bytes per row in the prologue instead of 108
        if shortcut is not None:
            xs, pks, scs, shs = shortcut
            xs = _need(xs, torch.float32, "shortcut input")
            ds = (_ptr(xs), xs.shape[1], _ptr(pks), _ptr(_need(scs, torch.float32, "shortcut scale")),
                  _ptr(_need(shs, torch.float32, "shortcut shift")))
            bf = int(use_bf16 and xs.shape[1] % 16 == 0)
        else:
            ds, bf = (None, 0, None, None, None), int(use_bf16)
        rc = lib.pp_spconv_fwd_cmap(_ptr(in0), c0, _ptr(in1), c1, in0.shape[0], _ptr(packed), _ptr(cmap.mask), _ptr(cmap.start),
                                    _ptr(cmap.entries), _ptr(cmap.tags), n_out, cout, _ptr(scale), _ptr(shift), int(bool(relu)),
                                    _ptr(residual), _ptr(out), bf, *ds, _stream())
        if rc == _lib.PP_UNSUPPORTED:
            if prof is not None:
                prof._pool.extend((e0, e1))  # nothing was launched: the events go back
            return None
        _lib.check(rc, "pp_spconv_fwd_cmap")
    elif t8:
        if variant is not None or shortcut is not None or K != 27 or row_order is None:
            raise ValueError("spconv_fwd: an 8-wide transposed map takes the default variant, K = 27 and its slot order")
        _lib.check(lib.pp_spconv_fwd_t8(_ptr(in0), c0, _ptr(in1), c1, in0.shape[0], _ptr(packed), _ptr(nbr), n_out, cout,
                                        _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(residual), _ptr(row_order), _ptr(out),
                                        int(use_bf16), _stream()), "pp_spconv_fwd_t8")
    elif shortcut is not None:
        xs, pks, scs, shs = shortcut
        xs = _need(xs, torch.float32, "shortcut input")
        rc = lib.pp_spconv_fwd_shortcut(*args[:14], _ptr(out), int(use_bf16 and xs.shape[1] % 16 == 0), _ptr(xs), xs.shape[1], _ptr(pks),
                                        _ptr(_need(scs, torch.float32, "shortcut scale")), _ptr(_need(shs, torch.float32, "shortcut shift")),
                                        _stream())
        if rc == _lib.PP_UNSUPPORTED:
            if prof is not None:
                prof._pool.extend((e0, e1))  # nothing was launched: the events go back
            return None
        _lib.check(rc, "pp_spconv_fwd_shortcut")
    elif variant is not None:
        rpw, pipe, split = variant
        _lib.check(lib.pp_spconv_fwd_ex(*args, int(use_bf16), int(rpw), int(pipe), int(split), _stream()), "pp_spconv_fwd_ex")
    else:
        fn = lib.pp_spconv_fwd_bf16 if use_bf16 else lib.pp_spconv_fwd
        _lib.check(fn(*args, _stream()), "pp_spconv_fwd")
    if prof is not None:
        e1.record()
        prof.records.append((e0, e1, in0.shape[0], n_out, c0 + c1, cout, K, _pairs_of(nbr), residual is not None,
                             shortcut[0].shape[1] if shortcut is not None else 0))
        prof.tags.append(PROFILE_TAG)
        prof.fams.append("fwd3" if variant is not None else LaunchProfiler.kernel_family(c0 + c1, cout, K, in0.shape[0], n_out, c1, shortcut is not None))
        prof.map_k.append(-1 if cmap is not None else (8 if t8 else K))  # (-1: compact map, bytes from the pair count)
    return out


def spconv_bwd_weight(inp, dout, nbr, K, bf16=False):
    lib = _lib.load()
    inp = _need(inp, torch.float32, "in")
    dout = _need(dout, torch.float32, "dout")
    cin, cout = inp.shape[1], dout.shape[1]
    dw = torch.empty((K, cin, cout), dtype=torch.float32, device=inp.device)
    fn = lib.pp_spconv_bwd_weight_bf16 if (bf16 and nbr is not None and cout <= 192) else lib.pp_spconv_bwd_weight
    prof = PROFILER
    if prof is not None:
        e0, e1 = prof.events()
        e0.record()
    _lib.check(fn(_ptr(inp), cin, inp.shape[0], _ptr(dout), cout, _ptr(nbr), K, dout.shape[0], _ptr(dw), _stream()),
               "pp_spconv_bwd_weight")
    if prof is not None:
        e1.record()
        prof.records_w.append((e0, e1, inp.shape[0], dout.shape[0], cin, cout, K, _pairs_of(nbr)))
    return dw


class WgradPairs:
    """the pairs of a kernel map compacted per offset (pp_wgrad_pairs_build): what spconv_bwd_weight_pairs walks"""
    __slots__ = ("pairs", "tile_start", "K", "rows", "n_pairs")

    def __init__(self, pairs, tile_start, K, rows, n_pairs):
        self.pairs, self.tile_start, self.K, self.rows, self.n_pairs = pairs, tile_start, K, rows, n_pairs


def wgrad_pairs(nbr, K, row_order=None):
    """(output row, input row) lists per offset of the dense map nbr [K, rows]; row_order folds a slot order in (row r of
    the map is output row row_order[r]).  The list buffer holds exactly the map's pairs (one host read per map, amortised
    over every layer and step that trains on it): typical maps hold 5.6 - 16 of the K = 27 neighbours per row, and the lists
    live as long as the coordinate manager."""
    lib = _lib.load()
    nbr = _need(nbr, torch.int32, "nbr")
    rows = nbr.shape[1]
    dev = nbr.device
    tiles = K * ((rows + 1023) // 1024)
    n_pairs = _pairs_host(_pairs_of(nbr))
    pairs = torch.empty((max(n_pairs, 1), 2), dtype=torch.int32, device=dev)
    tile_start = torch.empty(tiles + 1, dtype=torch.int32, device=dev)
    nbytes = lib.pp_wgrad_pairs_workspace(K, rows)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.pp_wgrad_pairs_build(_ptr(nbr), K, rows, _ptr(_need(row_order, torch.int32, "row_order")), _ptr(pairs),
                                        _ptr(tile_start), _ptr(ws), nbytes, _stream()), "pp_wgrad_pairs_build")
    return WgradPairs(pairs, tile_start, K, rows, _pairs_of(nbr))


# weight gradients without float atomics (PP_WGRAD_DETERMINISTIC=0: the atomic form; workspaces above the cap fall back to it)
WGRAD_DETERMINISTIC = os.environ.get("PP_WGRAD_DETERMINISTIC", "1") != "0"
WGRAD_DET_MAX_BYTES = int(os.environ.get("PP_WGRAD_DET_MAX_MB", "1024")) << 20
# PP_WGRAD_DETERMINISTIC=strict: a weight gradient whose partial-sum workspace exceeds the cap raises instead of falling back to the
# float-atomic kernel (the default warns once: from that size on the training step is no longer bit-reproducible)
WGRAD_STRICT = os.environ.get("PP_WGRAD_DETERMINISTIC", "1").lower() == "strict"
_WGRAD_WARNED = [False]


def spconv_bwd_weight_pairs(inp, dout, wp, bf16=False):
    """dW [K, cin, cout] over the pair lists of wgrad_pairs (dout in its own row order)"""
    lib = _lib.load()
    inp = _need(inp, torch.float32, "in")
    dout = _need(dout, torch.float32, "dout")
    cin, cout = inp.shape[1], dout.shape[1]
    dw = torch.empty((wp.K, cin, cout), dtype=torch.float32, device=inp.device)
    prof = PROFILER
    if prof is not None:
        e0, e1 = prof.events()
        e0.record()
    nbytes = lib.pp_spconv_bwd_weight_pairs_det_workspace(cin, cout, wp.K, wp.rows) if WGRAD_DETERMINISTIC else 0
    if WGRAD_DETERMINISTIC and nbytes <= WGRAD_DET_MAX_BYTES:
        # block partials + ordered reduction: the same bits run after run (float atomics made the loss differ in its last bits)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=inp.device)
        _lib.check(lib.pp_spconv_bwd_weight_pairs_det(_ptr(inp), cin, inp.shape[0], _ptr(dout), cout, dout.shape[0], _ptr(wp.pairs),
                                                      _ptr(wp.tile_start), wp.K, wp.rows, _ptr(dw), 1 if bf16 else 0, _ptr(ws), nbytes,
                                                      _stream()), "pp_spconv_bwd_weight_pairs_det")
    else:
        if WGRAD_DETERMINISTIC:
            msg = ("weight gradient %d -> %d over %d map rows: the ordered-reduction workspace (%d MiB) exceeds PP_WGRAD_DET_MAX_MB "
                   "(%d): float-atomic kernel, not bit-reproducible run to run" % (cin, cout, wp.rows, nbytes >> 20, WGRAD_DET_MAX_BYTES >> 20))
            if WGRAD_STRICT:
                raise _lib.PanopticHipError(msg)
            if not _WGRAD_WARNED[0]:
                _WGRAD_WARNED[0] = True
                import warnings
                warnings.warn(msg)
        _lib.check(lib.pp_spconv_bwd_weight_pairs(_ptr(inp), cin, inp.shape[0], _ptr(dout), cout, dout.shape[0], _ptr(wp.pairs),
                                                  _ptr(wp.tile_start), wp.K, wp.rows, _ptr(dw), 1 if bf16 else 0, _stream()),
                   "pp_spconv_bwd_weight_pairs")
    if prof is not None:
        e1.record()
        prof.records_w.append((e0, e1, inp.shape[0], dout.shape[0], cin, cout, wp.K, wp.n_pairs))
    return dw


def channel_stats(x):
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    n, c = x.shape
    s = torch.empty(c, dtype=torch.float64, device=x.device)
    ss = torch.empty(c, dtype=torch.float64, device=x.device)
    _lib.check(lib.pp_channel_stats(_ptr(x), n, c, _ptr(s), _ptr(ss), _stream()), "pp_channel_stats")
    return s, ss


def bn_bwd_reduce(x, dy):
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    dy = _need(dy, torch.float32, "dy")
    n, c = x.shape
    a = torch.empty(c, dtype=torch.float64, device=x.device)
    b = torch.empty(c, dtype=torch.float64, device=x.device)
    _lib.check(lib.pp_bn_bwd_reduce(_ptr(x), _ptr(dy), n, c, _ptr(a), _ptr(b), _stream()), "pp_bn_bwd_reduce")
    return a, b


# SyncBN for data-parallel training: the reference normalises over ALL cylinders of its batch of 4 on one GPU; with
# the batch sharded over ranks the per-channel sum / sum of squares / row count are all-reduced so that every rank uses the
# statistics of the whole batch.  None = per-replica statistics (the fused three-launch kernels).  training.enable_sync_bn sets it.
SYNC_BN_GROUP = None   # a torch.distributed process group, or True for the default group
SYNC_BN_STATS = {"all_reduces": 0}
# The collective schedule must be the same on every rank.  The backbone and the heads run the same layers on every rank whatever
# their cylinders hold; the proposal scorer does not -- a rank without proposals skips it, a rank with many runs it in several
# chunks (pointgroup3heads.group_and_score / _score_unique) -- so its BatchNorms keep per-replica statistics (their "batch" is the
# rank's proposals, not cylinders): `sync_bn_suspended()` around the scorer.  The decision is taken in the forward and kept by the
# autograd node (`use_sync` below), so that a layer's backward issues a collective exactly when its forward did.
_SYNC_BN_SUSPENDED = threading.local()


class sync_bn_suspended:
    """with ops.sync_bn_suspended(): training-mode BatchNorms inside use per-replica statistics even when SyncBN is on"""

    def __enter__(self):
        self.prev = getattr(_SYNC_BN_SUSPENDED, "on", False)
        _SYNC_BN_SUSPENDED.on = True
        return self

    def __exit__(self, *exc):
        _SYNC_BN_SUSPENDED.on = self.prev
        return False


def sync_bn_active():
    """True when a training-mode BatchNorm launched now would all-reduce its statistics"""
    return _sync_bn_group() is not None


def _sync_bn_group(use_sync=None):
    g = SYNC_BN_GROUP
    if g is None or use_sync is False or (use_sync is None and getattr(_SYNC_BN_SUSPENDED, "on", False)):
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = None if g is True else g
    return (dist, group) if dist.get_world_size(group) > 1 else None


def _all_reduce_f64(dist, group, t):
    """sum over the ranks of a small float64 device vector (RCCL all-reduce on device tensors; the gloo backend of the
    single-GPU test box reduces on the host)"""
    SYNC_BN_STATS["all_reduces"] += 1
    if dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        return h.to(t.device)
    dist.all_reduce(t, group=group)
    return t


def _bn_train_fwd_sync(sync, x, weight, bias, eps, momentum, running_mean, running_var, relu, num_batches_tracked):
    dist, group = sync
    n, c = x.shape
    if n:
        s, ss = channel_stats(x)
    else:  # a rank without rows still joins the collective
        s = ss = torch.zeros(c, dtype=torch.float64, device=x.device)
    tot = _all_reduce_f64(dist, group, torch.cat([s, ss, torch.tensor([float(n)], dtype=torch.float64, device=x.device)]))
    N = tot[2 * c]
    mean = tot[:c] / N
    var = (tot[c:2 * c] / N - mean * mean).clamp_min_(0.0)   # biased variance of the whole batch
    rstd = torch.rsqrt(var + eps)
    if running_mean is not None:
        with torch.no_grad():
            running_mean.mul_(1.0 - momentum).add_((momentum * mean).float())
            running_var.mul_(1.0 - momentum).add_((momentum * var * (N / (N - 1.0).clamp_min(1.0))).float())
            if num_batches_tracked is not None:
                num_batches_tracked.add_(1)
    w = torch.ones(c, dtype=torch.float64, device=x.device) if weight is None else weight.double()
    b = torch.zeros(c, dtype=torch.float64, device=x.device) if bias is None else bias.double()
    scale = w * rstd
    y = affine_act(x, scale.float(), (b - mean * scale).float(), act=1 if relu else 0)
    return y, mean, rstd


def _bn_train_bwd_sync(sync, x, dy, y_relu, weight, save_mean, save_rstd):
    dist, group = sync
    n, c = x.shape
    if y_relu is not None:
        dy = dy * (y_relu > 0).to(dy.dtype)
    if n:
        a, b = bn_bwd_reduce(x, dy)          # local sum(dy), sum(dy * x), float64
    else:
        a = b = torch.zeros(c, dtype=torch.float64, device=x.device)
    tot = _all_reduce_f64(dist, group, torch.cat([a, b, torch.tensor([float(n)], dtype=torch.float64, device=x.device)]))
    N = tot[2 * c]
    A = tot[:c]
    B = (tot[c:2 * c] - save_mean * A) * save_rstd      # sum over the whole batch of dy * xhat
    w = torch.ones(c, dtype=torch.float64, device=x.device) if weight is None else weight.double()
    # dx = w rstd (dy - mean(dy) - xhat mean(dy xhat)) with the means over the WHOLE batch, evaluated on CENTRED operands:
    #   dx = s (dy - A / N) + t (x - mean),  s = w rstd, t = -s rstd B / N
    # (folding the constants into one offset u = -s A / N - t mean makes t x and t mean cancel in fp32: an error of
    # eps |mean| / std per channel that the fused per-replica kernel does not have)
    sc = w * save_rstd
    t = -sc * save_rstd * B / N
    dx = (dy - (A / N).float()) * sc.float() + (x - save_mean.float()) * t.float()
    dweight = ((b - save_mean * a) * save_rstd).float()  # the local part: the gradient all-reduce adds the ranks'
    return dx, dweight, a.float()


def bn_train_fwd(x, weight, bias, eps, momentum, running_mean, running_var, relu, num_batches_tracked=None, use_sync=None):
    """Training-mode BatchNorm1d (+ fused ReLU); running statistics (nullable) updated in place, num_batches_tracked (nullable
    int64 scalar on the device) incremented by the same launches.  use_sync: None = SyncBN when it is on and not suspended
    (ask `sync_bn_active()` first and pass the answer to the backward), False = per-replica statistics.
    Returns (y, save_mean, save_rstd); the saved statistics are float64."""
    lib = _lib.load()
    sync = _sync_bn_group(use_sync)
    if sync is not None:
        return _bn_train_fwd_sync(sync, _need(x, torch.float32, "x"), weight, bias, eps, momentum, running_mean, running_var,
                                  relu, num_batches_tracked)
    x = _need(x, torch.float32, "x")
    n, c = x.shape
    y = torch.empty_like(x)
    stat = torch.empty(2, c, dtype=torch.float64, device=x.device)
    nbytes = lib.pp_bn_train_workspace(n, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.check(lib.pp_bn_train_fwd(_ptr(x), n, c, _ptr(_need(weight, torch.float32, "weight")),
                                   _ptr(_need(bias, torch.float32, "bias")), float(eps), float(momentum),
                                   _ptr(_need(running_mean, torch.float32, "running_mean")),
                                   _ptr(_need(running_var, torch.float32, "running_var")), 1 if relu else 0,
                                   _ptr(y), _ptr(stat[0]), _ptr(stat[1]),
                                   _ptr(_need(num_batches_tracked, torch.int64, "num_batches_tracked")), _ptr(ws), nbytes,
                                   _stream()),
               "pp_bn_train_fwd")
    return y, stat[0], stat[1]


def bn_train_bwd(x, dy, y_relu, weight, save_mean, save_rstd, use_sync=None):
    """Backward of bn_train_fwd: (dx, dweight, dbias); y_relu (the forward output) masks dy when the ReLU was fused.
    use_sync: what the forward of this layer did (a layer's backward all-reduces exactly when its forward did)."""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    dy = _need(dy, torch.float32, "dy")
    sync = _sync_bn_group(use_sync)
    if sync is not None:
        return _bn_train_bwd_sync(sync, x, dy, y_relu, weight, save_mean, save_rstd)
    n, c = x.shape
    dx = torch.empty_like(x)
    dwb = torch.empty(2, c, dtype=torch.float32, device=x.device)
    nbytes = lib.pp_bn_train_workspace(n, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.check(lib.pp_bn_train_bwd(_ptr(x), _ptr(dy), _ptr(_need(y_relu, torch.float32, "y_relu")), n, c,
                                   _ptr(_need(weight, torch.float32, "weight")),
                                   _ptr(_need(save_mean, torch.float64, "save_mean")),
                                   _ptr(_need(save_rstd, torch.float64, "save_rstd")), _ptr(dx), _ptr(dwb[0]),
                                   _ptr(dwb[1]), _ptr(ws), nbytes, _stream()), "pp_bn_train_bwd")
    return dx, dwb[0], dwb[1]


def affine_act(x, scale=None, shift=None, act=0, slope=0.0, residual=None):
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    n, c = x.shape
    y = torch.empty_like(x)
    _lib.check(lib.pp_affine_act(_ptr(x), n, c, _ptr(_need(scale, torch.float32, "scale")),
                                 _ptr(_need(shift, torch.float32, "shift")), int(act), float(slope),
                                 _ptr(_need(residual, torch.float32, "residual")), _ptr(y), _stream()), "pp_affine_act")
    return y


def linear_rows(x, weight, bias=None, transposed=False):
    """y = x W^T + bias (weight [cout, cin]) or, transposed, y = x W (weight [cin, cout] as stored) for skinny layers (<= 32
    channels either side): csrc/pp_dense.hip k_linear_rows instead of a library GEMM."""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    weight = _need(weight, torch.float32, "weight")
    n, cin = x.shape
    cout = weight.shape[1] if transposed else weight.shape[0]
    if (weight.shape[0] if transposed else weight.shape[1]) != cin:
        raise ValueError("linear_rows: weight %s does not fit %d input channels" % (tuple(weight.shape), cin))
    y = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    _lib.check(lib.pp_linear_rows(_ptr(x), _ptr(weight), _ptr(_need(bias, torch.float32, "bias")), n, cin, cout, int(bool(transposed)),
                                  _ptr(y), _stream()), "pp_linear_rows")
    return y


def linear_wgrad(x, dy, want_bias=True):
    """(dW [cout, cin], db [cout] or None) of y = x W^T + b for a skinny layer (cin, cout <= 32): one streaming pass over
    x and dy instead of a split-K GEMM with K = rows (csrc/pp_dense.hip)."""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    dy = _need(dy, torch.float32, "dy")
    n, cin = x.shape
    cout = dy.shape[1]
    dw = torch.empty((cout, cin), dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
    wsb = lib.pp_linear_wgrad_workspace(n, cin, cout)
    ws = _ws(wsb, x.device)
    _lib.check(lib.pp_linear_wgrad(_ptr(x), _ptr(dy), n, cin, cout, _ptr(dw), _ptr(db), _ptr(ws), wsb, _stream()), "pp_linear_wgrad")
    return dw, db


def head_mlp(x, w1, scale, shift, w2, b2, log_softmax=False, want_argmax=False):
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    n, cin = x.shape
    chid, cout = w1.shape[0], w2.shape[0]
    y = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    am = torch.empty(n, dtype=torch.int64, device=x.device) if want_argmax else None
    _lib.check(lib.pp_head_mlp(_ptr(x), n, cin, _ptr(_need(w1, torch.float32, "w1")), chid,
                               _ptr(_need(scale, torch.float32, "scale")), _ptr(_need(shift, torch.float32, "shift")),
                               _ptr(_need(w2, torch.float32, "w2")), _ptr(_need(b2, torch.float32, "b2")), cout,
                               int(bool(log_softmax)), _ptr(y), _ptr(am), _stream()), "pp_head_mlp")
    return (y, am) if want_argmax else y


def heads(x, specs, index=None):
    """All heads of the model in ONE pass over the features (csrc/pp_dense.hip, k_heads).  specs: up to three tuples
    (w1, scale, shift, w2, b2, log_softmax, want_argmax); index (int64 [n], optional): output row i reads x[index[i]] --
    the backbone's features may stay in the coordinate manager's internal order.  Returns [(y, argmax or None), ...]."""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    index = _need(index, torch.int64, "index")
    n = x.shape[0] if index is None else index.shape[0]
    dev = x.device
    flag = _GATHER_ERR.get(dev)
    if flag is None:
        flag = _GATHER_ERR[dev] = _zeros(1, torch.int32, dev)
    arr = (_lib.HeadDesc * len(specs))()
    outs, keep = [], []
    for d, (w1, scale, shift, w2, b2, log_softmax, want_argmax) in zip(arr, specs):
        w1, scale, shift, w2 = (_need(t, torch.float32, "head tensor") for t in (w1, scale, shift, w2))
        b2 = _need(b2, torch.float32, "b2")
        cout = w2.shape[0]
        y = torch.empty((n, cout), dtype=torch.float32, device=dev)
        am = torch.empty(n, dtype=torch.int64, device=dev) if want_argmax else None
        d.w1, d.scale, d.shift, d.w2 = w1.data_ptr(), scale.data_ptr(), shift.data_ptr(), w2.data_ptr()
        d.b2 = None if b2 is None else b2.data_ptr()
        d.y, d.argmax = y.data_ptr(), (None if am is None else am.data_ptr())
        d.cout, d.log_softmax = cout, int(bool(log_softmax))
        outs.append((y, am))
        keep.append((w1, scale, shift, w2, b2))
    _lib.check(lib.pp_heads(_ptr(x), x.shape[0], x.shape[1], _ptr(index), n, C.cast(arr, C.c_void_p), len(specs), _ptr(flag),
                            _stream()), "pp_heads")
    return outs


# ------------------------------------------------------------------------------------------ clustering
class ClusterCSR:
    """Proposals as CSR: offsets int32 [n+1] (device), points int64 [total] (device)."""

    def __init__(self, offsets, points, n):
        self.offsets = offsets
        self.points = points
        self.n = int(n)

    def sizes(self):
        return (self.offsets[1: self.n + 1] - self.offsets[: self.n]).to(torch.int64)

    def select(self, ids):
        """CSR of the proposals `ids` (int64 tensor), in that order."""
        dev = self.offsets.device
        sz = self.sizes()[ids]
        offs = torch.cat([_zeros(1, torch.int64, dev), torch.cumsum(sz, 0)])
        starts = self.offsets[:-1].long()[ids]
        rep = torch.repeat_interleave(torch.arange(ids.numel(), device=dev), sz)
        within = torch.arange(rep.numel(), device=dev) - offs[rep]  # (rep already has the total length: no second host read)
        return ClusterCSR(offs.to(torch.int32), self.points[starts[rep] + within], int(ids.numel()))

    def to_list(self):
        if self.n == 0:
            return []
        return list(torch.split(self.points, self.sizes().tolist()))

    @staticmethod
    def from_list(clusters, device):
        n = len(clusters)
        sizes = torch.tensor([0] + [int(c.numel()) for c in clusters], dtype=torch.int64)
        offsets = torch.cumsum(sizes, 0).to(torch.int32).to(device)
        points = (torch.cat([c.reshape(-1).to(device=device, dtype=torch.int64) for c in clusters]) if n else
                  _zeros(0, torch.int64, device))
        return ClusterCSR(offsets, points.contiguous(), n)

    @staticmethod
    def concat(parts):
        parts = [p for p in parts if p is not None]
        dev = parts[0].offsets.device
        offs = [_zeros(1, torch.int32, dev)]
        base = 0
        for p in parts:
            offs.append(p.offsets[1: p.n + 1] + base)
            base += int(p.points.numel())
        return ClusterCSR(torch.cat(offs), torch.cat([p.points for p in parts]), sum(p.n for p in parts))


class ProposalPairs:
    """Overlapping proposal pairs on the device (pp_proposal_pairs): a[i] < b[i] share inter[i] points, i < n_pairs[0]
    (device counter; the arrays have `capacity` slots, entries are in no particular order).  prop_of_entry int32 [total]
    = proposal of every CSR entry.  info int32[4]: error counters, see `check`."""

    def __init__(self, a, b, inter, n_pairs, capacity, prop_of_entry, info):
        self.a, self.b, self.inter, self.n_pairs, self.capacity = a, b, inter, n_pairs, capacity
        self.prop_of_entry, self.info = prop_of_entry, info

    def check(self, values=None):
        """one host read of the error counters (call where the host synchronises anyway); `values`: the four counters
        when the caller has already read them together with its own numbers"""
        over, full, bad_pt, bad_grp = self.info.tolist() if values is None else values
        if over:
            raise NotImplementedError("%d points belong to more than 8 proposals" % over)
        if full:
            raise _lib.PanopticHipError("proposal pair table overflow (%d): more than %d overlapping pairs" % (full, self.capacity))
        if bad_pt or bad_grp:
            raise _lib.PanopticHipError("proposal points / batch elements out of range (%d / %d)" % (bad_pt, bad_grp))


class UniqueProposals:
    """proposals_unique(): `csr` = the kept proposals (one per distinct point list, in index order), `pos_of` int64 [P] = position
    in `csr` of every original proposal's representative, `batch` int64 / `coords4` int32 [entries, 4] = the scorer's input rows
    (batch index = position of the proposal; (batch, x, y, z))."""

    def __init__(self, csr, pos_of, batch, coords4):
        self.csr, self.pos_of, self.batch, self.coords4 = csr, pos_of, batch, coords4


def proposals_unique(csr, n_points, coords=None):
    """Front end of the proposal scorer (csrc/pp_proposals.hip): duplicates of a point list are scored once.  coords int32
    [n_points, 3] (optional): also emit the (batch, x, y, z) rows of the scorer's input.  ONE host read (the kept counts)."""
    lib = _lib.load()
    dev = csr.offsets.device
    P = csr.n
    offs = _need(csr.offsets, torch.int32, "offsets")
    pts = _need(csr.points, torch.int64, "points")
    rep = torch.empty(max(P, 1), dtype=torch.int64, device=dev)
    pos_of = torch.empty(max(P, 1), dtype=torch.int64, device=dev)
    uoffs = torch.empty(P + 1, dtype=torch.int32, device=dev)
    counts = torch.empty(3, dtype=torch.int32, device=dev)
    wsb = lib.pp_proposals_unique_workspace(P)
    ws = _ws(wsb, dev)
    _lib.check(lib.pp_proposals_unique(_ptr(offs), _ptr(pts), P, int(n_points), _ptr(rep), _ptr(pos_of), _ptr(uoffs), _ptr(counts),
                                       _ptr(ws), wsb, _stream()), "pp_proposals_unique")
    nu, total, bad = counts.tolist()
    if bad:
        raise _lib.PanopticHipError("proposals_unique: %d proposal points outside [0, %d)" % (bad, n_points))
    out_pts = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    out_b = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    c4 = None
    if coords is not None:
        coords = _need(coords, torch.int32, "coords")
        if coords.dim() != 2 or coords.shape[1] != 3 or coords.shape[0] != n_points:
            raise ValueError("coords must be int32 [n_points, 3]")
        c4 = torch.empty((max(total, 1), 4), dtype=torch.int32, device=dev)
    _lib.check(lib.pp_proposals_emit(_ptr(offs), _ptr(pts), P, _ptr(rep), _ptr(pos_of), _ptr(uoffs), _ptr(coords), _ptr(out_pts),
                                     _ptr(out_b), _ptr(c4), _stream()), "pp_proposals_emit")
    return UniqueProposals(ClusterCSR(uoffs[: nu + 1], out_pts[:total], nu), pos_of[:P], out_b[:total],
                           None if c4 is None else c4[:total])


def proposal_pairs(csr, n_points):
    """Pairs of proposals that share points, with their intersection sizes, from the point -> proposal incidence (the
    sparse form of the dense mask @ mask.T of structure_3heads.py:40-60).  No host synchronisation.  Cached on the csr."""
    cached = getattr(csr, "_pairs", None)
    if cached is not None:
        return cached
    lib = _lib.load()
    dev = csr.offsets.device
    P = csr.n
    total = int(csr.points.numel())
    cap = int(lib.pp_proposal_pairs_capacity(P))
    a = torch.empty(cap, dtype=torch.int32, device=dev)
    b = torch.empty(cap, dtype=torch.int32, device=dev)
    inter = torch.empty(cap, dtype=torch.int32, device=dev)
    n_pairs = _zeros(1, torch.int32, dev)
    info = _zeros(4, torch.int32, dev)
    poe = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    wsb = lib.pp_proposal_pairs_workspace(total, int(n_points), P)
    ws = _ws(wsb, dev, tag="proposal_pairs")
    _lib.check(lib.pp_proposal_pairs(_ptr(_need(csr.offsets, torch.int32, "offsets")), _ptr(_need(csr.points, torch.int64, "points")),
                                     P, total, int(n_points), _ptr(poe), _ptr(a), _ptr(b), _ptr(inter), _ptr(n_pairs), _ptr(info),
                                     _ptr(ws), wsb, _stream()), "pp_proposal_pairs")
    csr._pairs = ProposalPairs(a, b, inter, n_pairs, cap, poe[:total], info)
    return csr._pairs


def nms_paint(csr, n_points, batch, n_groups, scores, nms_threshold=0.3, min_cluster_points=10, min_score=0.5, pairs=None):
    """get_instances (NMS, size and score filters; structure_3heads.py:28-71) + get_cur_ins_pre_label (tracker :326-337)
    per batch element on the device: -> (labels int32 [n_points] (-1 = none), counts int32 [n_groups], rank int32 [P]
    (-1 = dropped), ProposalPairs or None).  scores None = no ScoreNet.  No host synchronisation."""
    lib = _lib.load()
    dev = csr.offsets.device
    P = csr.n
    total = int(csr.points.numel())
    labels = torch.empty(max(int(n_points), 1), dtype=torch.int32, device=dev)
    counts = torch.empty(max(int(n_groups), 1), dtype=torch.int32, device=dev)
    rank = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
    if pairs is None and P:
        pairs = proposal_pairs(csr, n_points)
    ev = getattr(pairs, "ready", None)
    if ev is not None:  # the table was built on another stream (PointGroup3heads._pairs_async)
        torch.cuda.current_stream(dev).wait_event(ev)
        pairs.ready = None
    scores = None if scores is None else _need(scores.detach().float(), torch.float32, "scores")
    batch = _need(batch, torch.int64, "batch")
    cap = pairs.capacity if pairs is not None else 0
    wsb = lib.pp_nms_paint_workspace(P, int(n_groups), cap)
    ws = _ws(wsb, dev, tag="nms_paint")
    z = None
    _lib.check(lib.pp_nms_paint(_ptr(csr.offsets), _ptr(csr.points), P, total, int(n_points),
                                _ptr(pairs.prop_of_entry) if pairs else z, _ptr(pairs.a) if pairs else z,
                                _ptr(pairs.b) if pairs else z, _ptr(pairs.inter) if pairs else z,
                                _ptr(pairs.n_pairs) if pairs else z, cap, _ptr(batch), int(n_groups), _ptr(scores),
                                float(nms_threshold), int(min_cluster_points), float(min_score), _ptr(labels), _ptr(counts),
                                _ptr(rank), _ptr(pairs.info) if pairs else z, _ptr(ws), wsb, _stream()), "pp_nms_paint")
    return labels[: int(n_points)], counts[: int(n_groups)], rank[:P], pairs


def histogram2d(a, b, na, nb, allow_skipped=True):
    """int64 [na, nb]: out[a[i], b[i]] += 1 over the rows with a >= 0 and b >= 0 (confusion matrix, instance x class
    tables).  Returns the table on the HOST (numpy): the error counters travel in the same device->host copy, and labels
    outside [0, na) x [0, nb) raise -- the NumPy form's bincount fails on them too, silently dropping them would change
    the metrics.  allow_skipped=False: rows with a < 0 or b < 0 raise as well (a confusion matrix has none)."""
    lib = _lib.load()
    a = _need(a, torch.int64, "a")
    b = _need(b, torch.int64, "b")
    na, nb = int(na), int(nb)
    buf = torch.empty(na * nb + 1, dtype=torch.int64, device=a.device)  # table + {skipped, out of range} as 2 x int32
    info = buf[na * nb:].view(torch.int32)
    _lib.check(lib.pp_histogram2d(_ptr(a), _ptr(b), a.shape[0], na, nb, _ptr(buf), _ptr(info), _stream()), "pp_histogram2d")
    host = buf.cpu().numpy()
    skipped, bad = (int(v) for v in host[na * nb:].view("int32"))
    if bad:
        raise _lib.PanopticHipError("histogram2d: %d labels outside [0, %d) x [0, %d)" % (bad, na, nb))
    if skipped and not allow_skipped:
        raise _lib.PanopticHipError("histogram2d: %d rows with a negative label" % skipped)
    return host[: na * nb].reshape(na, nb)


def pair_counts(a, b, nb, capacity=None):
    """distinct (a[i], b[i]) pairs with a, b >= 0 and their multiplicities: (pair_a, pair_b, count) int64 tensors,
    sorted by (a, b).  One host synchronisation (the number of pairs)."""
    lib = _lib.load()
    a = _need(a, torch.int64, "a")
    b = _need(b, torch.int64, "b")
    dev = a.device
    n = a.shape[0]
    cap = int(capacity or max(4096, min(n, 1 << 22)))
    while True:
        pa = torch.empty(cap, dtype=torch.int64, device=dev)
        pb = torch.empty(cap, dtype=torch.int64, device=dev)
        cnt = torch.empty(cap, dtype=torch.int64, device=dev)
        n_pairs = _zeros(1, torch.int32, dev)
        info = _zeros(2, torch.int32, dev)
        wsb = lib.pp_pair_counts_workspace(cap)
        ws = _ws(wsb, dev, tag="pair_counts")
        _lib.check(lib.pp_pair_counts(_ptr(a), _ptr(b), n, int(nb), cap, _ptr(pa), _ptr(pb), _ptr(cnt), _ptr(n_pairs), _ptr(info),
                                      _ptr(ws), wsb, _stream()), "pp_pair_counts")
        k = int(n_pairs.item())
        over, bad = info.tolist()
        if bad:
            raise _lib.PanopticHipError("pair_counts: %d labels >= nb" % bad)
        if not over:
            break
        cap *= 4
    pa, pb, cnt = pa[:k], pb[:k], cnt[:k]
    order = torch.argsort(pa * int(nb) + pb)
    return pa[order], pb[order], cnt[order]


def block_merge(origin_ids, block_labels, scene_labels, max_instance, state=None):
    """block_merging of the tracker (panoptic_tracker_pointgroup_npm3d.py:339-452) for one cylinder, in place on the
    device: scene_labels int64 [N] (-1 = none), max_instance int64 [1] device tensor carried from block to block.
    Returns the int32[8] state tensor (check with block_merge_check; no synchronisation here)."""
    lib = _lib.load()
    origin_ids = _need(origin_ids, torch.int64, "origin_ids")
    block_labels = _need(block_labels, torch.int32, "block_labels")
    scene_labels = _need(scene_labels, torch.int64, "scene_labels")
    max_instance = _need(max_instance, torch.int64, "max_instance")
    dev = scene_labels.device
    n = origin_ids.shape[0]
    if state is None:
        state = _zeros(8, torch.int32, dev)
    wsb = lib.pp_block_merge_workspace(n)
    ws = _ws(wsb, dev, tag="block_merge")
    _lib.check(lib.pp_block_merge(_ptr(origin_ids), _ptr(block_labels), n, _ptr(scene_labels), scene_labels.shape[0],
                                  _ptr(max_instance), _ptr(state), _ptr(ws), wsb, _stream()), "pp_block_merge")
    return state


def block_merge_check(state):
    st = state.tolist()
    if st[4] or st[5]:
        raise _lib.PanopticHipError("block_merge: table overflow %d, bad ids / labels %d" % (st[4], st[5]))


_NOT_IGNORED_LUT = {}


def not_ignored(labels, ignore_labels, num_classes):
    """bool [n]: labels[i] is none of ignore_labels.  A table lookup (labels lie in [-1, num_classes)): one gather pass where
    torch.isin compares every element with every ignored label and reduces (0.43 ms for the bench scene's 9.8 M points)."""
    # (keyed by the VALUES: a host list costs nothing to read; a device tensor is read once per call, like before)
    key = (tuple(int(v) for v in ignore_labels.tolist()), int(num_classes), labels.device)
    lut = _NOT_IGNORED_LUT.get(key)
    if lut is None:  # built once per (ignore list, device): the boolean-mask indexing below synchronises the stream
        lut = torch.ones(int(num_classes) + 2, dtype=torch.bool, device=labels.device)
        ign = ignore_labels.to(labels.device).long()
        lut[(ign[(ign >= -1) & (ign < num_classes)] + 1)] = False
        if len(_NOT_IGNORED_LUT) > 64:
            _NOT_IGNORED_LUT.clear()
        _NOT_IGNORED_LUT[key] = lut
    return lut[(labels + 1).clamp_(0, int(num_classes) + 1)]


_IGNORE_ON_DEVICE = {}


def region_grow_csr(pos, labels, batch, ignore_labels, nsample, radius, min_cluster_size, num_classes):
    lib = _lib.load()
    pos = _need(pos, torch.float32, "pos")
    labels = _need(labels, torch.int64, "labels")
    batch = _need(batch, torch.int64, "batch")
    dev = pos.device
    n = pos.shape[0]
    # (the ignore list on the device, kept per list: a host tensor's .to(device) is a synchronising copy on every call)
    ikey = (tuple(int(v) for v in ignore_labels.tolist()), str(dev)) if not ignore_labels.is_cuda else None
    ign = _IGNORE_ON_DEVICE.get(ikey) if ikey is not None else None
    if ign is None:
        ign = _need(ignore_labels.to(device=dev, dtype=torch.int64), torch.int64, "ignore_labels")
        if ikey is not None:
            if len(_IGNORE_ON_DEVICE) > 64:
                _IGNORE_ON_DEVICE.clear()
            _IGNORE_ON_DEVICE[ikey] = ign
    pc = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    offs = torch.empty(n + 2, dtype=torch.int32, device=dev)
    pts = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    counts = _zeros(2, torch.int32, dev)
    def run(ws, wsb):
        return lib.pp_region_grow(_ptr(pos), _ptr(labels), _ptr(batch), n, _ptr(ign), ign.numel(), int(num_classes),
                                  int(nsample), float(radius), int(min_cluster_size), _ptr(pc), _ptr(offs), _ptr(pts),
                                  _ptr(counts), _ptr(ws), wsb, _stream())

    # The neighbour lists dominate the workspace and scale with the number of non-ignored points, which the library
    # counts itself before it carves them: try the kept workspace first (steady state: no extra pass, no extra host read)
    # and size it exactly -- one counting pass -- only when the library says it is too small.
    kept = _WS_CACHE.get(("region_grow", str(dev), _stream().value))  # (the key of _ws: tag, device, stream)
    rc = _lib.PP_ERR_WORKSPACE
    if kept is not None and kept.numel() >= lib.pp_region_grow_workspace_for(n, 0, int(nsample)):
        rc = run(kept, kept.numel())
    if rc == _lib.PP_ERR_WORKSPACE:
        n_sel = int(not_ignored(labels, ignore_labels, num_classes).sum().item()) if n else 0
        wsb = lib.pp_region_grow_workspace_for(n, n_sel, int(nsample))
        ws = _ws(max(wsb, 64 << 20), dev, tag="region_grow")
        rc = run(ws, ws.numel())
    _lib.check(rc, "pp_region_grow")
    nc, npts = counts.tolist()
    return ClusterCSR(offs[: nc + 1], pts[:npts], nc), pc[:n]


def meanshift(x, sample_offsets, bandwidth, min_points_exclusive=3, max_iter=300, want_centers=False):
    """x [m,dim] float32 (points of a sample contiguous); sample_offsets: python list / CPU tensor [ns+1]."""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    m, dim = x.shape
    dev = x.device
    so = [int(v) for v in (sample_offsets.tolist() if torch.is_tensor(sample_offsets) else sample_offsets)]
    ns = len(so) - 1
    so_arr = (C.c_int64 * (ns + 1))(*so)
    labels = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    ncl = torch.zeros(max(ns, 1), dtype=torch.int32, device=dev)
    centers = torch.zeros((max(m, 1), dim), dtype=torch.float32, device=dev) if want_centers else None
    wsb = lib.pp_meanshift_workspace(m, dim, ns)
    ws = _ws(wsb, dev, tag="meanshift")
    _lib.check(lib.pp_meanshift(_ptr(x), m, dim, C.cast(so_arr, C.c_void_p), ns, float(bandwidth),
                                int(min_points_exclusive), int(max_iter), _ptr(labels), _ptr(ncl), _ptr(centers), _ptr(ws),
                                wsb, _stream()), "pp_meanshift")
    return labels[:m], ncl[:ns], centers


def hdbscan(x, sample_offsets, min_cluster_size=15, min_samples=5, cluster_selection_epsilon=0.006, count_self=False,
            min_points_exclusive=3):
    """x [m,dim] float32 (points of a sample contiguous); returns (labels int32 [m], clusters per sample int32 [ns])."""
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    m, dim = x.shape
    dev = x.device
    so = [int(v) for v in (sample_offsets.tolist() if torch.is_tensor(sample_offsets) else sample_offsets)]
    ns = len(so) - 1
    so_arr = (C.c_int64 * (ns + 1))(*so)
    labels = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    ncl = torch.zeros(max(ns, 1), dtype=torch.int32, device=dev)
    wsb = lib.pp_hdbscan_workspace(m, ns)
    ws = _ws(wsb, dev, tag="hdbscan")
    _lib.check(lib.pp_hdbscan(_ptr(x), m, dim, C.cast(so_arr, C.c_void_p), ns, int(min_points_exclusive),
                              int(min_cluster_size), int(min_samples), 1 if count_self else 0,
                              float(cluster_selection_epsilon), _ptr(labels), _ptr(ncl), _ptr(ws), wsb, _stream()),
               "pp_hdbscan")
    return labels[:m], ncl[:ns]


def voxelize(pos, voxel_size, batch=None):
    """GridSampling3D(quantize_coords=True) on the GPU: (coords int32 [V,4] (b,x,y,z), rep_index int64 [V], inverse int64 [n])."""
    lib = _lib.load()
    pos = _need(pos, torch.float32, "pos")
    batch = _need(batch, torch.int64, "batch")
    n = pos.shape[0]
    dev = pos.device
    coords = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
    rep = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    inv = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    counts = _zeros(2, torch.int32, dev)
    wsb = lib.pp_voxelize_workspace(n)
    ws = _ws(wsb, dev, tag="voxelize")
    _lib.check(lib.pp_voxelize(_ptr(pos), _ptr(batch), n, float(voxel_size), _ptr(coords), _ptr(rep), _ptr(inv), _ptr(counts),
                               _ptr(ws), wsb, _stream()), "pp_voxelize")
    nv, bad = counts.tolist()
    if bad:
        raise _lib.PanopticHipError("%d points outside the +-32767-voxel coordinate range" % bad)
    return coords[:nv], rep[:nv].long(), inv[:n].long()


def cylinder_tiles(pos, centres_xy, radius):
    """CylinderSampling for all centres at once -> ClusterCSR (one ascending index list per cylinder)."""
    lib = _lib.load()
    pos = _need(pos, torch.float32, "pos")
    cen = _need(centres_xy, torch.float32, "centres_xy")
    n, nc = pos.shape[0], cen.shape[0]
    dev = pos.device
    n_pairs = _zeros(1, torch.int32, dev)
    wsb = lib.pp_cylinder_pairs_workspace(n)
    ws = _ws(wsb, dev, tag="cylinders")
    _lib.check(lib.pp_cylinder_pairs(_ptr(pos), n, _ptr(cen), nc, float(radius), None, None, 0, _ptr(n_pairs), _ptr(ws), wsb,
                                     _stream()), "pp_cylinder_pairs")
    total = int(n_pairs.item())
    pp = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    pc = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    _lib.check(lib.pp_cylinder_pairs(_ptr(pos), n, _ptr(cen), nc, float(radius), _ptr(pp), _ptr(pc), total, _ptr(n_pairs), _ptr(ws),
                                     wsb, _stream()), "pp_cylinder_pairs")
    offs, out, tot = group_by_key(pc[:total].contiguous(), nc, ids=pp[:total].contiguous())
    group_by_key_check(tot)
    return ClusterCSR(offs, out[:total], nc)


_GATHER_ERR = {}


def gather_rows(src, index):
    """src[index] for [n,c] float32 rows with c % 4 == 0 and an int64 index (other layouts: torch indexing).  Indices
    are trusted the way torch trusts them on the device: an out-of-range index is counted in a device flag that
    `gather_rows_check()` turns into an exception (no synchronisation on the hot path)."""
    if (src.dtype != torch.float32 or src.dim() != 2 or src.shape[1] % 4 or index.dtype != torch.int64
            or index.dim() != 1 or not src.is_contiguous()):
        return src[index]
    lib = _lib.load()
    index = _need(index, torch.int64, "index")
    dev = src.device
    flag = _GATHER_ERR.get(dev)
    if flag is None:
        flag = _GATHER_ERR[dev] = _zeros(1, torch.int32, dev)
    out = torch.empty((index.shape[0], src.shape[1]), dtype=torch.float32, device=dev)
    _lib.check(lib.pp_gather_rows(_ptr(src), src.shape[0], src.shape[1], _ptr(index), index.shape[0], _ptr(out),
                                  _ptr(flag), _stream()), "pp_gather_rows")
    return out


def gather_rows_flag(device):
    """the device counter behind gather_rows_check (int32 [1]) or None: for callers that read it with their own numbers"""
    return _GATHER_ERR.get(device)


def gather_rows_check(values=None):
    """raises if any gather_rows call since the last check saw an out-of-range index (one synchronisation); `values`:
    {device: count} already read by the caller"""
    for dev, flag in _GATHER_ERR.items():
        bad = int(flag.item()) if values is None or dev not in values else int(values[dev])
        if bad:
            flag.zero_()
            raise _lib.PanopticHipError("gather_rows: %d indices out of range" % bad)


def nearest(ref, query, cell, max_dist=0.0):
    """Exact 1-NN of every query row among the reference rows ([n,2] or [n,3] float32): (idx int64, dist2 float32).
    idx -1 / dist2 inf where no reference point lies within max_dist (> 0); ties -> smallest reference index."""
    lib = _lib.load()
    ref = _need(ref, torch.float32, "ref")
    query = _need(query, torch.float32, "query")
    if ref.dim() != 2 or query.dim() != 2 or ref.shape[1] != query.shape[1]:
        raise ValueError("nearest: ref and query must be [n,dim] with the same dim")
    nq, dim = query.shape
    dev = query.device
    idx = torch.empty(nq, dtype=torch.int64, device=dev)
    d2 = torch.empty(nq, dtype=torch.float32, device=dev)
    wsb = lib.pp_nearest_workspace(ref.shape[0])
    ws = _ws(wsb, dev, tag="nearest")
    _lib.check(lib.pp_nearest(_ptr(ref), ref.shape[0], _ptr(query), nq, dim, float(cell), float(max_dist), _ptr(idx),
                              _ptr(d2), _ptr(ws), wsb, _stream()), "pp_nearest")
    return idx, d2


def group_by_key(key, n_groups, ids=None):
    lib = _lib.load()
    key = _need(key, torch.int32, "key")
    n = key.shape[0]
    dev = key.device
    ids = _need(ids, torch.int64, "ids")
    offs = torch.empty(n_groups + 1, dtype=torch.int32, device=dev)
    out = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    total = _zeros(2, torch.int32, dev)  # [kept, keys >= n_groups (caller error, see group_by_key_check)]
    wsb = lib.pp_group_by_key_workspace(max(n, n_groups))
    ws = _ws(wsb, dev)
    _lib.check(lib.pp_group_by_key(_ptr(key), _ptr(ids), n, int(n_groups), _ptr(offs), _ptr(out), _ptr(total),
                                   _ptr(total[1:]), _ptr(ws), wsb, _stream()), "pp_group_by_key")
    return offs, out, total


def group_by_key_check(total):
    """`total` as returned by group_by_key, read on the host (one synchronisation, callers that need the count anyway):
    -> number of grouped entries; raises if any key was >= n_groups."""
    kept, bad = total.tolist()
    if bad:
        raise _lib.PanopticHipError("group_by_key: %d keys outside [0, n_groups)" % bad)
    return kept


# sums / means by segment without float atomics: rows grouped by segment (stable), one workgroup per segment adding in a fixed
# order -- bit-reproducible run to run (PP_SEGMENT_DETERMINISTIC=0: the atomic kernels; maxima are order-free either way)
SEGMENT_DETERMINISTIC = os.environ.get("PP_SEGMENT_DETERMINISTIC", "1") != "0"


class _GroupingCache(threading.local):
    """(offsets, rows) of the last few segment-id tensors: a loss uses the same ids for several sums and for the backward of
    its gathers (the discriminative loss: four times), and the grouping -- a radix sort -- is most of an ordered sum's cost.
    Entries hold the id tensor itself (address + shape + version counter decide a hit; views made by reshape share all three), so
    its storage stays alive and a recycled address cannot alias."""

    def __init__(self):
        self.items = []


_GROUPINGS = _GroupingCache()


def clear_groupings():
    """drop the calling thread's cached groupings (training.train_step calls this at every step boundary: the entries pin
    multi-million-row id tensors, and an id buffer rewritten through a raw pointer keeps its version counter)"""
    _GROUPINGS.items.clear()


def _grouping_of(index, n_seg):
    for it in _GROUPINGS.items:
        if it[0].data_ptr() == index.data_ptr() and it[0].shape == index.shape and it[0].stride() == index.stride() \
                and it[0].dtype == index.dtype and it[1] == index._version and it[2] == n_seg:
            return it[3]
    g = group_by_key(index.to(torch.int32), n_seg, ids=None)
    _GROUPINGS.items.append((index, index._version, n_seg, g))
    if len(_GROUPINGS.items) > 8:
        del _GROUPINGS.items[0]
    return g


def segment_reduce(src, index, n_seg, reduce, want_arg=False, check=True):
    """check=False: the ids are known to be in range (no validation, no stream synchronisation)"""
    lib = _lib.load()
    src = _need(src, torch.float32, "src")
    index = _need(index, torch.int64, "index")
    n, c = src.shape
    dev = src.device
    out = torch.empty((n_seg, c), dtype=torch.float32, device=dev)
    if SEGMENT_DETERMINISTIC and _REDUCE[reduce] != 2 and not want_arg and n_seg < (1 << 31) and n < (1 << 31):
        offs, rows, total = _grouping_of(index, n_seg)
        if check:
            kept, bad = total.tolist()
            if kept != n:
                raise _lib.PanopticHipError("segment_reduce: %d segment ids outside [0, %d)" % (n - kept, n_seg))
        _lib.check(lib.pp_segment_sum_ordered(_ptr(src), _ptr(rows), _ptr(offs), int(n_seg), c, 1 if _REDUCE[reduce] == 1 else 0,
                                              _ptr(out), _stream()), "pp_segment_sum_ordered")
        return out
    arg = torch.empty((n_seg, c), dtype=torch.int64, device=dev) if want_arg else None
    wsb = lib.pp_segment_reduce_workspace(n_seg)
    ws = _ws(wsb, dev)
    fn = lib.pp_segment_reduce if check else lib.pp_segment_reduce_unchecked
    _lib.check(fn(_ptr(src), _ptr(index), n, c, int(n_seg), _REDUCE[reduce], _ptr(out), _ptr(arg), _ptr(ws), wsb, _stream()),
               "pp_segment_reduce")
    return (out, arg) if want_arg else out


def instance_iou_csr(csr, gt_instances, batch, gt_offsets, gt_sizes):
    lib = _lib.load()
    dev = gt_instances.device
    total_gt = int(gt_sizes.numel())
    iou = torch.zeros((csr.n, max(total_gt, 1)), dtype=torch.float32, device=dev)
    if csr.n and total_gt:
        iou = torch.empty((csr.n, total_gt), dtype=torch.float32, device=dev)
        _lib.check(lib.pp_instance_iou(_ptr(csr.offsets), _ptr(csr.points), csr.n,
                                       _ptr(_need(gt_instances, torch.int64, "gt_instances")),
                                       _ptr(_need(batch, torch.int64, "batch")),
                                       _ptr(_need(gt_offsets, torch.int32, "gt_offsets")),
                                       _ptr(_need(gt_sizes, torch.int32, "gt_sizes")), total_gt, _ptr(iou), _stream()),
                   "pp_instance_iou")
    return iou[:, :total_gt]


def proposal_intersections(csr, n_points):
    lib = _lib.load()
    dev = csr.offsets.device
    inter = torch.zeros((csr.n, csr.n), dtype=torch.int32, device=dev)
    if csr.n == 0:
        return inter
    total = int(csr.points.numel())
    wsb = lib.pp_proposal_intersections_workspace(total, int(n_points))
    ws = _ws(wsb, dev)
    _lib.check(lib.pp_proposal_intersections(_ptr(csr.offsets), _ptr(csr.points), csr.n, int(n_points), _ptr(inter),
                                             _ptr(ws), wsb, _stream()), "pp_proposal_intersections")
    return inter
