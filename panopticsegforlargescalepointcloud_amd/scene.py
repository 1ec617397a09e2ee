"""Scene-level driver: cylinder tiles -> model -> per-tile instance labels -> (multi-GPU exchange) -> scene assembly.

Follows the reference's eval path: per cylinder `forward`, `get_instances` (NMS 0.3, score > 0.5,
size > 10), `get_cur_ins_pre_label` (clusters painted in ascending score order so the best score wins), semantic
vote accumulation and the ORDER-DEPENDENT greedy `block_merging`
(torch_points3d/metrics/panoptic_tracker_pointgroup_npm3d.py:147-277,326-337,339-452).

Multi-GPU: tiles are sharded over ranks (longest-first round robin); each rank runs its tiles with no
collective on the data path; ONE exchange step all-gathers the per-tile label arrays, origin ids and semantic vote
contributions (`exchange_tile_results`) so the merge can run in the original block order on every rank.  torch.distributed backend "nccl" is RCCL
on ROCm; the same code runs on gloo/CPU tensors for the world_size-2 tests.
"""
import os

import numpy as np
import torch

from .applications import Data


# ------------------------------------------------------------------------------------------------ per-tile post-processing
from . import ops  # noqa: E402

# TileRunner.run(next_batch=...): build the next batch's coordinate manager during the current batch (PP_INPUT_PREFETCH=0: off)
INPUT_PREFETCH = os.environ.get("PP_INPUT_PREFETCH", "1") != "0"
# TileRunner(backbone_ahead=True): the proposal scorer's convolutions wait for the backbone that runs ahead (PP_AHEAD_SCORER_WAIT=0,
# A/B runs: both convolution streams at once)
SCORER_WAITS = os.environ.get("PP_AHEAD_SCORER_WAIT", "1") != "0"


def instance_labels_per_tile(res, batch, n_tiles, nms_threshold=0.3, min_cluster_points=10, min_score=0.5):
    """get_instances per batch element (NMS, size and score filters; structure_3heads.py:28-71) and
    get_cur_ins_pre_label (surviving clusters painted in ascending score order; tracker :326-337), for all tiles of the
    batch at once and entirely on the device (ops.nms_paint: incidence pass -> overlapping pairs -> per-tile greedy NMS
    -> scatter-max painting).  The only host read is the per-tile instance counts.
    Returns int32 labels [N] (-1 = none; ids restart at 0 in every tile) and the number of instances per tile."""
    n = batch.shape[0]
    csr = res.clusters_csr
    if csr is not None and csr.n and res.mask_scores is not None:
        from .panoptic.structures import masked_csr
        csr = masked_csr(csr, res.mask_scores)  # mask-supervised scorer: points with mask logit <= -0.5 leave their proposal
    if csr is None or csr.n == 0:
        ops.gather_rows_check()
        return torch.full((n,), -1, dtype=torch.int32, device=batch.device), [0] * n_tiles
    labels, counts, _, pairs = ops.nms_paint(csr, n, batch, n_tiles, res.cluster_scores, nms_threshold, min_cluster_points,
                                             min_score)
    # the synchronisation point of the step: ONE host read brings the per-tile counts, the pair table's error counters and
    # the row-gather index check (three reads in a row used to leave the GPU idle between two steps)
    flag = ops.gather_rows_flag(batch.device)
    parts = [counts.to(torch.int32).view(-1), pairs.info.view(-1)] + ([flag.view(-1)] if flag is not None else [])
    vals = torch.cat(parts).tolist()
    nt = counts.numel()
    pairs.check(vals[nt: nt + 4])
    if flag is not None:
        ops.gather_rows_check({batch.device: vals[nt + 4]})
    return labels, vals[:nt]


# ------------------------------------------------------------------------------------------------ scene assembly
def block_merging(originids, pre_ins, all_pre_ins, max_instance):
    """Greedy merge of one block's instance labels into the scene labels (reference :339-452 with
    origin_sub_ids == originids, i.e. the 1-NN back-projection is the identity).  NumPy, in place on a copy."""
    all_pre_ins = all_pre_ins.copy()
    if not np.any(pre_ins != -1):
        return all_pre_ins, max_instance
    t_num = int(np.max(pre_ins)) + 1
    cur = all_pre_ins[originids]
    has, none = cur != -1, cur == -1
    if not has.any():
        valid = pre_ins != -1
        all_pre_ins[originids[valid]] = pre_ins[valid] + max_instance
        return all_pre_ins, max_instance + t_num
    if not none.any():
        return all_pre_ins, max_instance
    for ii in range(t_num):
        pts = originids[pre_ins == ii]
        old = all_pre_ins[pts]
        has_old = pts[old != -1]
        not_old = pts[old == -1]
        if len(has_old) == 0:
            all_pre_ins[not_old] = max_instance + 1
            max_instance += 1
        elif len(not_old) == 0:
            continue
        else:
            best_iou, best_label = 0.0, 0
            for g in np.unique(all_pre_ins[has_old]):
                idx_old_all = originids[all_pre_ins[originids] == g]
                inter = np.intersect1d(idx_old_all, pts).size
                union = np.union1d(idx_old_all, pts).size
                iou = float(inter) / float(union)
                if iou > best_iou:
                    best_iou, best_label = iou, g
            if best_iou > 0.1:  # hard-coded in the reference (:447)
                all_pre_ins[not_old] = best_label
            else:
                all_pre_ins[not_old] = max_instance + 1
                max_instance += 1
    return all_pre_ins, max_instance


class SceneAssembler:
    """votes[origin] += semantic log-probs, prediction_count[origin] += 1 (reference :244-245) and block merging
    in block order."""

    def __init__(self, n_scene_points, num_classes):
        self.votes = np.zeros((n_scene_points, num_classes), np.float32)
        self.prediction_count = np.zeros(n_scene_points, np.int32)
        self.ins_pre = np.full(n_scene_points, -1, np.int64)
        self.max_instance = 0

    def add_block(self, origin_ids, labels, semantic_logits=None):
        if semantic_logits is not None:
            np.add.at(self.votes, origin_ids, semantic_logits)
        np.add.at(self.prediction_count, origin_ids, 1)
        self.ins_pre, self.max_instance = block_merging(origin_ids, labels.astype(np.int64), self.ins_pre,
                                                        self.max_instance)

    def semantic_prediction(self):
        return self.votes.argmax(1)


class SceneAssemblerGPU:
    """SceneAssembler with every per-point array resident on the device: votes / prediction counts are index_add_'s, the
    order-dependent block merging is ops.block_merge (csrc/pp_eval.hip) chained on the stream -- no host round trip per
    block; `finish()` reads the error counters once.  Results equal SceneAssembler's bit for bit PROVIDED the origin ids of
    a block are distinct (a cylinder's points are distinct scene points; `finish()` checks it): both forms ACCUMULATE
    repeated ids (np.add.at / index_add_), but the device adds them with float atomics in no fixed order."""

    def __init__(self, n_scene_points, num_classes, device):
        self.votes = torch.zeros((n_scene_points, num_classes), dtype=torch.float32, device=device)
        self.prediction_count = torch.zeros(n_scene_points, dtype=torch.int32, device=device)
        self.ins_pre = torch.full((n_scene_points,), -1, dtype=torch.int64, device=device)
        self._max_instance = torch.zeros(1, dtype=torch.int64, device=device)
        self._states = []
        self._dup = torch.zeros((), dtype=torch.bool, device=device)

    def add_block(self, origin_ids, labels, semantic_logits=None):
        origin_ids = origin_ids.to(self.ins_pre.device).long()
        if semantic_logits is not None:
            self.votes.index_add_(0, origin_ids, semantic_logits.to(self.votes.device).float())
        before = self.prediction_count[origin_ids]
        self.prediction_count.index_add_(0, origin_ids, torch.ones_like(origin_ids, dtype=torch.int32))
        # a repeated id inside the block raises its count by more than one (device flag, read once in finish())
        self._dup = self._dup | ((self.prediction_count[origin_ids] - before) != 1).any()
        self._states.append(ops.block_merge(origin_ids, labels.to(self.ins_pre.device).to(torch.int32), self.ins_pre,
                                            self._max_instance))

    def finish(self):
        for st in self._states:
            ops.block_merge_check(st)
        self._states = []
        if bool(self._dup):
            raise ValueError("SceneAssemblerGPU: a block repeated an origin id (votes would depend on the atomic order)")
        return self

    @property
    def max_instance(self):
        return int(self._max_instance.item())

    def semantic_prediction(self):
        return self.votes.argmax(1)


# ------------------------------------------------------------------------------------------------ sharding / exchange
def shard_tiles(tile_sizes, world_size):
    """Static longest-first round robin. Returns list (per rank) of tile ids, ascending within a rank."""
    order = np.argsort(-np.asarray(tile_sizes), kind="stable")
    shards = [[] for _ in range(world_size)]
    for j, t in enumerate(order.tolist()):
        shards[j % world_size].append(t)
    return [sorted(s) for s in shards]


def allgather_varlen(t, group=None):
    """all_gather of 1-D / 2-D tensors whose first dimension differs per rank: ONE all-gather of the lengths and ONE
    of padded buffers (a single large collective per scene instead of one per cylinder)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:s] for b, s in zip(bufs, sizes)]


def exchange_tile_results(local, group=None, device=None):
    """local: dict tile_id -> (origin_ids int64 [n], labels int32 [n]) or (origin_ids, labels, semantic log-probs f32 [n,C])
    on this rank (tensors on one device).  Returns the same dict for ALL tiles on every rank -- what the tracker's scene
    assembly needs from every cylinder: the semantic vote contributions `votes[origin] += logits` and the per-cylinder
    instance labels for the order-dependent block merging (panoptic_tracker_pointgroup_npm3d.py:244-245,277).
    Two collectives per scene: a small header and ONE all-gather of a padded int32 [rows, 2 + C] buffer (origin id, label,
    C float32 log-probs bit-cast to int32: 8 + 4C bytes per point; scenes have fewer than 2^31 points -- 64-bit ids fall back
    to an int64 buffer plus a separate float buffer).  `device`: where the collective's buffers live when this rank owns
    no tile (nccl/RCCL needs device tensors even for an empty contribution); default = the current HIP device under nccl."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(local)
    world = dist.get_world_size(group)
    ids = sorted(local)
    gloo = dist.get_backend(group) == "gloo"
    if gloo:  # CPU collectives (tests / single-GPU dry runs)
        local = {t: tuple(x.cpu() for x in local[t]) for t in ids}
        dev = torch.device("cpu")
    elif ids:
        dev = local[ids[0]][0].device
    elif device is not None:
        dev = torch.device(device)
    else:  # a rank without tiles still joins the collectives, with device buffers
        dev = torch.device("cuda", torch.cuda.current_device())
    n_cls = max([local[t][2].shape[1] for t in ids if len(local[t]) > 2] + [0])
    meta = torch.tensor([[t, local[t][0].shape[0]] for t in ids], dtype=torch.int64, device=dev).reshape(-1, 2)
    origin = torch.cat([local[t][0] for t in ids]) if ids else torch.zeros(0, dtype=torch.int64, device=dev)
    labels = torch.cat([local[t][1] for t in ids]) if ids else torch.zeros(0, dtype=torch.int32, device=dev)
    # header: [rows, largest origin id, tiles, classes] of this rank, then the (tile, rows) pairs
    head = torch.tensor([origin.shape[0], int(origin.max().item()) if origin.numel() else 0, len(ids), n_cls],
                        dtype=torch.int64, device=dev)
    heads = torch.empty((world, 4), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(heads.view(-1), head, group=group)
    heads = heads.tolist()
    max_rows = max(max(h[0] for h in heads), 1)
    max_tiles = max(max(h[2] for h in heads), 1)
    n_cls = max(h[3] for h in heads)
    wide = max(h[1] for h in heads) >= 2 ** 31
    dt = torch.int64 if wide else torch.int32
    logits = None
    if n_cls:
        logits = torch.zeros((origin.shape[0], n_cls), dtype=torch.float32, device=dev)
        pos = 0
        for t in ids:
            n_t = local[t][0].shape[0]
            if len(local[t]) > 2:
                logits[pos: pos + n_t] = local[t][2].float()
            pos += n_t
    pack_logits = n_cls > 0 and not wide
    width = 2 + (n_cls if pack_logits else 0)
    # one buffer per rank: max_tiles (tile, rows) pairs followed by max_rows (origin, label[, log-probs]) rows
    send = torch.zeros((max_tiles + max_rows, width), dtype=dt, device=dev)
    send[: len(ids), :2] = meta.to(dt)
    send[max_tiles: max_tiles + origin.shape[0], 0] = origin.to(dt)
    send[max_tiles: max_tiles + origin.shape[0], 1] = labels.to(dt)
    if pack_logits:
        send[max_tiles: max_tiles + origin.shape[0], 2:] = logits.view(torch.int32)
    recv = torch.empty((world,) + tuple(send.shape), dtype=dt, device=dev)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    all_logits = None
    if pack_logits:
        all_logits = recv[:, max_tiles:, 2:].contiguous().view(torch.float32)
    elif n_cls:  # 64-bit ids: the float payload travels on its own
        fsend = torch.zeros((max_rows, n_cls), dtype=torch.float32, device=dev)
        fsend[: origin.shape[0]] = logits
        all_logits = torch.empty((world, max_rows, n_cls), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(all_logits.view(-1), fsend.view(-1), group=group)
    metas = recv[:, :max_tiles, :2].tolist()
    all_origin = recv[:, max_tiles:, 0].long()   # one conversion for the scene, per-tile results are views
    all_labels = recv[:, max_tiles:, 1].to(torch.int32)
    out = {}
    for r in range(world):
        pos = 0
        for t, n in metas[r][: heads[r][2]]:
            item = (all_origin[r, pos: pos + n], all_labels[r, pos: pos + n])
            if all_logits is not None:
                item = item + (all_logits[r, pos: pos + n],)
            out[int(t)] = item
            pos += n
    return out


def assemble_scene(results, tile_order, n_scene_points, num_classes):
    """Scene assembly from the exchanged per-tile results, in the ORIGINAL block order (the greedy merge is
    order-dependent): semantic votes + prediction counts + merged instance labels.  Runs identically on every rank."""
    asm = SceneAssembler(n_scene_points, num_classes)
    for t in tile_order:
        item = results[t]
        origin = item[0].cpu().numpy()
        asm.add_block(origin, item[1].cpu().numpy(), item[2].cpu().numpy() if len(item) > 2 else None)
    return asm


# ------------------------------------------------------------------------------------------------ the hot path over tiles
class _BackboneAhead:
    """what stage A (backbone + heads) of a batch leaves for its stage B (grouping, scorer, NMS)"""
    __slots__ = ("key", "data", "outs", "done", "hold", "thread", "err")


def _batch_key(b):
    return (b["coords"].data_ptr() if torch.is_tensor(b["coords"]) else id(b["coords"]), int(b["coords"].shape[0]))


class TileRunner:
    """Runs batches of cylinder tiles through the model on one device (eval mode, no_grad).

    backbone_ahead=True (a stream of batches, `run(..., next_batch=, after_next=)`): the NEXT batch's backbone + heads are launched
    on a stream of their own before this batch's grouping starts, so that the convolutions of batch i + 1 fill the GPU while
    batch i is in region growing / mean shift / the proposal scorer's front end (latency- and atomics-bound kernels and host reads
    that leave most of the chip idle, profiles/r06_kt/timeline.txt); the scorer's own convolutions wait for that backbone (two
    saturating convolution streams only take turns).  Results are those of the one-batch-at-a-time order bit for bit
    (tests/test_scene_gpu.py::test_backbone_ahead_does_not_change_results); the batch after next gets its coordinate manager
    built meanwhile (`after_next`), as `next_batch` does without the option.
    ahead_thread=True (PP_AHEAD_THREAD): the backbone ahead is ENQUEUED by a host thread of its own as well -- its ~250 launches
    (a few ms of Python) overlap this batch's host reads instead of preceding them."""

    def __init__(self, model, device, epoch=10 ** 6, stage_timing=False, backbone_ahead=False, ahead_thread=None):
        self.model, self.device, self.epoch = model, device, epoch
        self.stage_timing = stage_timing
        self.stage_ms = {}
        self.backbone_ahead = bool(backbone_ahead) and not stage_timing
        self.ahead_thread = (os.environ.get("PP_AHEAD_THREAD", "0") == "1") if ahead_thread is None else bool(ahead_thread)
        self._ahead = None
        # PP_STREAM_ORDER (A/B runs): first-use order of the model's streams, letters S (map prefetch), P / Q (the two preparation
        # streams), C (clustering / NMS side stream) -- e.g. "SCPQ"; unset: whatever order the first pass uses them in (P, Q, C, S)
        order = os.environ.get("PP_STREAM_ORDER", "")
        if order and torch.cuda.is_available() and torch.device(device).type == "cuda":
            from . import MinkowskiEngine as _ME
            for ch in order:
                if ch == "S":
                    st = _ME._side_stream(torch.device(device))
                elif ch in "PQ":
                    st = _ME._prep_stream(torch.device(device), 0 if ch == "P" else 1)
                elif ch == "C":
                    st = model._side_streams.get(torch.device(device)) if hasattr(model, "_side_streams") else None
                    if st is None and hasattr(model, "_side_streams"):
                        st = model._side_streams[torch.device(device)] = torch.cuda.Stream(device=device)
                else:
                    st = None
                if st is not None:
                    with torch.cuda.stream(st):
                        torch.zeros(1, device=device)
        # The stream the backbone ahead runs on is created AND USED here, before the model's first pass uses its side / preparation /
        # clustering streams: a process' HIP streams share a few hardware queues, bound in order of first use, and a stream first used
        # late landed on a queue where the two batches ran one after the other (profiles/r06_backbone_ahead.txt: 108.6 ms per step
        # when the option was switched on after a few serial steps, 104.2 from the start, 105.4 either way with this first submission).
        # Only with the option: the extra stream shifts the other streams' queues (PP_AHEAD_PRIORITY: its HIP priority, A/B runs)
        self._ahead_stream = None
        if self.backbone_ahead and torch.cuda.is_available() and torch.device(device).type == "cuda":
            self._ahead_stream = torch.cuda.Stream(device=device, priority=int(os.environ.get("PP_AHEAD_PRIORITY", "0")))
            with torch.cuda.stream(self._ahead_stream):
                torch.zeros(1, device=device)

    def _tick(self, name, t0):
        """wall time of a stage incl. its device work (only when stage_timing is on: it synchronises)."""
        if self.stage_timing:
            import time
            torch.cuda.synchronize()
            now = time.perf_counter()
            self.stage_ms[name] = self.stage_ms.get(name, 0.0) + 1e3 * (now - t0)
            return now
        return t0

    def _data(self, batch_np):
        dev = self.device
        to = lambda a: a.to(dev) if torch.is_tensor(a) else torch.from_numpy(a).to(dev)  # noqa: E731
        return Data(pos=to(batch_np["pos"]), coords=to(batch_np["coords"]), batch=to(batch_np["batch"]), x=to(batch_np["x"]))

    def _stage_a(self, batch_np):
        """backbone + heads of a batch on the current stream, WITHOUT touching the model's per-batch attributes (`input`, `raw_pos`,
        `labels` may still belong to the batch another stage is grouping): `set_input(record.data)` installs them for stage B"""
        c = _BackboneAhead()
        c.key, c.data = _batch_key(batch_np), self._data(batch_np).to(self.device)
        c.outs = self.model.backbone_and_heads(c.data)
        c.done = c.hold = c.thread = c.err = None
        return c

    def _launch_ahead(self, next_batch, after_next, main):
        """stage A of the next batch on the ahead stream (from this thread, or from one of its own), then the preparation of the
        batch after it"""
        sa = self._ahead_stream
        if sa is None:
            sa = self._ahead_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("PP_AHEAD_PRIORITY", "0")))
        sa.wait_stream(main)  # (the batch tensors and everything this stream has freed so far)
        rec = _BackboneAhead()
        rec.key, rec.data, rec.outs, rec.hold, rec.err = _batch_key(next_batch), None, None, None, None
        rec.done = torch.cuda.Event()
        dev = self.device

        def work():
            try:
                with torch.no_grad(), torch.cuda.device(dev):   # (grad mode and current device are per thread)
                    with torch.cuda.stream(sa):
                        c = self._stage_a(next_batch)
                        rec.done.record(sa)
                    rec.data, rec.outs = c.data, c.outs
                    f = c.outs[0]
                    rec.hold = [t for t in ((f.base, f.index) if hasattr(f, "base") else (f,)) + tuple(c.outs[1:]) if torch.is_tensor(t)]
                    if after_next is not None:
                        self.model.Backbone.prepare_input(Data(coords=after_next["coords"], batch=after_next["batch"]))
            except BaseException as e:  # (re-raised by the thread that joins)
                rec.err = e

        if self.ahead_thread:
            import threading
            rec.thread = threading.Thread(target=work, name="pp-backbone-ahead", daemon=True)
            rec.thread.start()
        else:
            rec.thread = None
            work()
            if rec.err is not None:
                raise rec.err
        return rec

    @staticmethod
    def _join(rec):
        if rec.thread is not None:
            rec.thread.join()
            rec.thread = None
        if rec.err is not None:
            err, rec.err = rec.err, None
            raise err

    def drain(self):
        """wait for (and drop) a backbone that was launched ahead and never consumed"""
        a, self._ahead = self._ahead, None
        if a is not None:
            self._join(a)
            a.done.synchronize()

    @torch.no_grad()
    def run(self, batch_np, n_tiles, override=None, next_batch=None, after_next=None):
        """batch_np: dict of numpy/torch arrays (pos, coords, batch, x, origin_id). override: optional
        (pred int64 [N], offsets [N,3], embeddings [N,D]) device tensors replacing the heads' outputs for grouping.
        next_batch: the batch the NEXT call will be given (device tensors): its coordinate manager -- Morton order, block index,
        level chain and kernel maps of the backbone -- is built on a stream of its own while this batch is in its grouping and
        scorer stages (BaseMinkowski.prepare_input), so the next call's first convolution does not wait for it; with
        backbone_ahead its backbone + heads run now as well and `after_next` (the batch after it) is the one that is prepared.
        Returns (labels int32 [N] device, PanopticResults)."""
        dev = self.device
        import time
        t0 = time.perf_counter() if self.stage_timing else 0.0
        if self.stage_timing:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        main = torch.cuda.current_stream(dev)
        ctx, self._ahead = self._ahead, None
        if ctx is not None:
            self._join(ctx)
        if ctx is not None and ctx.key == _batch_key(batch_np):
            # stage A of this batch ran ahead on its own stream: this stream takes its tensors over
            main.wait_event(ctx.done)
            for t in ctx.hold:
                t.record_stream(main)
        else:
            if ctx is not None:
                ctx.done.synchronize()  # (another batch than announced: let its launches finish before its tensors are dropped)
            ctx = self._stage_a(batch_np)
        self.model.set_input(ctx.data, dev)   # (stage B reads the batch through the model's attributes)
        feats, sem, off, emb, pred = ctx.outs
        t0 = self._tick("backbone+heads", t0)
        prefetchable = next_batch is not None and INPUT_PREFETCH
        if prefetchable and not all(torch.is_tensor(next_batch[k]) and next_batch[k].is_cuda for k in ("coords", "batch")):
            raise ValueError("next_batch must hold device tensors (the build reads them on its own stream)")
        nctx = None
        if prefetchable and self.backbone_ahead and not self.stage_timing:
            nctx = self._ahead = self._launch_ahead(next_batch, after_next, main)
            # the scorer's convolutions start when the backbone ahead is through (its record exists once its thread has enqueued it)
            scorer = getattr(self.model, "ScorerUnet", None)
            if scorer is not None and SCORER_WAITS:
                def wait_for_backbone(rec=nctx, dv=dev):
                    if rec.thread is not None:
                        rec.thread.join()
                    torch.cuda.current_stream(dv).wait_event(rec.done)
                scorer._before_first_conv = wait_for_backbone
        elif prefetchable:
            self.model.Backbone.prepare_input(Data(coords=next_batch["coords"], batch=next_batch["batch"]))
        if override is not None:
            pred, off, emb = override
        try:
            res = self.model.group_and_score(self.epoch, feats, sem, off, emb, pred, timer=self._tick if self.stage_timing else None,
                                             t0=t0)
        finally:
            scorer = getattr(self.model, "ScorerUnet", None)
            if scorer is not None:
                scorer.__dict__.pop("_before_first_conv", None)  # (no proposals / another scorer type: never consumed)
        t0 = time.perf_counter() if self.stage_timing else 0.0
        labels, counts = instance_labels_per_tile(res, self.model.input.batch, n_tiles)  # (reads the row-gather flag too)
        self._tick("nms+paint", t0)
        return labels, res, counts


def tile_batch_gpu(pos, coords, voxel, centres_xy, radius):
    """Row f1 on the device: cut a voxelised scene (pos f32 [U,3], coords i32 [U,3], both on the GPU) into vertical
    cylinders (`ops.cylinder_tiles`, CylinderSampling semantics) and collate them into one batch with the layout of
    SURVEY.md 8 a0 -- pos centred per cylinder (z kept), integer coords shifted by the rounded centre, features
    x = (x_rel, y_rel, z_rel, z), batch, origin_id -- exactly what `synthetic.tile_batch` builds with NumPy.
    Returns a dict of device tensors plus `tiles` (ops.ClusterCSR of origin ids per cylinder)."""
    dev = pos.device
    tiles = ops.cylinder_tiles(pos, centres_xy, radius)
    idx = tiles.points                                     # origin ids, tile-major, ascending inside a tile
    sizes = tiles.sizes().long()
    nt = tiles.n
    b = torch.repeat_interleave(torch.arange(nt, device=dev), sizes)
    p = pos[idx]
    # float32 means exactly like NumPy's pairwise p.mean(0) are not reproducible with atomics: accumulate in float64
    cen = torch.zeros((nt, 3), dtype=torch.float64, device=dev).index_add_(0, b, p.double())
    cen = (cen / sizes.clamp(min=1)[:, None].double()).float()
    cen[:, 2] = 0.0
    pc = p - cen[b]
    shift = torch.round(cen / voxel).to(torch.int32)
    ci = coords[idx] - shift[b]
    mean_pc = torch.zeros((nt, 3), dtype=torch.float64, device=dev).index_add_(0, b, pc.double())
    mean_pc = (mean_pc / sizes.clamp(min=1)[:, None].double()).float()
    x = torch.cat([pc - mean_pc[b], pc[:, 2:3]], 1)
    return {"pos": pc, "coords": ci, "batch": b, "x": x, "origin_id": idx, "tiles": tiles}


def grid_cylinder_centres(pos, grid_size, svd_flip="u"):
    """Centre grid of GridCylinderSampling on the device (torch_points3d/core/data_transform/transforms.py:224-243):
    PCA of the xy coordinates (mean + 2x2 covariance reduced on the GPU in float64, eigen-decomposition of the 2x2 matrix
    on the host), bounding box in the PCA frame, nodes every grid_size from min to (max + grid_size) exclusive with x
    outer / y inner, mapped back to world xy.  svd_flip picks the sign of the axes, which moves the grid when the extent
    is not a multiple of grid_size: "u" = the reference's pinned sklearn 0.24.2 (the sample with the largest |projection|
    gets a positive coordinate), "v" = sklearn >= 1.5 (largest |entry| of each axis positive).  float64 [m,2] on
    pos.device; empty cylinders are still included (see grid_cylinder_tiles)."""
    if svd_flip not in ("u", "v"):
        raise ValueError("svd_flip must be 'u' or 'v'")
    xy = pos[:, :2].double()
    n = xy.shape[0]
    if n < 2:
        raise ValueError("grid_cylinder_centres needs at least two points")
    mean = xy.mean(0)
    xc = xy - mean
    cov = (xc.T @ xc / (n - 1)).cpu().numpy()
    w, v = np.linalg.eigh(cov)                       # ascending eigenvalues, eigenvectors in columns
    comps = torch.from_numpy(np.ascontiguousarray(v[:, ::-1].T)).to(xy.device)  # rows = axes, largest variance first
    red = xc @ comps.T
    if svd_flip == "u":
        sel = red.abs().argmax(0)
        sign = torch.sign(red[sel, torch.arange(2, device=xy.device)])
    else:
        sel = comps.abs().argmax(1)
        sign = torch.sign(comps[torch.arange(2, device=xy.device), sel])
    sign = torch.where(sign == 0, torch.ones_like(sign), sign)
    comps = comps * sign[:, None]
    red = red * sign[None, :]
    lo, hi = red.min(0)[0].cpu().numpy(), red.max(0)[0].cpu().numpy()
    gx = np.arange(lo[0], hi[0] + grid_size, grid_size)
    gy = np.arange(lo[1], hi[1] + grid_size, grid_size)
    nodes = np.stack([np.repeat(gx, len(gy)), np.tile(gy, len(gx))], 1)
    return torch.from_numpy(nodes).to(xy.device) @ comps + mean


def grid_cylinder_tiles(pos, radius, grid_size=None, labels=None, svd_flip="u", nn_cell=None):
    """GridCylinderSampling (transforms.py:182-267) for a whole scene on the device: the PCA-aligned centre grid, one
    vertical cylinder of `radius` per node (CylinderSampling, inclusive radius), nodes without points dropped.
    Returns (tiles: ops.ClusterCSR with ascending origin ids per kept cylinder -- the reference lists them in KD-tree
    order --, centres float32 [k,2], centre_label: labels[nearest point in xy] per kept cylinder or None)."""
    grid_size = float(radius if grid_size is None else grid_size)
    cen = grid_cylinder_centres(pos, grid_size, svd_flip).float()
    tiles = ops.cylinder_tiles(pos, cen.contiguous(), radius)
    keep = torch.nonzero(tiles.sizes() > 0).view(-1)
    if keep.numel() < tiles.n:
        tiles = tiles.select(keep)
        cen = cen[keep]
    centre_label = None
    if labels is not None:
        idx, _ = ops.nearest(pos[:, :2].contiguous(), cen.contiguous(), cell=float(nn_cell or max(grid_size / 4, 1e-3)))
        centre_label = labels[idx]
    return tiles, cen, centre_label


def back_project(pos_full, votes, prediction_count, ins_pre, stuff_classes, max_dist=1.0, min_points=10, cell=0.25):
    """Full-resolution assignment at the end of a test area (metrics/panoptic_tracker_pointgroup_npm3d.py:555-631) on the
    device.  pos_full [N,3] is the whole cloud, votes [N,C] / prediction_count [N] / ins_pre [N] (-1 = none) hold what the
    cylinders produced.  Semantic: votes of the nearest point that has a prediction (knn_interpolate with k = 1 is a
    copy), argmax.  Instance: label of the nearest point that has an instance; -1 where the semantic prediction is a
    stuff class, where that neighbour is farther than max_dist, and for instances left with fewer than min_points
    points.  `cell` = edge of the search grid (2-4x the sub-sampling voxel).  Returns (sem int64 [N], ins int64 [N])."""
    pos_full = pos_full.float().contiguous()
    has_sem = torch.nonzero(prediction_count > 0).view(-1)
    j, _ = ops.nearest(pos_full[has_sem], pos_full, cell)
    sem = torch.argmax(votes[has_sem[j]], 1)
    ins_pre = ins_pre.long()
    has_ins = torch.nonzero(ins_pre != -1).view(-1)
    # instance labels only matter within max_dist: bound the search (same result, far fewer rings for far points)
    j, d2 = ops.nearest(pos_full[has_ins], pos_full, cell, max_dist=float(max_dist) * 1.0001)
    ins = torch.where(j >= 0, ins_pre[has_ins[j.clamp(min=0)]], torch.full_like(j, -1))
    stuff = torch.as_tensor(stuff_classes, device=sem.device).view(-1)
    ins[torch.isin(sem, stuff)] = -1
    ins[torch.sqrt(d2) > max_dist] = -1
    if ins.numel():
        lab, inv, cnt = torch.unique(ins, return_inverse=True, return_counts=True)
        small = (cnt < min_points) & (lab != -1)
        ins[small[inv]] = -1
    return sem, ins
