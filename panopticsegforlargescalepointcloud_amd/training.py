"""Data-parallel training step: one process per GPU, replicas of the model, gradients
averaged with bucketed all-reduces over RCCL (backend "nccl" on ROCm) / gloo, launched DURING backward.

The reference trains single-GPU (`train.py` -> `Trainer._train_epoch`, torch_points3d/trainer.py:150-200: set_input ->
optimize_parameters = forward + backward + optimizer.step); its DDP switch is unused by the panoptic configs.  BatchNorm
statistics stay per replica by default, as they would with DDP there; `enable_sync_bn` (PP_SYNC_BN=1) all-reduces them so that
N ranks x B/N cylinders normalise like the reference's one GPU x B cylinders (tests/test_syncbn_gpu.py).

Buckets: xGMI is point-to-point (ring all-reduce is per-link bound), so few large messages beat many small ones; the
whole model is 11.3 M fp32 parameters = 45 MB, i.e. two 32 MB buckets.  Parameters are bucketed in REVERSE registration
order (the order their gradients become ready); a post-accumulate-grad hook per parameter counts a bucket down and
launches its asynchronous all-reduce as soon as the bucket is complete and every earlier bucket has been launched, so
the collective of the decoder's gradients runs while backward is still in the encoder.  `finish()` launches what is
left (buckets holding parameters that got no gradient on this rank: ScorerEncoder / ScorerMLP are constructed but unused
with scorer_type "unet", the whole scorer before `prepare_epoch`), always in bucket order, so every rank issues the same
collectives in the same order whatever its local batch did.  A parameter that received a gradient on NO rank keeps
`.grad = None` (one small MAX all-reduce of a has-gradient mask): the optimizer skips it exactly as on one GPU -- no
weight decay, no Adam state for never-used parameters.  The same mask re-sorts the buckets after a step: parameters
that were absent everywhere move to trailing buckets and stop blocking the overlap of the others.

Collective schedule with SyncBN (`enable_sync_bn`): the BatchNorms of the backbone and the heads all-reduce their statistics from
inside forward and backward, in layer order -- the same on every rank; the proposal scorer, whose launches depend on the rank's own
proposals, is excluded (`ops.sync_bn_suspended`, per-replica statistics over the rank's proposals), and the gradient buckets are
all launched by `finish()`, after backward, so that no bucket interleaves with a BatchNorm all-reduce differently on different
ranks (tests/test_syncbn_gpu.py: two ranks, one of them without any proposal, past `prepare_epoch`)."""
import os

import torch
import torch.distributed as dist

from . import ops

BUCKET_BYTES = 32 << 20


def gradient_buckets(params, bucket_bytes=BUCKET_BYTES):
    """Deterministic partition of the parameter list into buckets of at most bucket_bytes (at least one tensor each)."""
    buckets, cur, size = [], [], 0
    for p in params:
        nbytes = p.numel() * p.element_size()
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += nbytes
    if cur:
        buckets.append(cur)
    return buckets


class GradientReducer:
    """Bucketed gradient averaging overlapped with backward.

        reducer = GradientReducer(model.parameters(), group)      # once
        loss.backward()                                          # hooks launch the buckets as they complete
        reducer.finish()                                         # launch the rest, wait, average, write back

    Every rank must construct it over the same parameter list."""

    def __init__(self, params, group=None, bucket_bytes=BUCKET_BYTES, world_size=None):
        self.group = group
        self.world = dist.get_world_size(group) if world_size is None else world_size
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes = bucket_bytes
        self._absent = frozenset()
        self._layout()
        self._handles = []
        self.launched_in_backward = 0  # buckets whose all-reduce started before finish() (overlap evidence)
        self._reset()
        if self.world > 1:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _layout(self):
        """gradients arrive in reverse registration order; parameters that received no gradient on any rank in the last
        step (known identically on every rank from the has-gradient all-reduce) go to trailing buckets so that they do not
        hold back the buckets in front of them"""
        present = [p for p in self.params[::-1] if id(p) not in self._absent]
        absent = [p for p in self.params[::-1] if id(p) in self._absent]
        self.buckets = gradient_buckets(present, self.bucket_bytes) + gradient_buckets(absent, self.bucket_bytes)
        self._bucket_of = {id(p): i for i, bucket in enumerate(self.buckets) for p in bucket}

    def _reset(self):
        self._missing = [len(b) for b in self.buckets]
        self._seen = set()
        self._next = 0
        self._pending = []

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def _on_grad(self, p):
        if id(p) in self._seen:  # a second accumulation into the same parameter (shared weights): already counted
            return
        self._seen.add(id(p))
        self._missing[self._bucket_of[id(p)]] -= 1
        self._launch_ready()

    def _launch(self, i):
        bucket = self.buckets[i]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, flat, bucket))

    def _launch_ready(self):
        # strictly in bucket order: ranks whose gradients arrive in a different order still issue identical collectives
        if ops.SYNC_BN_GROUP is not None:
            # SyncBN's backward issues blocking all-reduces of its own from inside backward; how many buckets are complete between
            # two of them differs per rank (a rank without proposals has no scorer gradients), so a bucket launched here would sit
            # at a different place of the collective sequence on different ranks -- mismatched collectives (gloo aborts, RCCL
            # hangs).  With SyncBN on every bucket waits for finish(): the sequence is [BatchNorm all-reduces in layer order]
            # [buckets in bucket order] [mask] on every rank.  (Costs the overlap: 45 MB of gradients after backward.)
            return
        while self._next < len(self.buckets) and self._missing[self._next] == 0:
            self._launch(self._next)
            self._next += 1
            self.launched_in_backward += 1

    def finish(self):
        """Launch the buckets backward did not complete, wait for all, average, write the gradients back.
        Returns the number of buckets."""
        if self.world == 1:
            self._reset()
            return 0
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        # the has-gradient mask goes out AFTER the last bucket: how many buckets backward already launched differs between
        # ranks (a rank whose batch gave no proposals completes fewer buckets in backward), so any earlier position would
        # interleave the mask with the buckets differently per rank -- mismatched collectives (gloo aborts, RCCL hangs)
        has = torch.tensor([0 if p.grad is None else 1 for p in self.params], dtype=torch.int32,
                           device=self.params[0].device if self.params else "cpu")
        has_work = dist.all_reduce(has, op=dist.ReduceOp.MAX, group=self.group, async_op=True) if self.params else None
        if has_work is not None:
            has_work.wait()
        anywhere = {id(p): bool(h) for p, h in zip(self.params, has.tolist())}
        for work, flat, bucket in self._pending:
            work.wait()
            flat.div_(self.world)
            o = 0
            for p in bucket:
                n = p.numel()
                if anywhere[id(p)]:
                    if p.grad is None:
                        p.grad = flat[o:o + n].view_as(p).clone()
                    else:
                        p.grad.copy_(flat[o:o + n].view_as(p))
                o += n
        n_buckets = len(self._pending)
        absent = frozenset(i for i, h in anywhere.items() if not h)
        if absent != self._absent:  # same decision on every rank: `anywhere` is the reduced mask
            self._absent = absent
            self._layout()
        self._reset()
        return n_buckets


def allreduce_gradients(params, world_size=None, bucket_bytes=BUCKET_BYTES, group=None):
    """Average .grad over the ranks of `group`, in place, after backward has finished (no overlap): the one-shot form of
    GradientReducer for callers that already hold the gradients.  Every rank must pass the same parameter list."""
    if world_size is None:
        world_size = dist.get_world_size(group)
    if world_size == 1:
        return 0
    reducer = GradientReducer([], group, bucket_bytes, world_size=world_size)  # no hooks
    reducer.params = [p for p in params if p.requires_grad]
    reducer._layout()
    reducer._reset()
    return reducer.finish()


def enable_sync_bn(group=True):
    """Batch statistics over the cylinders of ALL ranks (SyncBN): every training-mode BatchNorm all-reduces its per-channel
    sum, sum of squares and row count (forward) and sum(dy), sum(dy x) (backward), so 2 ranks x 2 cylinders normalise exactly
    like the reference's one GPU x 4 cylinders (conf/training/7_area1.yaml:5 batch_size 4; the reference itself is single-GPU,
    trainer.py:61-66).  group: a process group, True for the default one, None / False to go back to per-replica statistics.
    PP_SYNC_BN=1 in the environment turns it on when train_step first runs with world_size > 1."""
    ops.SYNC_BN_GROUP = group if group else None


def train_step(model, data, optimizer, epoch, device, world_size=1, group=None, reducer=None):
    """set_input -> forward -> loss -> backward (gradient all-reduces overlapped) -> optimizer step.  Returns the local
    loss.  Pass a GradientReducer built once over model.parameters() to overlap; without one the reduction runs after
    backward."""
    if not model.training:  # nn.Module.train() walks every submodule (1.5 ms of host time for this model)
        model.train()
    if world_size > 1 and ops.SYNC_BN_GROUP is None and os.environ.get("PP_SYNC_BN", "0") == "1":
        enable_sync_bn(group if group is not None else True)
    model.set_input(data, device)
    optimizer.zero_grad(set_to_none=True)
    model.forward(epoch=epoch)
    model.backward(epoch)
    if world_size > 1:
        if reducer is not None:
            reducer.finish()
        else:
            allreduce_gradients(list(model.parameters()), world_size, group=group)
    optimizer.step()
    ops.clear_groupings()  # (the cached segment groupings of this step's id tensors)
    return float(model.loss.detach())
