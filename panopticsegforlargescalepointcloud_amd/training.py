"""Data-parallel training step: one process per GPU, replicas of the model, gradients
averaged with bucketed all-reduces over RCCL (backend "nccl" on ROCm) / gloo.

The reference trains single-GPU (`train.py` -> `Trainer._train_epoch`, torch_points3d/trainer.py:150-200: set_input ->
optimize_parameters = forward + backward + optimizer.step); its DDP switch is unused by the panoptic configs.  BatchNorm
statistics stay per replica, as they would with DDP there (no SyncBN in the reference).

Buckets: xGMI is point-to-point (ring all-reduce is per-link bound), so few large messages beat many small ones; the
whole model is 11.3 M fp32 parameters = 45 MB, i.e. two 32 MB buckets.  All buckets are launched asynchronously right
after backward and waited for together.  Parameters that received no gradient on a rank (ScorerEncoder / ScorerMLP are
constructed but unused with scorer_type "unet"; the scorer itself before `prepare_epoch`) contribute zeros, so every
rank issues the same collectives in the same order."""
import torch
import torch.distributed as dist

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


def allreduce_gradients(params, world_size=None, bucket_bytes=BUCKET_BYTES, group=None):
    """Average .grad over the ranks of `group`, in place.  Every rank must pass the same parameter list."""
    if world_size is None:
        world_size = dist.get_world_size(group)
    if world_size == 1:
        return 0
    params = [p for p in params if p.requires_grad]
    pending = []
    for bucket in gradient_buckets(params, bucket_bytes):
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, bucket))
    for work, flat, bucket in pending:
        work.wait()
        flat.div_(world_size)
        o = 0
        for p in bucket:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[o:o + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[o:o + n].view_as(p))
            o += n
    return len(pending)


def train_step(model, data, optimizer, epoch, device, world_size=1, group=None):
    """set_input -> forward -> loss -> backward -> gradient all-reduce -> optimizer step.  Returns the local loss."""
    model.train()
    model.set_input(data, device)
    optimizer.zero_grad(set_to_none=True)
    model.forward(epoch=epoch)
    model.backward(epoch)
    if world_size > 1:
        allreduce_gradients(list(model.parameters()), world_size, group=group)
    optimizer.step()
    return float(model.loss.detach())
