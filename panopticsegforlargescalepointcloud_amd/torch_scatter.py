"""torch_scatter.scatter(src, index, dim=0, reduce=...) on the MI355X segment-reduce kernel
(call sites torch_points3d/models/panoptic/PointGroup3heads.py:419-452, core/losses/panoptic_losses.py:260,276).
Differentiable (sum / mean / max) through a small autograd Function."""
import torch

from . import ops


class _ScatterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, n_seg, reduce, check=True):
        src2 = src.reshape(src.shape[0], -1).contiguous()
        if reduce == "max" and not ctx.needs_input_grad[0]:
            out = ops.segment_reduce(src2, index, n_seg, "max", check=check)  # inference: no arg-max pass
        elif reduce == "max":
            out, arg = ops.segment_reduce(src2, index, n_seg, "max", want_arg=True)
            ctx.save_for_backward(arg)
        else:
            out = ops.segment_reduce(src2, index, n_seg, reduce, check=check)
            cnt = None
            if reduce == "mean":
                cnt = torch.zeros(n_seg, device=src.device).index_add_(0, index, torch.ones(index.numel(), device=src.device))
            ctx.save_for_backward(index, cnt)
        ctx.reduce, ctx.shape = reduce, src.shape
        return out.reshape((n_seg,) + tuple(src.shape[1:]))

    @staticmethod
    def backward(ctx, dout):
        n = ctx.shape[0]
        d2 = dout.reshape(dout.shape[0], -1)
        if ctx.reduce == "max":
            (arg,) = ctx.saved_tensors
            dsrc = torch.zeros((n, d2.shape[1]), dtype=dout.dtype, device=dout.device)
            valid = arg >= 0
            cols = torch.arange(d2.shape[1], device=dout.device).expand_as(arg)
            dsrc[arg[valid], cols[valid]] = d2[valid]
        else:
            index, cnt = ctx.saved_tensors
            dsrc = d2[index]
            if ctx.reduce == "mean":
                dsrc = dsrc / cnt[index].clamp(min=1).unsqueeze(1)
        return dsrc.reshape(ctx.shape), None, None, None, None


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum", check=True):
    """check=False (not in torch_scatter's signature): the caller built the ids itself, so the range validation -- a stream
    synchronisation and a host read -- is skipped."""
    if dim != 0 or out is not None:
        raise NotImplementedError("only scatter(src, index, dim=0) is used by the reference path")
    if reduce == "add":
        reduce = "sum"
    if reduce not in ("sum", "mean", "max"):
        raise ValueError("reduce must be sum, mean or max")
    index = index.long().reshape(-1)
    n_seg = int(dim_size) if dim_size is not None else (int(index.max().item()) + 1 if index.numel() else 0)
    squeeze = src.dim() == 1
    s = src.float().unsqueeze(1) if squeeze else src.float()
    y = _ScatterFn.apply(s, index, n_seg, reduce, bool(check))
    return y.squeeze(1) if squeeze else y


class _GatherFn(torch.autograd.Function):
    """src[index] along dim 0 (the dual of scatter-sum): the gradient is a segment sum over the rows that took the same
    source row -- torch's own backward of advanced indexing sorts the index and was the slowest kernel of the loss."""

    @staticmethod
    def forward(ctx, src, index):
        ctx.index, ctx.shape = index, src.shape
        src2 = src.reshape(src.shape[0], -1)
        return ops.gather_rows(src2.contiguous(), index).reshape((index.shape[0],) + tuple(src.shape[1:]))

    @staticmethod
    def backward(ctx, dout):
        d2 = dout.reshape(dout.shape[0], -1).contiguous().float()
        # the forward gather has already dereferenced every id, so they are in range
        dsrc = ops.segment_reduce(d2, ctx.index, ctx.shape[0], "sum", check=False)
        return dsrc.reshape(ctx.shape), None


def gather(src, index):
    """src[index] (index: int64 ids or a boolean row mask) with a segment-sum backward."""
    if index.dtype == torch.bool:
        index = torch.nonzero(index).view(-1)
    if not src.requires_grad:
        return src[index]
    return _GatherFn.apply(src, index.long())
