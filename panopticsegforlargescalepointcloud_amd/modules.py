"""Build-owned mirror of the reference's sparse U-Net building blocks, on the MI355X ME surface.

Same class names, constructor arguments, sub-module names (=> identical state_dict keys and shapes, SURVEY.md
App. A) and forward semantics as
  torch_points3d/modules/MinkowskiEngine/api_modules.py:9-82 (ResBlock), :85-232 (BottleneckBlock, SELayer, SEBlock,
  SEBottleneckBlock), :235-285 (ResNetDown), :288-311 (ResNetUp)
  torch_points3d/core/common_modules/base_modules.py:35-45 (MLP), :128-153 (FastBatchNorm1d), :156-164 (Seq).
In eval mode every conv -> BN -> ReLU (+ residual, + skip concat) chain is ONE fused kernel launch
(ME.conv_bn_act); in training mode conv + BN (+ ReLU) run as one autograd node per pair (ME.conv_bn_act_train; same launches
and results as module by module, which remains the fallback).
"""
import os
import sys

import torch
from torch import nn

from . import MinkowskiEngine as ME
from . import ops


class Seq(nn.Sequential):
    """nn.Sequential with a chaining `append`: children are named "0", "1", ... in insertion order, which fixes the
    state_dict keys of SURVEY.md App. A (base_modules.py:156-164)."""

    def append(self, module):
        self.add_module(str(len(self)), module)
        return self


class Identity(nn.Module):
    def forward(self, data):
        return data


class FastBatchNorm1d(nn.Module):
    def __init__(self, num_features, momentum=0.1, **kwargs):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(num_features, momentum=momentum, **kwargs)

    def forward(self, x):
        if x.dim() == 2:
            bn = self.batch_norm
            if self.training and x.is_cuda and bn.momentum is not None and ops._sync_bn_group() is not None:
                # SyncBN (training.enable_sync_bn): the heads' BatchNorm joins the all-reduced statistics of the sparse layers
                running = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if bn.track_running_stats else None
                return ME._BatchNormTrainFn.apply(x, bn.weight, bn.bias, bn.eps, False, bn.momentum, running)
            return self.batch_norm(x)
        if x.dim() == 3:
            return self.batch_norm(x.permute(0, 2, 1)).permute(0, 2, 1)
        raise ValueError("Non supported number of dimensions {}".format(x.dim()))


# forward / input gradient of the skinny Linear layers from the library (PP_LINEAR_ROWS=0: torch GEMMs, A/B runs)
LINEAR_ROWS = os.environ.get("PP_LINEAR_ROWS", "1") != "0"


class _SkinnyLinearFn(torch.autograd.Function):
    """y = x W^T + b with the weight gradient as ONE streaming reduction (ops.linear_wgrad): for [millions, <= 32] inputs the
    rocBLAS split-K GEMM torch.nn.Linear's backward dispatches takes 340 - 720 us per layer (2.8 ms of a 43 ms training
    step for the six layers of the three heads).  Round 4: forward and input gradient are the library's too (ops.linear_rows:
    a thread per row, fixed summation order) -- the twelve hipBLASLt launches per step the heads still made are gone."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.linear_rows(x.contiguous(), weight, bias) if LINEAR_ROWS else torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_rows(dy, weight, None, transposed=True) if LINEAR_ROWS else dy @ weight
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.linear_wgrad(x.contiguous(), dy, want_bias=ctx.has_bias)
        return dx, dw, db


class Linear(nn.Linear):
    """torch.nn.Linear (same parameters, same state_dict keys) whose backward takes the streaming weight gradient when the
    layer is skinny and the input is a large device matrix; everything else is the parent's."""
    SKINNY_MIN_ROWS = 4096

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and self.in_features <= 32 and self.out_features <= 32
                and x.shape[0] >= self.SKINNY_MIN_ROWS and x.dtype == torch.float32):
            return _SkinnyLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


def MLP(channels, activation=None, bn_momentum=0.1, bias=True):
    activation = activation if activation is not None else nn.LeakyReLU(0.2)
    return nn.Sequential(*[
        nn.Sequential(Linear(channels[i - 1], channels[i], bias=bias),
                      FastBatchNorm1d(channels[i], momentum=bn_momentum), activation)
        for i in range(1, len(channels))
    ])


def head_spec(head, log_softmax=False, want_argmax=False):
    """tensors of a head  Seq[ MLP([c, c], bias=False), Linear(c, k) (, LogSoftmax) ]  for ops.heads (eval mode)"""
    lin1, fbn = head[0][0][0], head[0][0][1]
    lin2 = head[1]
    bn = fbn.batch_norm
    with torch.no_grad():
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        if lin1.bias is not None:
            shift = shift + lin1.bias * scale
    return (lin1.weight, scale.contiguous(), shift.contiguous(), lin2.weight, lin2.bias, log_softmax, want_argmax)


def fused_head(head, x, log_softmax=False, want_argmax=False):
    """Eval-mode fused launch of a head  Seq[ MLP([c, c], bias=False), Linear(c, k) (, LogSoftmax) ]
    (PointGroup3heads.py:69-81).  Falls back to the torch modules in training mode (autograd)."""
    mlp = head[0]
    lin1, fbn = mlp[0][0], mlp[0][1]
    lin2 = head[1]
    bn = fbn.batch_norm
    with torch.no_grad():
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
        if lin1.bias is not None:
            shift = shift + lin1.bias * scale
    return ops.head_mlp(x.contiguous(), lin1.weight, scale.contiguous(), shift.contiguous(), lin2.weight, lin2.bias,
                        log_softmax=log_softmax, want_argmax=want_argmax)


FUSE_SHORTCUT = os.environ.get("PP_FUSE_SHORTCUT", "1") != "0"


class ResBlock(ME.MinkowskiNetwork):
    """conv3-BN-ReLU-conv3-BN-ReLU, plus (1x1 conv-BN of the input | the input); ReLU BEFORE the add, none after."""

    def __init__(self, input_nc, output_nc, convolution, dimension=3):
        ME.MinkowskiNetwork.__init__(self, dimension)
        self.block = (
            Seq()
            .append(convolution(in_channels=input_nc, out_channels=output_nc, kernel_size=3, stride=1, dilation=1,
                                bias=False, dimension=dimension))
            .append(ME.MinkowskiBatchNorm(output_nc))
            .append(ME.MinkowskiReLU())
            .append(convolution(in_channels=output_nc, out_channels=output_nc, kernel_size=3, stride=1, dilation=1,
                                bias=False, dimension=dimension))
            .append(ME.MinkowskiBatchNorm(output_nc))
            .append(ME.MinkowskiReLU())
        )
        if input_nc != output_nc:
            self.downsample = (
                Seq()
                .append(convolution(in_channels=input_nc, out_channels=output_nc, kernel_size=1, stride=1, dilation=1,
                                    bias=False, dimension=dimension))
                .append(ME.MinkowskiBatchNorm(output_nc))
            )
        else:
            self.downsample = None

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled():
            b = self.block
            h = ME.conv_bn_act(x, b[0], b[1], relu=True)
            if self.downsample and FUSE_SHORTCUT:
                # the 1x1 shortcut rides on the block's last convolution (one launch, no intermediate tensor) ...
                out = ME.conv_bn_act(h, b[3], b[4], relu=True, shortcut=(x, self.downsample[0], self.downsample[1]))
                if out is not None:
                    return out
            # ... unless the library does not serve the shape that way (small split-K launches, 4-GiB inputs)
            res = ME.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False) if self.downsample else x
            return ME.conv_bn_act(h, b[3], b[4], relu=True, residual=res)
        b = self.block
        if self.training:
            # training: one autograd node per conv + BN (+ ReLU) instead of one per module (same launches, same results)
            h = ME.conv_bn_act_train(x, b[0], b[1], relu=True)
            if h is not None:
                # the first pair has run (and updated its running statistics): never run it again.  A second pair that is
                # not the plain case (bias, BatchNorm in eval mode in a partially frozen block) goes module by module
                # FROM h -- falling back to self.block(x) would count the first BatchNorm's batch twice
                out = ME.conv_bn_act_train(h, b[3], b[4], relu=True)
                if out is None:
                    out = h
                    for m in list(b)[3:]:
                        out = m(out)
                res = x
                if self.downsample:
                    res = ME.conv_bn_act_train(x, self.downsample[0], self.downsample[1], relu=False)
                    if res is None:
                        res = self.downsample(x)
                return out + res
        out = self.block(x)
        if self.downsample:
            out = out + self.downsample(x)
        else:
            out = out + x
        return out


class BottleneckBlock(ME.MinkowskiNetwork):
    """1x1 (C/reduction) - 3x3 - 1x1 (C) convolutions, each followed by BN + ReLU, plus (1x1 conv-BN of the input | the
    input) (api_modules.py:85-159; selected by `block: BottleneckBlock` in a backbone YAML).  Same sub-module names as the
    reference's class.  (The reference's constructor assigns `self.block` without calling the Module constructor first and
    raises AttributeError; pinned in tests/test_reference_binding.py.)"""

    def __init__(self, input_nc, output_nc, convolution, dimension=3, reduction=4):
        ME.MinkowskiNetwork.__init__(self, dimension)
        mid = output_nc // reduction
        self.block = Seq()
        for cin, cout, ks in ((input_nc, mid, 1), (mid, mid, 3), (mid, output_nc, 1)):
            self.block.append(convolution(in_channels=cin, out_channels=cout, kernel_size=ks, stride=1, dilation=1, bias=False,
                                          dimension=dimension))
            self.block.append(ME.MinkowskiBatchNorm(cout))
            self.block.append(ME.MinkowskiReLU())
        if input_nc != output_nc:
            self.downsample = (
                Seq()
                .append(convolution(in_channels=input_nc, out_channels=output_nc, kernel_size=1, stride=1, dilation=1,
                                    bias=False, dimension=dimension))
                .append(ME.MinkowskiBatchNorm(output_nc))
            )
        else:
            self.downsample = None

    def _body(self, x):
        """the conv-BN-ReLU chain: one fused launch per triple in inference, one autograd node per triple in training"""
        b = self.block
        out = x
        for i in range(0, len(b), 3):
            if not self.training and not torch.is_grad_enabled():
                out = ME.conv_bn_act(out, b[i], b[i + 1], relu=True)
            else:
                nxt = ME.conv_bn_act_train(out, b[i], b[i + 1], relu=True) if self.training else None
                out = nxt if nxt is not None else b[i + 2](b[i + 1](b[i](out)))
        return out

    def _shortcut(self, x):
        if not self.downsample:
            return x
        if not self.training and not torch.is_grad_enabled():
            return ME.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        return self.downsample(x)

    def forward(self, x):
        return self._body(x) + self._shortcut(x)


class SELayer(nn.Module):
    """squeeze and excite (api_modules.py:162-191): per batch element the mean feature vector -> Linear(C, C/reduction) ->
    ReLU -> Linear(C/reduction, C) -> Sigmoid, multiplied back onto every row of the element"""

    def __init__(self, channel, reduction=16, dimension=3):
        super().__init__()
        self.fc = nn.Sequential(ME.MinkowskiLinear(channel, channel // reduction), ME.MinkowskiReLU(),
                                ME.MinkowskiLinear(channel // reduction, channel), ME.MinkowskiSigmoid())
        self.pooling = ME.MinkowskiGlobalPooling()
        self.broadcast_mul = ME.MinkowskiBroadcastMultiplication()

    def forward(self, x):
        return self.broadcast_mul(x, self.fc(self.pooling(x)))


class SEBlock(ResBlock):
    """ResBlock with the SE layer between the block and the residual add (api_modules.py:194-210)"""

    def __init__(self, input_nc, output_nc, convolution, dimension=3, reduction=16):
        super().__init__(input_nc, output_nc, convolution, dimension=3)
        self.SE = SELayer(output_nc, reduction=reduction, dimension=dimension)

    def forward(self, x):
        b = self.block
        if not self.training and not torch.is_grad_enabled():
            out = ME.conv_bn_act(ME.conv_bn_act(x, b[0], b[1], relu=True), b[3], b[4], relu=True)
            res = ME.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False) if self.downsample else x
        else:
            out = self.block(x)
            res = self.downsample(x) if self.downsample else x
        return self.SE(out) + res


class SEBottleneckBlock(BottleneckBlock):
    """BottleneckBlock with the SE layer (api_modules.py:213-232)"""

    def __init__(self, input_nc, output_nc, convolution, dimension=3, reduction=16):
        super().__init__(input_nc, output_nc, convolution, dimension=3, reduction=4)
        self.SE = SELayer(output_nc, reduction=reduction, dimension=dimension)

    def forward(self, x):
        return self.SE(self._body(x)) + self._shortcut(x)


_res_blocks = sys.modules[__name__]


class ResNetDown(ME.MinkowskiNetwork):
    CONVOLUTION = ME.MinkowskiConvolution

    def __init__(self, down_conv_nn=[], kernel_size=2, dilation=1, dimension=3, stride=2, N=1, block="ResBlock", **kwargs):
        block = getattr(_res_blocks, block)
        ME.MinkowskiNetwork.__init__(self, dimension)
        conv1_output = down_conv_nn[0] if stride > 1 else down_conv_nn[1]
        self.conv_in = (
            Seq()
            .append(self.CONVOLUTION(in_channels=down_conv_nn[0], out_channels=conv1_output, kernel_size=kernel_size,
                                     stride=stride, dilation=dilation, bias=False, dimension=dimension))
            .append(ME.MinkowskiBatchNorm(conv1_output))
            .append(ME.MinkowskiReLU())
        )
        if N > 0:
            self.blocks = Seq()
            for _ in range(N):
                self.blocks.append(block(conv1_output, down_conv_nn[1], self.CONVOLUTION, dimension=dimension))
                conv1_output = down_conv_nn[1]
        else:
            self.blocks = None

    def _conv_in(self, x, skip=None):
        if not self.training and not torch.is_grad_enabled():
            return ME.conv_bn_act(x, self.conv_in[0], self.conv_in[1], relu=True, skip=skip)
        if skip is not None:
            x = ME.cat(x, skip)
        if self.training:
            out = ME.conv_bn_act_train(x, self.conv_in[0], self.conv_in[1], relu=True)
            if out is not None:
                return out
        return self.conv_in(x)

    def forward(self, x):
        out = self._conv_in(x)
        if self.blocks:
            out = self.blocks(out)
        return out


class ResNetUp(ResNetDown):
    CONVOLUTION = ME.MinkowskiConvolutionTranspose

    def __init__(self, up_conv_nn=[], kernel_size=2, dilation=1, dimension=3, stride=2, N=1, **kwargs):
        super().__init__(down_conv_nn=up_conv_nn, kernel_size=kernel_size, dilation=dilation, dimension=dimension,
                         stride=stride, N=N, **kwargs)

    def forward(self, x, skip):
        out = self._conv_in(x, skip)  # ME.cat(x, skip): upsampled channels first, then the skip's
        if self.blocks:
            out = self.blocks(out)
        return out
