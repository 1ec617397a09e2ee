"""`Minkowski(architecture, input_nc, num_layers, config)` -- the backbone factory the reference models call
(torch_points3d/applications/minkowski.py:25-196), rebuilt on the MI355X ME surface.

Keeps: the compact-YAML assembly of torch_points3d/models/base_architectures/unet.py:400-474 (down_modules /
inner_modules / up_modules, per-level kwargs fetched from lists), kaiming init of MinkowskiConvolution kernels and
BN weight 1 / bias 0 (applications/minkowski.py:104-111), the forward order with the skip stack
(applications/minkowski.py:160-196) and the row-order guarantee: output row i belongs to input row i.
"""
import copy
import os

import torch
from torch import nn

from . import MinkowskiEngine as ME
from . import modules as _modules
from .config import Config, is_list, resolve_model, to_config
from .modules import MLP, Identity
from . import ops

SPECIAL_NAMES = ["radius", "max_num_neighbors", "block_names"]
EARLY_PREFETCH = os.environ.get("PP_EARLY_PREFETCH", "1") != "0"  # builder thread started inside the coordinate manager's constructor


class Data:
    """Attribute bag standing in for torch_geometric.data.Data / Batch (only what the path touches)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __getitem__(self, k):
        return getattr(self, k)

    def __contains__(self, k):
        return k in self.__dict__

    def to(self, device):
        out = Data()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device) if torch.is_tensor(v) else v)
        return out


Batch = Data


class GlobalBaseModule(nn.Module):
    """Innermost module of the (unused when scorer_type == 'unet') ScorerEncoder:
    torch_points3d/core/base_conv/message_passing.py:132-151.  MLP then per-batch-element pooling."""

    def __init__(self, nn=None, aggr="max", *args, **kwargs):
        super().__init__()
        self.nn = MLP(nn)
        self.aggr = "max" if aggr == "max" else "mean"

    def forward(self, data, **kwargs):
        x = self.nn(data.x)
        nb = int(data.batch.max().item()) + 1
        x = ops.segment_reduce(x.contiguous(), data.batch.long(), nb, self.aggr)
        return Data(x=x, batch=torch.arange(nb, device=x.device))


class _ModulesLib:
    ResNetDown = _modules.ResNetDown
    ResNetUp = _modules.ResNetUp
    ResBlock = _modules.ResBlock
    GlobalBaseModule = GlobalBaseModule


def extract_output_nc(model_config):
    if model_config.get("up_conv") is not None:
        return model_config.up_conv.up_conv_nn[-1][-1]
    if model_config.get("innermost") is not None:
        return model_config.innermost.nn[-1]
    raise ValueError("Input model_config does not match expected pattern")


class UnwrappedUnetBasedModel(nn.Module):
    def __init__(self, opt, model_type, dataset, modules_lib):
        super().__init__()
        opt = copy.deepcopy(opt)
        self.opt = opt
        if is_list(opt.down_conv) or "down_conv_nn" not in opt.down_conv:
            raise NotImplementedError
        self.down_modules = nn.ModuleList()
        self.inner_modules = nn.ModuleList()
        self.up_modules = nn.ModuleList()
        if opt.get("innermost") is not None:
            args = dict(opt.innermost)
            cls = getattr(modules_lib, args.pop("module_name"))
            self.inner_modules.append(cls(**args))
        else:
            self.inner_modules.append(Identity())
        for i in range(len(opt.down_conv.down_conv_nn)):
            args = self._fetch_arguments_from_list(opt.down_conv, i)
            cls = getattr(modules_lib, args.pop("module_name"))
            self.down_modules.append(cls(index=i, **args))
        if opt.get("up_conv") is not None:
            for i in range(len(opt.up_conv.up_conv_nn)):
                args = self._fetch_arguments_from_list(opt.up_conv, i)
                cls = getattr(modules_lib, args.pop("module_name"))
                self.up_modules.append(cls(index=i, **args))

    @staticmethod
    def _fetch_arguments_from_list(opt, index):
        """Constructor arguments of layer `index` from a down_conv / up_conv section: list-valued options are indexed
        per layer and lose a plural "s" (`down_conv_nn` style names in SPECIAL_NAMES keep theirs), scalars are shared by
        all layers (unet.py:456-474)."""
        def per_layer(key, value):
            if not (is_list(value) and len(value) > 0):
                return key, (list(value) if is_list(value) else value)
            item = value[index]
            singular = key[:-1] if key.endswith("s") and key not in SPECIAL_NAMES else key
            return singular, (list(item) if is_list(item) else item)

        return dict(per_layer(str(k), v) for k, v in opt.items())


class BaseMinkowski(UnwrappedUnetBasedModel):
    CONV_TYPE = "sparse"

    def __init__(self, model_config, model_type, dataset, modules, *args, default_output_nc=None, **kwargs):
        """Same construction order as the reference (applications/minkowski.py:81-93) because it decides the RNG stream of
        the seeded initial values: layers, then their initialisation, then -- only if the caller asks for a different
        `output_nc` -- a bias-free Linear + BN + LeakyReLU(0.2) head named `mlp` (a state_dict key)."""
        super().__init__(model_config, model_type, dataset, modules)
        self.weight_initialization()
        backbone_nc = default_output_nc or extract_output_nc(model_config)
        head_nc = kwargs.get("output_nc")
        self.mlp = None
        if head_nc is not None:
            self.mlp = MLP([backbone_nc, head_nc], activation=nn.LeakyReLU(0.2), bias=False)
        self._widths = (backbone_nc, head_nc)

    has_mlp_head = property(lambda self: self._widths[1] is not None)
    output_nc = property(lambda self: self._widths[1] if self._widths[1] is not None else self._widths[0])

    @property
    def device(self):
        return next(self.parameters()).device

    def weight_initialization(self):
        """kaiming-normal (fan_out, relu) on the kernels of MinkowskiConvolution layers -- NOT the transposed ones, which
        keep ME's default uniform init, exactly as applications/minkowski.py:104-111 tests the class -- and BN (1, 0)."""
        convs = [m for m in self.modules() if isinstance(m, ME.MinkowskiConvolution)]
        norms = [m for m in self.modules() if isinstance(m, ME.MinkowskiBatchNorm)]
        for conv in convs:
            ME.utils.kaiming_normal_(conv.kernel, mode="fan_out", nonlinearity="relu")
        for bn in norms:
            nn.init.ones_(bn.bn.weight)
            nn.init.zeros_(bn.bn.bias)

    def _plan(self):
        # (inference: the previous pass's request log lets the coordinate manager start building the coarser levels at once)
        return getattr(self, "_map_plan", None) if (ME.MAP_PREFETCH and EARLY_PREFETCH and not torch.is_grad_enabled()) else None

    def prepare_input(self, data):
        """Start building the coordinate manager of a batch this network will be given LATER (ME.PreparedCoordinates: its own
        thread and stream), e.g. the next tile batch while the current one is in its grouping / scorer stages.  The next
        forward whose `data.batch` / `data.coords` are these very tensors (unmodified) takes it over; any other input
        simply builds its own.  Inference only."""
        if torch.is_grad_enabled():
            raise RuntimeError("prepare_input is an inference-time overlap (no autograd across streams)")
        dev = self.device
        pending = self.__dict__.pop("_prepared_input", None)
        if pending is not None:
            pending[2].take()  # never taken over: let its threads finish before it is dropped
        coords = torch.cat([data.batch.unsqueeze(-1).int().to(dev), data.coords.int().to(dev)], -1)
        key = (data.batch.data_ptr(), data.coords.data_ptr(), int(data.coords.shape[0]))
        # (the source tensors are held too: while the build is pending their storage cannot be recycled for another batch of the
        # same shape, so an equal address means the same tensors)
        self._prepared_input = (key, coords, ME.PreparedCoordinates(coords, prefetch_plan=self._plan()), (data.batch, data.coords))

    def _set_input(self, data):
        dev = self.device
        prepared = None
        held = self.__dict__.pop("_prepared_input", None)
        if held is not None and held[0] == (data.batch.data_ptr(), data.coords.data_ptr(), int(data.coords.shape[0])) \
                and not torch.is_grad_enabled():
            coords, prepared = held[1], held[2]
        else:
            if held is not None:
                held[2].take()  # not this batch: let the build finish (its thread and streams) and drop it
            c4 = getattr(data, "coords4", None)  # (batch, x, y, z) rows already assembled (ops.proposals_unique)
            coords = c4 if c4 is not None else \
                torch.cat([data.batch.unsqueeze(-1).int().to(dev), data.coords.int().to(dev)], -1)
        self.input = ME.SparseTensor(features=data.x.to(dev), coordinates=coords, device=dev, prefetch_plan=self._plan(),
                                     prepared=prepared)
        self.xyz = data.pos.to(dev) if getattr(data, "pos", None) is not None else data.coords.to(dev)


class MinkowskiEncoder(BaseMinkowski):
    def forward(self, data, *args, **kwargs):
        """down modules -> (features, batch id) of the coarsest level -> innermost module (global pooling head) -> optional
        `mlp` head; returns one row per batch element (applications/minkowski.py:129-156)."""
        self._set_input(data)
        x = self.input
        for down in self.down_modules:
            x = down(x)
        pooled = Batch(x=x.F, batch=x.C[:, 0].long())
        inner = self.inner_modules[0]
        pooled = pooled if isinstance(inner, Identity) else inner(pooled)
        if self.mlp is not None:
            pooled.x = self.mlp(pooled.x)
        return pooled


class MinkowskiUnet(BaseMinkowski):
    def forward(self, data, *args, internal_order=False, **kwargs):
        """internal_order=True returns features and batch ids in the coordinate manager's row order (no un-permute
        gather) -- enough for order-independent consumers such as the per-proposal max of the scorer."""
        self._set_input(data)
        data = self.input
        cm = data.coordinate_manager
        prefetch = ME.MAP_PREFETCH and not torch.is_grad_enabled()
        plan = getattr(self, "_map_plan", None)
        if prefetch and plan is None:
            cm._log = []            # first inference pass of this model: record the level / map requests ...
        elif prefetch:
            cm.prefetch(plan)       # ... later passes replay them ahead of the convolutions on the side stream
        hook = self.__dict__.pop("_before_first_conv", None)
        if hook is not None:
            hook()                  # (scene.TileRunner: the proposal scorer's convolutions wait for the next batch's backbone here)
        try:
            stack_down = []
            for i in range(len(self.down_modules) - 1):
                data = self.down_modules[i](data)
                stack_down.append(data)
            data = self.down_modules[-1](data)
            stack_down.append(None)
            for i in range(len(self.up_modules)):
                data = self.up_modules[i](data, stack_down.pop())
        finally:
            cm.join_prefetch()
            if cm._log is not None:
                self._map_plan, cm._log = cm._log, None
        if internal_order:
            out = Data(x=data.feats, pos=None, batch=data.coordinate_manager.level(1).coords[:, 0])
        else:
            out = Data(x=data.F, pos=self.xyz, batch=data.C[:, 0])
        if self.mlp is not None:
            out.x = self.mlp(out.x)
        return out


def Minkowski(architecture=None, input_nc=None, num_layers=None, config=None, *args, **kwargs):
    """Create a sparse U-Net / encoder backbone (reference signature, applications/minkowski.py:25-54)."""
    if not architecture:
        raise ValueError()
    architecture = architecture.lower()
    if architecture not in ("unet", "encoder"):
        raise Exception("The provided argument model_architecture with value {} isn't within {}".format(
            architecture, ["unet", "encoder", "decoder"]))
    if not config:
        raise NotImplementedError("default applications/conf/sparseconv3d/*.yaml fallbacks are not shipped; pass config=")
    model_config = to_config(copy.deepcopy(config)) if not isinstance(config, Config) else copy.deepcopy(config)
    resolve_model(model_config, input_nc, kwargs)
    cls = MinkowskiUnet if architecture == "unet" else MinkowskiEncoder
    return cls(model_config, None, None, _ModulesLib, **kwargs)
