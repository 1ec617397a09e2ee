"""MI355X-native panoptic hot path (sparse-voxel U-Net + instance grouping) behind the module surface the
reference model code calls.  See DESIGN.md.  The HIP library has no CPU fallback.

Submodules load on first attribute access (`pp.MinkowskiEngine`, `pp.torch_points_kernels`, `pp.torch_scatter`, ...)
so that `import panopticsegforlargescalepointcloud_amd as pp` stays cheap and INTEGRATION.md's aliasing snippet works
as written."""
import importlib

__version__ = "0.2.0"

_SUBMODULES = ("MinkowskiEngine", "torch_points_kernels", "torch_scatter", "ops", "modules", "applications", "config",
               "panoptic", "scene", "training", "io", "synthetic", "sparseconv3d_nn", "utils", "evaluation")


def __getattr__(name):
    if name in _SUBMODULES:
        return importlib.import_module("." + name, __name__)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def __dir__():
    return sorted(list(globals()) + list(_SUBMODULES))
