"""Minimal stand-in for the hydra/omegaconf pieces the hot path touches, so the reference's
conf/models/panoptic/*.yaml load UNCHANGED (pyyaml only; hydra/omegaconf are not required).

Mirrors: group composition `models=<file>` -> cfg.models.<model_name>; `${a.b.c}` interpolation against the root;
the second-stage python-expression resolver with constants FEAT / in_feat / ...
(torch_points3d/utils/model_building_utils/model_definition_resolver.py:29-58,
 torch_points3d/applications/modelfactory.py:81-99)."""
import copy
import re

import yaml


class Config(dict):
    """dict with attribute access and omegaconf-like .get()."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_config(obj):
    if isinstance(obj, dict):
        return Config({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_config(v) for v in obj]
    return obj


def is_list(v):
    return isinstance(v, (list, tuple))


_INTERP = re.compile(r"\$\{([^}]+)\}")


def _lookup(root, path):
    node = root
    for part in path.split("."):
        node = node[part] if isinstance(node, dict) else node[int(part)]
    return node


def _interpolate(node, root):
    if isinstance(node, dict):
        for k in list(node.keys()):
            node[k] = _interpolate(node[k], root)
        return node
    if isinstance(node, list):
        return [_interpolate(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node.strip())
        if m:  # whole-string interpolation keeps the type
            return _interpolate(copy.deepcopy(_lookup(root, m.group(1))), root)

        def rep(mm):
            return str(_interpolate(copy.deepcopy(_lookup(root, mm.group(1))), root))

        return _INTERP.sub(rep, node)
    return node


def resolve(obj, constants):
    """Evaluate string expressions in place (same behaviour as the reference resolver: names that are not
    constants stay strings)."""
    it = obj.keys() if isinstance(obj, dict) else range(len(obj)) if isinstance(obj, list) else None
    if it is None:
        return True
    for k in list(it):
        if resolve(obj[k], constants) and isinstance(obj[k], str):
            try:
                val = eval(obj[k], dict(constants))
                # omegaconf refuses non-primitive values (e.g. the builtin `max` for aggr: "max"): keep the string.
                # None IS assigned ("scorer_type: None" in area4_ablation_14/19.yaml must become Python None,
                # model_definition_resolver.py:44-48)
                if val is None or isinstance(val, (int, float, bool, list, tuple, str)):
                    obj[k] = val
            except NameError:
                pass
            except ValueError:
                pass
            except Exception:
                pass
    return False


def resolve_model(model_config, num_features, kwargs=None):
    constants = {"FEAT": max(num_features, 0)}
    kwargs = kwargs or {}
    if "define_constants" in model_config:
        constants.update(dict(model_config["define_constants"]))
        for key in model_config["define_constants"].keys():
            if kwargs.get(key):
                constants[key] = kwargs[key]
    resolve(model_config, constants)


def load_model_config(yaml_path, model_name, data=None, extra_root=None):
    """Compose like `models=<yaml> model_name=<name>`: returns cfg.models[model_name] with ${...} resolved
    against a root holding `models` and `data` (e.g. data={'grid_size': 0.05})."""
    with open(yaml_path) as f:
        models = yaml.safe_load(f)
    root = {"models": models, "data": dict(data or {})}
    if extra_root:
        root.update(extra_root)
    _interpolate(root["models"][model_name], root)
    cfg = to_config(root["models"][model_name])
    # top-level arithmetic such as "1.5 * 0.05" (cluster_radius_search) -- evaluated like the reference resolver does
    resolve(cfg, {})
    return cfg
