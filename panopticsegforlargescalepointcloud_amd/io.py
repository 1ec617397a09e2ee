"""On-disk formats either side of the hot path: point-cloud PLY files and the trainer's checkpoint.

PLY: the reference reads NPM3D scenes with fields x, y, z, scalar_class, scalar_label
(torch_points3d/datasets/segmentation/npm3d.py:76-93) and writes evaluation clouds with `write_ply`
(torch_points3d/models/panoptic/ply.py:116-315, datasets/panoptic/npm3d.py:70-85).  Same container here: one `vertex`
element with scalar properties, ascii or binary (either endianness) on read, binary_little_endian on write; no faces.

Checkpoint: `ModelCheckpoint` saves one torch file with `models = {weight_name: state_dict}` (weight names "latest",
"best_<metric>"), `optimizer = (class name, state_dict)`, `schedulers`, `stats`, `run_config`, `dataset_properties`
(torch_points3d/metrics/model_checkpoint.py:38-52).  Parameter names and shapes of this package's models equal the
reference's, so a state_dict loads key for key; MinkowskiEngine's `kernel` tensors are [K, Cin, Cout].
"""
import numpy as np
import torch

_PLY_DTYPES = {"int8": "i1", "char": "i1", "uint8": "u1", "uchar": "u1", "int16": "i2", "short": "i2", "uint16": "u2",
               "ushort": "u2", "int32": "i4", "int": "i4", "uint32": "u4", "uint": "u4", "float32": "f4", "float": "f4",
               "float64": "f8", "double": "f8"}
_PLY_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def read_ply(path):
    """-> numpy structured array with one field per vertex property."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt = f.readline().split()
        if len(fmt) < 2 or fmt[0] != b"format":
            raise ValueError("%s: missing format line" % path)
        kind = fmt[1].decode()
        if kind not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError("%s: unsupported PLY format %s" % (path, kind))
        n, props, element = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated header" % path)
            tok = line.split()
            if not tok or tok[0] == b"comment" or tok[0] == b"obj_info":
                continue
            if tok[0] == b"end_header":
                break
            if tok[0] == b"element":
                element = tok[1].decode()
                if element == "vertex":
                    n = int(tok[2])
                elif int(tok[2]) != 0:
                    raise ValueError("%s: only vertex elements are supported (found %s)" % (path, element))
            elif tok[0] == b"property" and element == "vertex":
                if tok[1] == b"list":
                    raise ValueError("%s: list properties are not supported" % path)
                props.append((tok[2].decode(), _PLY_DTYPES[tok[1].decode()]))
        if n is None:
            raise ValueError("%s: no vertex element" % path)
        if kind == "ascii":
            raw = np.loadtxt(f, dtype=np.float64, max_rows=n, ndmin=2)
            out = np.empty(n, dtype=[(name, "<" + t) for name, t in props])
            for i, (name, _) in enumerate(props):
                out[name] = raw[:, i]
            return out
        end = "<" if kind == "binary_little_endian" else ">"
        data = np.fromfile(f, dtype=[(name, end + t) for name, t in props], count=n)
        if len(data) != n:
            raise ValueError("%s: expected %d vertices, file holds %d" % (path, n, len(data)))
        return data


def write_ply(path, fields, names, text=False):
    """fields: list of arrays ([n] or [n, k], k columns get consecutive names); binary_little_endian like the reference's
    `write_ply` (models/panoptic/ply.py:116-315), or ascii (text=True) like its PlyData(..., text=True) writers."""
    cols = []
    for a in fields:
        a = np.asarray(a)
        cols += [a] if a.ndim == 1 else [a[:, j] for j in range(a.shape[1])]
    if len(cols) != len(names):
        raise ValueError("write_ply: %d columns but %d names" % (len(cols), len(names)))
    n = len(cols[0])
    if any(len(c) != n for c in cols):
        raise ValueError("write_ply: columns of different length")
    dt = []
    for name, c in zip(names, cols):
        code = c.dtype.str[1:]
        if code == "i8":
            code = "i4"          # PLY has no 64-bit integers; the reference casts labels to int32 before writing
        if code not in _PLY_NAMES:
            raise ValueError("write_ply: dtype %s of %s is not representable" % (c.dtype, name))
        dt.append((name, "<" + code))
    rec = np.empty(n, dtype=dt)
    for (name, _), c in zip(dt, cols):
        rec[name] = c
    if not path.endswith(".ply"):
        path += ".ply"
    with open(path, "wb") as f:
        head = ["ply", "format %s 1.0" % ("ascii" if text else "binary_little_endian"), "element vertex %d" % n]
        head += ["property %s %s" % (_PLY_NAMES[t[1:]], name) for name, t in dt]
        f.write(("\n".join(head) + "\nend_header\n").encode())
        if text:
            fmt = " ".join("%.9g" if t[1] == "f" else "%d" for _, t in dt)
            np.savetxt(f, np.stack([rec[name].astype(np.float64) for name, _ in dt], 1), fmt=fmt)
        else:
            rec.tofile(f)
    return path


def to_eval_ply(pos, pre_label, gt, path):
    """The evaluation cloud `final_eval` reads back (torch_points3d/datasets/panoptic/npm3d.py:70-85): x, y, z float32 and
    `preds` / `gt` int16, written as an ascii PLY like the reference's PlyData(..., text=True)."""
    pos, pre_label, gt = np.asarray(pos, np.float32), np.asarray(pre_label), np.asarray(gt)
    if pre_label.ndim != 1 or gt.ndim != 1 or len(pre_label) != len(pos) or len(gt) != len(pos):
        raise ValueError("to_eval_ply: pos [n,3], pre_label [n], gt [n] expected")
    return write_ply(path, [pos, pre_label.astype(np.int16), gt.astype(np.int16)], ["x", "y", "z", "preds", "gt"], text=True)


def read_npm3d(path, with_labels=True):
    """NPM3D scene -> (xyz float32 [n,3], semantic int64 in -1..8, instance int64 >= 0), npm3d.py:76-93."""
    d = read_ply(path)
    xyz = np.stack([d["x"], d["y"], d["z"]], 1).astype(np.float32)
    if not with_labels:
        return torch.from_numpy(xyz)
    return (torch.from_numpy(xyz), torch.from_numpy(d["scalar_class"].astype(np.int64) - 1),
            torch.from_numpy(d["scalar_label"].astype(np.int64) + 1))


def save_checkpoint(path, model, optimizer=None, weight_name="latest", stats=None, run_config=None, **extra):
    obj = {"models": {weight_name: model.state_dict()}, "stats": stats or {"train": [], "test": [], "val": []},
           "optimizer": None if optimizer is None else (optimizer.__class__.__name__, optimizer.state_dict()),
           "schedulers": {}, "run_config": run_config, "dataset_properties": {}}
    obj.update(extra)
    torch.save(obj, path)


def load_checkpoint(path, model, weight_name="latest", strict=True, optimizer=None, map_location="cpu"):
    """Loads `models[weight_name]` of a ModelCheckpoint file into `model` (falls back to "latest" like the reference,
    model_checkpoint.py:178-189); returns (missing keys, unexpected keys)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    models = ckpt.get("models")
    if not models:
        raise ValueError("%s holds no `models` entry" % path)
    if weight_name not in models:
        if "latest" not in models:
            raise KeyError("%s: weight %r not found (available: %s)" % (path, weight_name, sorted(models)))
        weight_name = "latest"
    res = model.load_state_dict(models[weight_name], strict=strict)
    if optimizer is not None and ckpt.get("optimizer"):
        name, state = ckpt["optimizer"]
        if name != optimizer.__class__.__name__:
            raise ValueError("checkpoint optimizer is %s, got %s" % (name, optimizer.__class__.__name__))
        optimizer.load_state_dict(state)
    return list(res.missing_keys), list(res.unexpected_keys)
