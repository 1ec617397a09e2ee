"""NumPy front-end of the CPU oracle (oracle/panoptic_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.  See the header of panoptic_oracle.c for what is pinned
against the reference and what is not ("parity unpinned" for the MinkowskiEngine / torch-points-kernels
pieces: the reference ships no vectors and those libraries are absent).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpanoptic_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "panoptic_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpanoptic_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError("oracle %s failed with status %d" % (name, rc))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


# ------------------------------------------------------------------ coordinates
def hash_first_rows(coords):
    coords = _i32(coords)
    n = coords.shape[0]
    first = np.empty(n, np.int64)
    ndup = C.c_int64(0)
    _chk(lib().ppo_hash_first_rows(_p(coords), C.c_int64(n), _p(first), C.byref(ndup)), "hash_first_rows")
    return first, ndup.value


def stride_coords(coords, ts_out):
    coords = _i32(coords)
    n = coords.shape[0]
    out = np.empty((max(n, 1), 4), np.int32)
    f2c = np.empty(max(n, 1), np.int32)
    n_out = C.c_int64(0)
    _chk(lib().ppo_stride_coords(_p(coords), C.c_int64(n), C.c_int32(ts_out), _p(out), C.byref(n_out), _p(f2c)),
         "stride_coords")
    return out[: n_out.value].copy(), f2c[:n].copy()


def kernel_map(out_coords, in_coords, ksize, step, sign):
    out_coords = _i32(out_coords)
    in_coords = _i32(in_coords)
    K = ksize ** 3
    nbr = np.empty((K, out_coords.shape[0]), np.int32)
    _chk(lib().ppo_kernel_map(_p(out_coords), C.c_int64(out_coords.shape[0]), _p(in_coords),
                              C.c_int64(in_coords.shape[0]), C.c_int32(ksize), C.c_int32(step), C.c_int32(sign),
                              _p(nbr)), "kernel_map")
    return nbr


# ------------------------------------------------------------------ convolution
def spconv_fwd(in0, weight, nbr, n_out, in1=None, scale=None, shift=None, relu=False, residual=None):
    in0 = _f32(in0)
    in1 = _f32(in1)
    weight = _f32(weight)
    if weight.ndim == 2:
        weight = weight[None]
    K, cin, cout = weight.shape
    c0 = in0.shape[1]
    c1 = 0 if in1 is None else in1.shape[1]
    assert c0 + c1 == cin
    nbr = _i32(nbr)
    out = np.empty((n_out, cout), np.float32)
    _chk(lib().ppo_spconv_fwd(_p(in0), C.c_int32(c0), _p(in1), C.c_int32(c1), _p(weight), _p(nbr), C.c_int32(K),
                              C.c_int64(n_out), C.c_int32(cout), _p(_f32(scale)), _p(_f32(shift)),
                              C.c_int32(int(relu)), _p(_f32(residual)), _p(out)), "spconv_fwd")
    return out


def spconv_bwd(inp, dout, weight, nbr, want_din=True, want_dw=True):
    inp = _f32(inp)
    dout = _f32(dout)
    weight = _f32(weight)
    if weight.ndim == 2:
        weight = weight[None]
    K, cin, cout = weight.shape
    n_in = inp.shape[0]
    n_out = dout.shape[0]
    nbr = _i32(nbr)
    din = np.empty((n_in, cin), np.float32) if want_din else None
    dw = np.empty((K, cin, cout), np.float32) if want_dw else None
    _chk(lib().ppo_spconv_bwd(_p(inp), C.c_int32(cin), C.c_int64(n_in), _p(dout), C.c_int32(cout), _p(weight),
                              _p(nbr), C.c_int32(K), C.c_int64(n_out), _p(din), _p(dw)), "spconv_bwd")
    return din, dw


def channel_stats(x):
    x = _f32(x)
    n, c = x.shape
    s = np.empty(c, np.float64)
    ss = np.empty(c, np.float64)
    _chk(lib().ppo_channel_stats(_p(x), C.c_int64(n), C.c_int32(c), _p(s), _p(ss)), "channel_stats")
    return s, ss


def affine_act(x, scale=None, shift=None, act=0, slope=0.0, residual=None):
    x = _f32(x)
    n, c = x.shape
    y = np.empty_like(x)
    _chk(lib().ppo_affine_act(_p(x), C.c_int64(n), C.c_int32(c), _p(_f32(scale)), _p(_f32(shift)), C.c_int32(act),
                              C.c_float(slope), _p(_f32(residual)), _p(y)), "affine_act")
    return y


def head_mlp(x, w1, scale, shift, w2, b2, log_softmax=False, want_argmax=False):
    x = _f32(x)
    w1 = _f32(w1)
    w2 = _f32(w2)
    n, cin = x.shape
    chid = w1.shape[0]
    cout = w2.shape[0]
    y = np.empty((n, cout), np.float32)
    am = np.empty(n, np.int64) if want_argmax else None
    _chk(lib().ppo_head_mlp(_p(x), C.c_int64(n), C.c_int32(cin), _p(w1), C.c_int32(chid), _p(_f32(scale)),
                            _p(_f32(shift)), _p(w2), _p(_f32(b2)), C.c_int32(cout), C.c_int32(int(log_softmax)),
                            _p(y), _p(am)), "head_mlp")
    return (y, am) if want_argmax else y


# ------------------------------------------------------------------ clustering
def region_grow(pos, labels, batch, ignore_labels=(), nsample=16, radius=0.02, min_cluster_size=32):
    """Returns list of int64 index arrays (ascending inside a cluster; reference order of clusters)."""
    pos = _f32(pos)
    labels = _i64(labels)
    batch = _i64(batch)
    ign = _i64(np.asarray(list(ignore_labels), dtype=np.int64))
    n = pos.shape[0]
    pc = np.empty(max(n, 1), np.int32)
    offs = np.empty(n + 2, np.int32)
    pts = np.empty(max(n, 1), np.int64)
    counts = np.zeros(2, np.int32)
    _chk(lib().ppo_region_grow(_p(pos), _p(labels), _p(batch), C.c_int64(n), _p(ign), C.c_int32(ign.shape[0]),
                               C.c_int32(nsample), C.c_float(radius), C.c_int32(min_cluster_size), _p(pc), _p(offs),
                               _p(pts), _p(counts)), "region_grow")
    nc = int(counts[0])
    return [pts[offs[i]: offs[i + 1]].copy() for i in range(nc)], pc[:n].copy()


def meanshift(x, sample_offsets, bandwidth, min_points_exclusive=3, max_iter=300):
    x = _f32(x)
    m, dim = x.shape
    so = _i64(sample_offsets)
    ns = so.shape[0] - 1
    labels = np.empty(max(m, 1), np.int32)
    ncl = np.zeros(max(ns, 1), np.int32)
    centers = np.zeros((max(m, 1), dim), np.float32)
    _chk(lib().ppo_meanshift(_p(x), C.c_int64(m), C.c_int32(dim), _p(so), C.c_int32(ns), C.c_float(bandwidth),
                             C.c_int32(min_points_exclusive), C.c_int32(max_iter), _p(labels), _p(ncl),
                             _p(centers)), "meanshift")
    return labels[:m].copy(), ncl[:ns].copy(), centers


def hdbscan(x, sample_offsets, min_cluster_size=15, min_samples=5, eps=0.006, count_self=True, min_points_exclusive=3):
    x = _f32(x)
    m, dim = x.shape
    so = _i64(sample_offsets)
    ns = so.shape[0] - 1
    labels = np.empty(max(m, 1), np.int32)
    ncl = np.zeros(max(ns, 1), np.int32)
    _chk(lib().ppo_hdbscan(_p(x), C.c_int64(m), C.c_int32(dim), _p(so), C.c_int32(ns), C.c_int32(min_points_exclusive),
                           C.c_int32(min_cluster_size), C.c_int32(min_samples), C.c_int32(1 if count_self else 0),
                           C.c_double(eps), _p(labels), _p(ncl)), "hdbscan")
    return labels[:m].copy(), ncl[:ns].copy()


def group_by_key(key, n_groups, ids=None):
    key = _i32(key)
    n = key.shape[0]
    offs = np.empty(n_groups + 1, np.int32)
    out = np.empty(max(n, 1), np.int64)
    total = C.c_int32(0)
    _chk(lib().ppo_group_by_key(_p(key), _p(_i64(ids)), C.c_int64(n), C.c_int32(n_groups), _p(offs), _p(out),
                                C.byref(total)), "group_by_key")
    return offs, out[: total.value].copy()


def segment_reduce(src, index, n_seg, reduce):
    src = _f32(src)
    index = _i64(index)
    n, c = src.shape
    code = {"sum": 0, "add": 0, "mean": 1, "max": 2}[reduce]
    out = np.empty((n_seg, c), np.float32)
    arg = np.empty((n_seg, c), np.int64)
    _chk(lib().ppo_segment_reduce(_p(src), _p(index), C.c_int64(n), C.c_int32(c), C.c_int64(n_seg), C.c_int32(code),
                                  _p(out), _p(arg)), "segment_reduce")
    return out, arg


def clusters_to_csr(clusters):
    offs = np.zeros(len(clusters) + 1, np.int32)
    for i, c in enumerate(clusters):
        offs[i + 1] = offs[i] + len(c)
    pts = np.concatenate([np.asarray(c, np.int64) for c in clusters]) if clusters else np.zeros(0, np.int64)
    return offs, _i64(pts)


def gt_layout(gt_instances, batch):
    """Per-sample GT counts / sizes in the layout torch_points_kernels.instance_iou uses."""
    gt_instances = _i64(gt_instances)
    batch = _i64(batch)
    nb = int(batch.max()) + 1 if batch.size else 0
    gt_off = np.zeros(nb + 1, np.int32)
    sizes = []
    for b in range(nb):
        g = gt_instances[batch == b]
        k = int(g.max()) if g.size else 0
        gt_off[b + 1] = gt_off[b] + k
        for i in range(1, k + 1):
            sizes.append(int((g == i).sum()))
    return gt_off, np.asarray(sizes, np.int32)


def instance_iou(clusters, gt_instances, batch):
    offs, pts = clusters_to_csr(clusters)
    gt_off, gt_sizes = gt_layout(gt_instances, batch)
    total_gt = int(gt_off[-1])
    iou = np.zeros((len(clusters), max(total_gt, 1)), np.float32)
    _chk(lib().ppo_instance_iou(_p(offs), _p(pts), C.c_int32(len(clusters)), _p(_i64(gt_instances)), _p(_i64(batch)),
                                _p(gt_off), _p(gt_sizes), C.c_int32(total_gt), _p(iou)), "instance_iou")
    return iou[:, :total_gt]


def proposal_intersections(clusters, n_points):
    offs, pts = clusters_to_csr(clusters)
    n = len(clusters)
    inter = np.zeros((n, n), np.int32)
    _chk(lib().ppo_proposal_intersections(_p(offs), _p(pts), C.c_int32(n), C.c_int64(n_points), _p(inter)),
         "proposal_intersections")
    return inter


# ------------------------------------------------------------------ f1: voxelisation / cylinder cutting (NumPy restatements)
def voxelize(pos, voxel_size, batch=None):
    """GridSampling3D(size, quantize_coords=True) -- torch_points3d/core/data_transform/grid_transform.py:181-198 with
    torch_geometric's voxel_grid / consecutive_cluster: coords = round-half-even(pos / size) on the float32 quotient,
    voxels ordered by (batch, z, y, x), representative = last point of the voxel in input order.
    Returns (coords int32 [V,4] (b,x,y,z), rep_index int64 [V], inverse int64 [n])."""
    pos = np.asarray(pos, np.float32)
    n = len(pos)
    b = np.zeros(n, np.int64) if batch is None else np.asarray(batch, np.int64)
    q = np.rint(pos / np.float32(voxel_size)).astype(np.int64)
    key = (b << 48) | ((q[:, 2] + 32768) << 32) | ((q[:, 1] + 32768) << 16) | (q[:, 0] + 32768)
    uniq, inverse = np.unique(key, return_inverse=True)
    rep = np.zeros(len(uniq), np.int64)
    rep[inverse] = np.arange(n)  # later points overwrite earlier ones: the last point of each voxel wins
    coords = np.stack([b[rep], q[rep, 0], q[rep, 1], q[rep, 2]], 1).astype(np.int32)
    return coords, rep, inverse.astype(np.int64)


def cylinder_tiles(pos, centres_xy, radius):
    """CylinderSampling (KDTree.query_radius: inclusive) for every centre -- transforms.py:388-441; ascending indices."""
    pos = np.asarray(pos, np.float32)
    r2 = np.float32(radius) * np.float32(radius)
    out = []
    for c in np.asarray(centres_xy, np.float32):
        dx, dy = pos[:, 0] - c[0], pos[:, 1] - c[1]
        d2 = (dy.astype(np.float64) ** 2 + (dx * dx).astype(np.float64)).astype(np.float32)  # fmaf(dy, dy, dx * dx)
        out.append(np.nonzero(d2 <= r2)[0])
    return out


def grid_cylinder_centres(pos, grid_size, u_based=False):
    """Centre grid of GridCylinderSampling -- torch_points3d/core/data_transform/transforms.py:224-243: PCA(n_components=2)
    of the xy coordinates, bounding box in the PCA frame, centres every grid_size from (min) to (max + grid_size)
    exclusive (np.arange), x outer / y inner, mapped back with inverse_transform.  sklearn's PCA is the reference's own
    dependency; u_based=True re-applies the sign rule of the pinned sklearn 0.24.2 (svd_flip on U: the sample with the
    largest |projection| gets a positive coordinate), False keeps the installed sklearn's (>= 1.5: largest |entry| of each
    component positive).  Returns float64 [m,2] (all grid nodes, empty cylinders included)."""
    from sklearn.decomposition import PCA
    xy = np.asarray(pos, np.float32)[:, :2]
    pca = PCA(n_components=2)
    pca.fit(xy)
    comps = pca.components_.copy()
    if u_based:
        proj = np.dot(xy - pca.mean_, comps.T)
        sel = np.argmax(np.abs(proj), axis=0)
        comps = comps * np.sign(proj[sel, np.arange(2)])[:, None]
    red = np.dot(xy - pca.mean_, comps.T)
    minx, miny = np.min(red[:, 0]), np.min(red[:, 1])
    maxx, maxy = np.max(red[:, 0]), np.max(red[:, 1])
    out = []
    for c_x in np.arange(minx, maxx + grid_size, grid_size):
        for c_y in np.arange(miny, maxy + grid_size, grid_size):
            out.append(np.dot(np.vstack((c_x, c_y)).T, comps) + pca.mean_)  # = pca.inverse_transform
    return np.stack(out).reshape(-1, 2).astype(np.float64)


def nearest(ref, query, max_dist=0.0):
    """torch_geometric knn(x=ref, y=query, k=1) by brute force (metrics/panoptic_tracker_pointgroup_npm3d.py:593):
    float32 squared distance ((dx*dx + dy*dy) + dz*dz), first minimum = smallest reference index.
    Returns (idx int64 [-1 where nothing within max_dist > 0], dist2 float32)."""
    ref = np.asarray(ref, np.float32)
    query = np.asarray(query, np.float32)
    nq, dim = query.shape
    idx = np.full(nq, -1, np.int64)
    d2 = np.full(nq, np.inf, np.float32)
    if len(ref) == 0:
        return idx, d2
    for s in range(0, nq, 1024):
        q = query[s:s + 1024]
        d = np.zeros((len(q), len(ref)), np.float32)
        for a in range(dim):
            t = q[:, a:a + 1] - ref[None, :, a]
            d = d + t * t if a else t * t
        j = np.argmin(d, axis=1)
        idx[s:s + 1024] = j
        d2[s:s + 1024] = d[np.arange(len(q)), j]
    if max_dist > 0:
        far = d2 > np.float32(max_dist) * np.float32(max_dist)
        idx[far] = -1
        d2[far] = np.inf
    return idx, d2


def back_project(pos_full, votes, prediction_count, ins_pre, stuff_classes, max_dist=1.0, min_points=10):
    """Full-resolution assignment at the end of a test area -- metrics/panoptic_tracker_pointgroup_npm3d.py:555-631.
    pos_full [N,3] is the whole cloud; votes [N,C] / prediction_count [N] / ins_pre [N] (-1 = none) hold what the cylinders
    produced (rows without a prediction are zero / -1).  Semantic: class votes of the nearest point that has a
    prediction (knn_interpolate, k=1), argmax.  Instance: label of the nearest point that has an instance (knn, k=1);
    -1 where the semantic prediction is a stuff class, where that neighbour is farther than max_dist, and for
    instances left with fewer than min_points points.  Returns (sem int64 [N], ins int64 [N])."""
    pos_full = np.asarray(pos_full, np.float32)
    has_sem = np.asarray(prediction_count) > 0
    j, _ = nearest(pos_full[has_sem], pos_full)
    full_pred = np.asarray(votes, np.float32)[has_sem][j]
    sem = np.argmax(full_pred, 1).astype(np.int64)
    ins_pre = np.asarray(ins_pre, np.int64)
    has_ins = ins_pre != -1
    j, d2 = nearest(pos_full[has_ins], pos_full)
    ins = ins_pre[has_ins][j].copy()
    for l in np.asarray(stuff_classes).reshape(-1):
        ins[sem == l] = -1
    ins[np.sqrt(d2) > max_dist] = -1
    for l in np.unique(ins):
        if l == -1:
            continue
        m = ins == l
        if m.sum() < min_points:
            ins[m] = -1
    return sem, ins


def round_bf16(x):
    """float32 -> nearest bfloat16 (ties to even) -> float32: the operand rounding of the bf16 convolution entries
    (pp_spconv_fwd_bf16 / pp_spconv_bwd_weight_bf16).  The oracle of those entries is the fp32 restatement applied to
    inputs rounded this way -- products of two bfloat16 numbers are exact in fp32, so only the summation order differs."""
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = r.astype(np.uint32).view(np.float32).copy()
    out[~np.isfinite(x)] = x[~np.isfinite(x)]
    return out.reshape(x.shape)
