"""Generates the committed golden fixtures in tests/golden/ by running the REFERENCE's own Python
(importable pieces only) in the build container.  Run:  python tests/golden/make_golden.py

Not needed at test time -- /root/reference does not exist on the GPU box; only the .npz files travel.
What it pins:
  meanshift_cases.npz  <- torch_points3d/utils/meanshift_cluster.py:9-18  (sklearn MeanShift, bin_seeding)
  loss_cases.npz       <- torch_points3d/core/losses/panoptic_losses.py (offset_loss, discriminative_loss,
                          instance_iou_loss) imported by file path with stub third-party modules
  nms_cases.npz        <- torch_points3d/models/panoptic/structure_3heads.py get_instances / NMS
                          (Tensor.cuda patched to identity)
  block_merging_cases.npz <- metrics/panoptic_tracker_pointgroup_npm3d.py:339-452 block_merging, block after block
Fixtures hold inputs and expected outputs only (no reference source text).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("PP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def canon(labels):
    """relabel by first appearance so partitions compare bit-exactly"""
    labels = np.asarray(labels)
    out = np.full(labels.shape, -1, np.int64)
    seen = {}
    for i, l in enumerate(labels.tolist()):
        if l < 0:
            continue
        if l not in seen:
            seen[l] = len(seen)
        out[i] = seen[l]
    return out


def blobs(rng, n, dim, n_inst, spread=3.0, sigma=0.15):
    cen = rng.normal(0, spread, size=(n_inst, dim))
    ids = rng.integers(0, n_inst, size=n)
    x = cen[ids] + rng.normal(0, sigma, size=(n, dim))
    return x.astype(np.float32), ids


def make_meanshift():
    ms = _load(os.path.join(REF, "torch_points3d/utils/meanshift_cluster.py"), "ref_meanshift_cluster")
    rng = np.random.default_rng(2022)
    cases = {}
    specs = [("a", 2000, 5, 12, 0.6), ("b", 2000, 5, 30, 0.6), ("c", 1500, 5, 6, 0.6), ("d3", 1200, 3, 8, 0.6),
             ("wide", 1500, 5, 10, 1.0)]
    for name, n, dim, k, bw in specs:
        x, _ = blobs(rng, n, dim, k)
        lab = ms.meanshift_cluster(x, bw).numpy()
        cases["x_" + name] = x
        cases["bw_" + name] = np.float32(bw)
        cases["labels_" + name] = lab.astype(np.int64)
    # degenerate: 4 points (the smallest sample the wrapper clusters: > 3 points)
    x = np.array([[0, 0, 0, 0, 0], [0.1, 0, 0, 0, 0], [5, 5, 5, 5, 5], [5.1, 5, 5, 5, 5]], np.float32)
    cases["x_tiny"] = x
    cases["bw_tiny"] = np.float32(0.6)
    cases["labels_tiny"] = ms.meanshift_cluster(x, 0.6).numpy().astype(np.int64)
    # every point its own bin -> "using data points as seeds" branch
    x = (rng.normal(0, 20, size=(40, 5))).astype(np.float32)
    cases["x_sparse"] = x
    cases["bw_sparse"] = np.float32(0.6)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cases["labels_sparse"] = ms.meanshift_cluster(x, 0.6).numpy().astype(np.int64)
    cases["names"] = np.array([s[0] for s in specs] + ["tiny", "sparse"])
    np.savez_compressed(os.path.join(OUT, "meanshift_cases.npz"), **cases)
    print("meanshift:", {k: int(cases["labels_" + k].max()) + 1 for k in cases["names"]})


def _stub_modules():
    tpk = types.ModuleType("torch_points_kernels")

    def instance_iou(clusters, gt, batch):
        # brute-force definition -- only used to produce golden values
        nb = int(batch.max()) + 1
        ks = [int(gt[batch == b].max()) for b in range(nb)]
        off = np.concatenate([[0], np.cumsum(ks)])
        out = torch.zeros(len(clusters), int(off[-1]))
        for p, c in enumerate(clusters):
            b = int(batch[c[0]])
            for g in range(1, ks[b] + 1):
                gm = (gt == g) & (batch == b)
                inter = int((gt[c] == g).sum())
                out[p, off[b] + g - 1] = inter / float(len(c) + int(gm.sum()) - inter)
        return out

    tpk.instance_iou = instance_iou
    sys.modules["torch_points_kernels"] = tpk
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, reduce="sum"):
        n = int(index.max()) + 1
        shape = (n,) + tuple(src.shape[1:])
        idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
        red = {"sum": "sum", "add": "sum", "mean": "mean", "max": "amax"}[reduce]
        return torch.zeros(shape, dtype=src.dtype).scatter_reduce(0, idx, src, red, include_self=False)

    ts.scatter = scatter
    sys.modules["torch_scatter"] = ts
    nb = types.ModuleType("numba")
    nb.prange = range
    sys.modules["numba"] = nb
    torch.Tensor.cuda = lambda self, *a, **k: self


def make_losses():
    _stub_modules()
    L = _load(os.path.join(REF, "torch_points3d/core/losses/panoptic_losses.py"), "ref_panoptic_losses")
    g = torch.Generator().manual_seed(0)
    n = 1000
    batch = torch.cat([torch.zeros(n // 2), torch.ones(n - n // 2)]).long()
    inst = torch.randint(0, 6, (n,), generator=g)  # 0 = no instance, 1..5
    pred_off = torch.randn(n, 3, generator=g)
    gt_off = torch.randn(n, 3, generator=g)
    embed = torch.randn(n, 5, generator=g)
    mask = inst > 0
    ol = L.offset_loss(pred_off[mask], gt_off[mask], int(mask.sum()))
    dl = L.discriminative_loss(embed[mask], inst[mask], batch[mask], 5)
    # proposals: 12 random subsets inside one batch element each
    clusters = []
    for p in range(12):
        b = p % 2
        pool = torch.nonzero(batch == b).view(-1)
        sel = pool[torch.randperm(pool.numel(), generator=g)[: 30 + 5 * p]]
        clusters.append(torch.sort(sel)[0])
    scores = torch.rand(12, generator=g)
    ious = L.instance_ious(clusters, scores, inst, batch, None, False)
    sl = L.instance_iou_loss(ious, clusters, scores, inst, batch, 0.25, 0.75)
    np.savez_compressed(
        os.path.join(OUT, "loss_cases.npz"), batch=batch.numpy(), inst=inst.numpy(), pred_off=pred_off.numpy(),
        gt_off=gt_off.numpy(), embed=embed.numpy(),
        offset_norm_loss=ol["offset_norm_loss"].numpy(), offset_dir_loss=ol["offset_dir_loss"].numpy(),
        ins_loss=dl["ins_loss"].numpy(), ins_var_loss=dl["ins_var_loss"].numpy(),
        ins_dist_loss=dl["ins_dist_loss"].numpy(), ins_reg_loss=dl["ins_reg_loss"].numpy(),
        cluster_offsets=np.cumsum([0] + [len(c) for c in clusters]), cluster_points=torch.cat(clusters).numpy(),
        scores=scores.numpy(), ious=ious.numpy(), score_loss=sl.numpy())
    print("losses:", float(ol["offset_norm_loss"]), float(ol["offset_dir_loss"]), float(dl["ins_loss"]), float(sl))


def make_mask_losses():
    """The mask-supervised branch of the reference's losses (core/losses/panoptic_losses.py:25-90 with
    cal_iou_based_on_mask=True, mask_loss :156-201), executed here on seeded inputs -> mask_loss_cases.npz."""
    _stub_modules()
    L = _load(os.path.join(REF, "torch_points3d/core/losses/panoptic_losses.py"), "ref_panoptic_losses")
    g = torch.Generator().manual_seed(4)
    n = 1200
    batch = torch.cat([torch.zeros(n // 3), torch.ones(n // 3), 2 * torch.ones(n - 2 * (n // 3))]).long()
    inst = torch.randint(0, 5, (n,), generator=g)  # 0 = no instance, 1..4 in every batch element
    clusters = []
    for p in range(15):
        b = p % 3
        pool = torch.nonzero(batch == b).view(-1)
        if p < 9:   # mostly one ground-truth instance (IoU > 0.5 -> supervised mask), plus a few strangers
            own = pool[inst[pool] == 1 + p % 4]
            other = pool[inst[pool] != 1 + p % 4]
            sel = torch.cat([own[torch.randperm(own.numel(), generator=g)[: int(0.9 * own.numel())]],
                             other[torch.randperm(other.numel(), generator=g)[: 5 + p]]])
        else:       # random subsets (IoU < 0.5 -> weight 0)
            sel = pool[torch.randperm(pool.numel(), generator=g)[: 40 + 3 * p]]
        clusters.append(torch.sort(sel)[0])
    n_rows = sum(len(c) for c in clusters)
    # mask logits of a half-trained MaskScore: positive on the proposal's dominant instance, negative elsewhere, noisy
    dom = torch.cat([(inst[c] == torch.mode(inst[c])[0]).float() for c in clusters])
    mask_logits = ((dom * 2 - 1) * 1.5 + torch.randn(n_rows, generator=g) * 1.5).unsqueeze(1)
    scores = torch.rand(15, generator=g)
    sig = torch.sigmoid(mask_logits).squeeze(-1)
    ious = L.instance_ious(clusters, scores, inst, batch, sig, True)
    ml = L.mask_loss(ious.clone(), clusters, sig, inst, batch)
    sl = L.instance_iou_loss(ious.clone(), clusters, scores, inst, batch, 0.25, 0.75)
    np.savez_compressed(os.path.join(OUT, "mask_loss_cases.npz"), batch=batch.numpy(), inst=inst.numpy(),
                        cluster_offsets=np.cumsum([0] + [len(c) for c in clusters]), cluster_points=torch.cat(clusters).numpy(),
                        mask_logits=mask_logits.numpy(), scores=scores.numpy(), ious=ious.numpy(), mask_loss=ml.numpy(),
                        score_loss=sl.numpy())
    print("mask losses: ious max per proposal", ious.max(1)[0].numpy().round(3).tolist(), "mask_loss", float(ml), "score_loss", float(sl))


def make_nms():
    _stub_modules()
    S = _load(os.path.join(REF, "torch_points3d/models/panoptic/structure_3heads.py"), "ref_structure_3heads")
    g = torch.Generator().manual_seed(1)
    n = 4000
    clusters = []
    for p in range(30):
        c0 = int(torch.randint(0, n - 600, (1,), generator=g))
        size = int(torch.randint(5, 400, (1,), generator=g))
        idx = c0 + torch.randperm(600, generator=g)[:size]
        clusters.append(torch.sort(idx)[0])
    scores = torch.rand(30, generator=g)
    res = S.PanopticResults(semantic_logits=torch.zeros(n, 9), offset_logits=None, embed_logits=None,
                            cluster_scores=scores, mask_scores=None, clusters=clusters, cluster_type=None)
    out = {}
    for tag, kw in [("default", {}), ("tracker", dict(min_cluster_points=10)),
                    ("loose", dict(nms_threshold=0.6, min_cluster_points=0, min_score=0.0))]:
        ids, cl = res.get_instances(**kw)
        out["ids_" + tag] = np.asarray([int(i) for i in ids], np.int64)
        out["sizes_" + tag] = np.asarray([len(c) for c in cl], np.int64)
    # the tracker's painting of the surviving proposals (get_cur_ins_pre_label, panoptic_tracker_pointgroup_npm3d.py:326-337),
    # method extracted with `ast` and fed exactly like track() does (:249-257): with scores (after get_instances with the
    # tracker's min_cluster_points) and without a ScoreNet (scores None: every proposal, proposal order)
    import ast
    tpath = os.path.join(REF, "torch_points3d/metrics/panoptic_tracker_pointgroup_npm3d.py")
    ttree = ast.parse(open(tpath).read())
    tfn = [m for c in ttree.body if isinstance(c, ast.ClassDef) for m in c.body
           if isinstance(m, ast.FunctionDef) and m.name == "get_cur_ins_pre_label"][0]
    tns = {"np": np}
    exec(compile(ast.Module(body=[tfn], type_ignores=[]), tpath, "exec"), tns)
    sem = np.zeros(n, np.int64)
    ids, cl = res.get_instances(min_cluster_points=10)
    c_scores = scores[[int(i) for i in ids]].numpy()
    out["paint_tracker"] = tns["get_cur_ins_pre_label"](None, cl, c_scores, sem).astype(np.int64)
    out["paint_noscore"] = tns["get_cur_ins_pre_label"](None, clusters, None, sem).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "nms_cases.npz"), n=np.int64(n),
                        cluster_offsets=np.cumsum([0] + [len(c) for c in clusters]),
                        cluster_points=torch.cat(clusters).numpy(), scores=scores.numpy(), **out)
    print("nms:", {k: v.tolist() for k, v in out.items() if k.startswith("ids")})


def _hdbscan_canonical_ties(x, min_cluster_size, min_samples, eps):
    """sklearn's own single-linkage / condensed-tree / stability / EOM / epsilon / labelling code (its Cython
    make_single_linkage + tree_to_labels) fed with THE minimum spanning tree of the mutual-reachability graph under the
    strict edge order (weight, min(a,b), max(a,b)) instead of whatever Prim + an unstable argsort leave: removes the
    implementation-defined handling of equal-weight edges, so the result is a function of the data alone."""
    from sklearn.cluster._hdbscan._linkage import MST_edge_dtype, make_single_linkage
    from sklearn.cluster._hdbscan._tree import tree_to_labels
    xd = x.astype(np.float64)
    n = len(xd)
    d2 = np.zeros((n, n))
    for c in range(xd.shape[1]):
        t = xd[:, None, c] - xd[None, :, c]
        d2 += t * t
    core2 = np.sort(d2, axis=1)[:, min_samples - 1]            # the point itself counts (sklearn convention)
    a, b = np.triu_indices(n, 1)
    w2 = np.maximum(np.maximum(core2[a], core2[b]), d2[a, b])
    order = np.lexsort((b, a, w2))
    parent = list(range(n))

    def find(v):
        while parent[v] != v:
            parent[v] = parent[parent[v]]
            v = parent[v]
        return v
    mst = np.empty(n - 1, dtype=MST_edge_dtype)
    k = 0
    for e in order:
        ra, rb = find(int(a[e])), find(int(b[e]))
        if ra != rb:
            parent[ra] = rb
            mst[k] = (int(a[e]), int(b[e]), np.sqrt(w2[e]))
            k += 1
            if k == n - 1:
                break
    labels, _ = tree_to_labels(make_single_linkage(mst), min_cluster_size, "eom", False, eps)
    return labels


def make_hdbscan():
    """hdbscan 0.8.27 (reference: torch_points3d/utils/hdbscan_cluster.py:8-13) is not installed here; sklearn's port of
    the same algorithm (sklearn.cluster.HDBSCAN, core distance counts the point itself) produces the vectors."""
    from sklearn.cluster import HDBSCAN
    rng = np.random.default_rng(15)
    cases, names = {}, []
    for t in range(14):
        n_inst = int(rng.integers(1, 8))
        dim = int(rng.choice([3, 5]))
        centers = rng.normal(0, 3, size=(n_inst, dim))
        sizes = rng.integers(5, 160, size=n_inst)
        x = np.concatenate([c + rng.normal(0, rng.uniform(0.05, 0.6), size=(s, dim)) for c, s in zip(centers, sizes)] +
                           [rng.uniform(-6, 6, size=(int(rng.integers(0, 30)), dim))]).astype(np.float32)
        eps = float([0.006, 0.006, 0.0, 0.3][t % 4])
        lab = HDBSCAN(min_cluster_size=15, min_samples=5, cluster_selection_epsilon=eps, copy=True).fit_predict(
            x.astype(np.float64))
        name = "c%02d" % t
        names.append(name)
        cases["x_" + name], cases["eps_" + name], cases["labels_" + name] = x, np.float64(eps), lab.astype(np.int64)
        cases["canon_" + name] = _hdbscan_canonical_ties(x, 15, 5, eps).astype(np.int64)
    cases["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "hdbscan_cases.npz"), **cases)
    print("hdbscan:", {k: int(cases["labels_" + k].max()) + 1 for k in names})


def make_final_eval():
    """Runs the reference's OWN final_eval (torch_points3d/datasets/panoptic/npm3d.py:107-397) on synthetic label arrays.
    The module itself cannot be imported (torch_geometric, omegaconf, ...), so the function definition is extracted
    from the file with `ast` and executed with numpy/scipy; it only logs its results, so the log lines are parsed."""
    import ast
    import re
    import tempfile
    from scipy import stats
    path = os.path.join(REF, "torch_points3d/datasets/panoptic/npm3d.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "final_eval"][0]
    ns = {"np": np, "stats": stats, "os": os, "write_ply": lambda *a, **k: None, "torch": torch}
    if not hasattr(np, "int"):
        np.int, np.float = int, float  # removed from numpy >= 1.24; the reference was written against 1.19
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    rng = np.random.default_rng(77)
    cases, names = {}, []
    for t in range(4):
        n = 6000
        n_inst = 25
        gt_ins = rng.integers(0, n_inst, size=n)
        thing = np.array([2, 3, 4, 6, 7, 8])
        stuff = np.array([0, 1, 5])
        inst_cls = np.where(rng.random(n_inst) < 0.75, rng.choice(thing, n_inst), rng.choice(stuff, n_inst))
        if t == 3:
            inst_cls[inst_cls == 7] = 6               # a thing class without ground truth
        gt_sem = inst_cls[gt_ins]
        gt_ins_lab = np.where(np.isin(gt_sem, stuff), -1, gt_ins)
        unl = rng.random(n) < 0.03
        gt_sem = np.where(unl, -1, gt_sem)
        gt_ins_lab = np.where(unl, -1, gt_ins_lab)
        pred_sem = np.where(rng.random(n) < 0.85, np.maximum(gt_sem, 0), rng.integers(0, 9, size=n))
        pred_ins = np.where(np.isin(pred_sem, stuff), -1, gt_ins + 100)
        # split / merge / drop some predicted instances
        split = rng.random(n) < 0.2
        pred_ins = np.where((pred_ins >= 0) & split & (gt_ins % 3 == 0), pred_ins + 1000, pred_ins)
        pred_ins = np.where((pred_ins >= 0) & (gt_ins % 7 == 1), 107, pred_ins)
        pred_ins = np.where((pred_ins >= 0) & (gt_ins % 11 == 5), -1, pred_ins)
        pos = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32))
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                ns["final_eval"](pred_sem.copy(), pred_ins.copy(), pred_ins.copy(), pos, gt_sem.copy(), gt_ins_lab.copy())
                log = open("evaluation.txt").read()
            finally:
                os.chdir(cwd)
        name = "e%d" % t
        names.append(name)
        cases["pred_sem_" + name], cases["pred_ins_" + name] = pred_sem.astype(np.int64), pred_ins.astype(np.int64)
        cases["gt_sem_" + name], cases["gt_ins_" + name] = gt_sem.astype(np.int64), gt_ins_lab.astype(np.int64)
        for line in log.splitlines():
            if ":" not in line:
                continue
            key, val = line.split(":", 1)
            nums = [float(v) for v in re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?|nan", val)]
            if nums:
                cases["log_%s_%s" % (name, re.sub(r"[^A-Za-z0-9]+", "_", key.strip()))] = np.asarray(nums)
    cases["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "final_eval_cases.npz"), **cases)
    print("final_eval:", {k: cases[k].tolist() for k in cases if k.startswith("log_e0_") and ("mean" in k or "mIoU" in k or "F1" in k)})



def _parse_log(log, prefix, cases):
    import re
    section = ""
    for line in log.splitlines():
        if line.strip().startswith("Instance Segmentation for"):
            section = "offset_" if "Offset" in line else "embed_"
            continue
        if ":" not in line:
            continue
        key, val = line.split(":", 1)
        val = val.replace("np.float64(", "(")  # numpy 2 prints the scalars of a list with their type
        nums = [float(v) for v in re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?|nan|inf", val)]
        if nums:
            cases["log_%s_%s%s" % (prefix, section, re.sub(r"[^A-Za-z0-9]+", "_", key.strip()))] = np.asarray(nums)


def make_treeins_eval():
    """Runs the reference's OWN FOR-instance final_eval (torch_points3d/datasets/panoptic/treeins.py:99-497: two instance
    predictions, three classes) on synthetic label arrays; the function is extracted with `ast` as for NPM3D and its log
    file is parsed (section headers "for Offset" / "for Embeddings" become key prefixes)."""
    import ast
    import tempfile
    import warnings
    from scipy import stats
    path = os.path.join(REF, "torch_points3d/datasets/panoptic/treeins.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "final_eval"][0]
    ns = {"np": np, "stats": stats, "os": os, "torch": torch}
    if not hasattr(np, "int"):
        np.int, np.float = int, float
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    rng = np.random.default_rng(91)
    cases, names = {}, []
    for t in range(4):
        n = 5000
        n_inst = 18
        gt_ins = rng.integers(0, n_inst, size=n)
        inst_cls = np.where(rng.random(n_inst) < 0.7, 1, 0)          # 1 = tree (thing), 0 = non-tree (stuff)
        if t == 2:
            inst_cls[:] = 1                                           # no stuff instance at all
        gt_sem = inst_cls[gt_ins]
        gt_ins_lab = np.where(gt_sem == 0, -1, gt_ins)
        unl = rng.random(n) < 0.04
        gt_sem = np.where(unl, -1, gt_sem)
        gt_ins_lab = np.where(unl, -1, gt_ins_lab)
        pred_sem = np.where(rng.random(n) < 0.88, np.maximum(gt_sem, 0), rng.integers(0, 2, size=n))
        def noisy(seed_shift):
            r = np.random.default_rng(1000 * t + seed_shift)
            p = np.where(pred_sem == 0, -1, gt_ins + 50)
            p = np.where((p >= 0) & (r.random(n) < 0.25) & (gt_ins % 3 == seed_shift % 3), p + 500, p)
            p = np.where((p >= 0) & (gt_ins % 5 == 1 + seed_shift), 77, p)
            p = np.where((p >= 0) & (gt_ins % 7 == 3 - seed_shift), -1, p)
            return p
        pre_off, pre_emb = noisy(0), noisy(1)
        if t == 3:
            pre_emb = np.full(n, -1)                                  # the embedding branch found nothing
        with tempfile.TemporaryDirectory() as tmp, warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ns["final_eval"](pred_sem.copy(), pre_emb.copy(), pre_off.copy(), gt_sem.copy(), gt_ins_lab.copy(), os.path.join(tmp, "ev"))
            log = open(os.path.join(tmp, "ev.txt")).read()
        name = "t%d" % t
        names.append(name)
        for k, v in (("pred_sem_", pred_sem), ("pre_off_", pre_off), ("pre_emb_", pre_emb), ("gt_sem_", gt_sem), ("gt_ins_", gt_ins_lab)):
            cases[k + name] = v.astype(np.int64)
        _parse_log(log, name, cases)
    cases["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "treeins_eval_cases.npz"), **cases)
    print("treeins final_eval:", {k: cases[k].tolist() for k in cases if k.startswith("log_t0_") and ("mIoU" in k or "F1" in k or "meanPQ" in k)})


def make_tracker_metrics():
    """Runs the reference tracker's OWN per-batch metrics (_compute_acc / _compute_eval,
    torch_points3d/metrics/panoptic_tracker_pointgroup_npm3d.py:678-879; static methods extracted with `ast`, instance_iou =
    the brute-force stand-in of _stub_modules) on synthetic batches: clusters, predicted labels, labels -> the returned scalars."""
    import ast
    _stub_modules()
    path = os.path.join(REF, "torch_points3d/metrics/panoptic_tracker_pointgroup_npm3d.py")
    tree = ast.parse(open(path).read())
    fns = []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in ("_compute_acc", "_compute_eval"):
            node.decorator_list = []
            fns.append(node)
    if not hasattr(np, "int"):
        np.int, np.float = int, float
    ns = {"np": np, "torch": torch, "instance_iou": sys.modules["torch_points_kernels"].instance_iou}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    rng = np.random.default_rng(123)
    cases, names = {}, []
    thing = np.array([2, 3, 4, 6, 7, 8])
    stuff = np.array([0, 1, 5])
    for t in range(5):
        nb = 2 if t < 4 else 1
        n = 1600
        batch = np.sort(rng.integers(0, nb, size=n))
        inst = np.zeros(n, np.int64)
        y = np.zeros(n, np.int64)
        num_instances = []
        for b in range(nb):
            m = np.nonzero(batch == b)[0]
            k = int(rng.integers(4, 9))
            owner = rng.integers(0, k + 2, size=len(m))                 # ids k, k+1 -> no instance (stuff)
            cls_of = rng.choice(thing if t != 3 else thing[:2], k)
            inst[m] = np.where(owner < k, owner + 1, 0)
            y[m] = np.where(owner < k, cls_of[np.minimum(owner, k - 1)], rng.choice(stuff, len(m)))
            num_instances.append(k)
        y = np.where(rng.random(n) < 0.02, -1, y)                         # a few unlabelled points
        pred_labels = np.where(rng.random(n) < 0.85, np.maximum(y, 0), rng.integers(0, 9, size=n))
        # clusters: the ground-truth instances, some split in two, some merged with a neighbour, some dropped, one of pure stuff
        clusters = []
        for b in range(nb):
            for g in range(1, num_instances[b] + 1):
                idx = np.nonzero((batch == b) & (inst == g))[0]
                r = rng.random()
                if r < 0.15 or len(idx) < 4:
                    continue
                if r < 0.4:
                    cut = len(idx) // 3
                    clusters += [idx[:cut], idx[cut:]]
                elif r < 0.55 and clusters and batch[clusters[-1][0]] == b:
                    clusters[-1] = np.concatenate([clusters[-1], idx])
                else:
                    clusters.append(idx[rng.random(len(idx)) < 0.9])
            extra = np.nonzero((batch == b) & (inst == 0))[0][:25]
            if len(extra) > 3:
                clusters.append(extra)
        clusters = [np.sort(c) for c in clusters if len(c)]
        labels = types.SimpleNamespace(instance_labels=torch.from_numpy(inst), y=torch.from_numpy(y),
                                       num_instances=torch.tensor(num_instances))
        cl_t = [torch.from_numpy(c) for c in clusters]
        acc = ns["_compute_acc"](cl_t, torch.from_numpy(pred_labels), labels, torch.from_numpy(batch), labels.num_instances, 0.5)
        ev = ns["_compute_eval"](cl_t, torch.from_numpy(pred_labels), labels, torch.from_numpy(batch), labels.num_instances, 9, 0.5)
        name = "b%d" % t
        names.append(name)
        cases["batch_" + name], cases["inst_" + name], cases["y_" + name] = batch, inst, y
        cases["pred_" + name], cases["num_instances_" + name] = pred_labels, np.asarray(num_instances)
        cases["cl_points_" + name] = np.concatenate(clusters)
        cases["cl_offsets_" + name] = np.concatenate([[0], np.cumsum([len(c) for c in clusters])])
        cases["acc_" + name] = np.asarray([float(v) for v in acc])
        cases["eval_" + name] = np.asarray([float(v) for v in ev])
    cases["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "tracker_metric_cases.npz"), **cases)
    print("tracker metrics:", {n: (cases["acc_" + n].round(4).tolist(), cases["eval_" + n].round(4).tolist()) for n in names})


def make_grid_cylinders():
    """Runs the reference's OWN GridCylinderSampling + CylinderSampling (torch_points3d/core/data_transform/transforms.py:
    182-267, 388-441) on a synthetic rotated strip of points.  The module cannot be imported (torch_geometric, numba, ...),
    so the two class definitions are extracted with `ast` and executed with their real dependencies (sklearn KDTree and
    PCA, installed here: 1.7.2) and a minimal stand-in for torch_geometric's Data container (attribute bag)."""
    import ast
    import itertools
    from sklearn.decomposition import PCA
    from sklearn.neighbors import KDTree

    class Data:
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        @property
        def keys(self):
            return [k for k in self.__dict__ if not k.startswith("_")]

        def __getitem__(self, k):
            return getattr(self, k)

    path = os.path.join(REF, "torch_points3d/core/data_transform/transforms.py")
    tree = ast.parse(open(path).read())
    classes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("GridCylinderSampling", "CylinderSampling")]
    ns = {"np": np, "torch": torch, "KDTree": KDTree, "KDTREE_KEY": "kd_tree", "PCA": PCA, "Data": Data,
          "GridSampling3D_PCA": lambda size: None, "tq": lambda x: x, "itertools": itertools}
    exec(compile(ast.Module(body=classes, type_ignores=[]), path, "exec"), ns)
    rng = np.random.default_rng(31)
    n = 6000
    p = np.stack([rng.uniform(-45, 45, n), rng.uniform(-12, 12, n), rng.normal(0, 1.5, n)], 1)
    p[: n // 6, 1] += 20  # an L-shaped annex, so some grid nodes have no points
    p[: n // 6, 0] = rng.uniform(20, 45, n // 6)
    a = 0.6
    rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    p[:, :2] = p[:, :2] @ rot.T + [130.0, -40.0]
    pos = p.astype(np.float32)
    y = rng.integers(0, 9, n)
    data = Data(pos=torch.from_numpy(pos.copy()), y=torch.from_numpy(y), origin_id=torch.arange(n))
    radius, grid = 8.0, 8.0
    out = ns["GridCylinderSampling"](radius, grid)(data)
    offsets = np.cumsum([0] + [len(d.origin_id) for d in out]).astype(np.int64)
    origin = np.concatenate([d.origin_id.numpy() for d in out])
    centred = np.concatenate([d.pos.numpy() for d in out])
    centre_label = np.array([int(d.center_label) for d in out])
    # centre of every kept cylinder: original minus centred xy of its first point (float32 rounding of the subtraction)
    centres = np.stack([pos[o.origin_id[0], :2].astype(np.float64) - o.pos[0, :2].numpy().astype(np.float64) for o in out])
    np.savez_compressed(os.path.join(OUT, "grid_cylinder_cases.npz"), pos=pos, y=y, radius=radius, grid_size=grid,
                        offsets=offsets, origin=origin.astype(np.int32), centred_pos=centred, centre_label=centre_label,
                        centres=centres)
    print("grid cylinders:", len(out), "kept cylinders,", len(origin), "memberships")


def make_block_merging():
    """Runs the reference's OWN block_merging (torch_points3d/metrics/panoptic_tracker_pointgroup_npm3d.py:339-452) block
    after block on synthetic scenes.  The tracker module cannot be imported (torchnet, torch_geometric, ...), so the method
    is extracted from the file with `ast` and executed as a plain function: `knn` is replaced by the identity assignment
    (origin_sub_ids == originids: every point is its own nearest neighbour, which is what the restatement assumes) and
    `write_ply` by a no-op (the method dumps debugging clouds into ./viz)."""
    import ast
    import tempfile
    path = os.path.join(REF, "torch_points3d/metrics/panoptic_tracker_pointgroup_npm3d.py")
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and any(isinstance(m, ast.FunctionDef) and m.name == "block_merging" for m in n.body)][0]
    fn = [m for m in cls.body if isinstance(m, ast.FunctionDef) and m.name == "block_merging"][0]

    def knn_identity(x, y, k=1):
        assert x.shape[0] == y.shape[0]
        idx = torch.arange(y.shape[0])
        return idx, idx

    ns = {"np": np, "torch": torch, "os": os, "join": os.path.join, "knn": knn_identity, "write_ply": lambda *a, **k: None,
          "normalize": lambda a, axis=0: a}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    ref_block_merging = ns["block_merging"]
    rng = np.random.default_rng(2024)
    cases, names = {}, []
    for t, (n_scene, n_blocks, per_block, k_max) in enumerate([(3000, 6, 1000, 5), (20000, 14, 4000, 25), (400, 3, 200, 0), (12000, 9, 5000, 40)]):
        me = types.SimpleNamespace(_test_area=types.SimpleNamespace(pos=torch.zeros((n_scene, 3))), block_count=0)
        outputs = types.SimpleNamespace()  # no embed_logits / offset_logits: the debugging dumps are skipped
        all_pre = np.full(n_scene, -1, np.int64)
        max_instance = 0
        origins, labels, after, maxes = [], [], [], []
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                for b in range(n_blocks):
                    start = int(b * (n_scene - per_block) / max(n_blocks - 1, 1))
                    origin = np.sort(rng.choice(np.arange(start, start + per_block), size=int(per_block * 0.8), replace=False)).astype(np.int64)
                    lab = np.full(len(origin), -1, np.int64)
                    k = int(rng.integers(0, k_max + 1))
                    if k:
                        cuts = np.sort(rng.choice(len(origin), size=2 * k, replace=False))
                        ids = rng.permutation(k + 2)[:k]  # non-contiguous ids: unused ids still burn a label in the reference
                        for i in range(k):
                            lab[cuts[2 * i]: cuts[2 * i + 1]] = ids[i]
                    out, max_instance = ref_block_merging(me, origin, origin, lab.copy(), all_pre.copy(), max_instance, 0.01, outputs, None)
                    all_pre = out.numpy().copy() if torch.is_tensor(out) else np.asarray(out).copy()
                    max_instance = int(max_instance)
                    origins.append(origin)
                    labels.append(lab)
                    after.append(all_pre.copy())
                    maxes.append(max_instance)
            finally:
                os.chdir(cwd)
        name = "m%d" % t
        names.append(name)
        cases["n_scene_" + name] = np.int64(n_scene)
        cases["block_offsets_" + name] = np.cumsum([0] + [len(o) for o in origins]).astype(np.int64)
        cases["origin_" + name] = np.concatenate(origins)
        cases["labels_" + name] = np.concatenate(labels)
        cases["after_" + name] = np.stack(after)           # scene labels after every block
        cases["max_instance_" + name] = np.asarray(maxes, np.int64)
    cases["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "block_merging_cases.npz"), **cases)
    print("block merging:", {n: int(cases["max_instance_" + n][-1]) for n in names})


def make_proposals():
    """Proposal generation of the reference model: PointGroup3heads._cluster / _cluster2 / _cluster3 / _cluster4 / _cluster5 /
    _cluster6 (torch_points3d/models/panoptic/PointGroup3heads.py:163-390) extracted with `ast` and executed here, with the
    reference's OWN utils/meanshift_cluster.py (sklearn MeanShift workers) behind them.  torch_points_kernels.region_grow
    (absent library) is the one stand-in: the CPU oracle's restatement with tpk's defaults (nsample 16) -- its own output is
    part of the fixture only through these functions.  Pins: which coordinates / nsample every call uses, the order in which
    the groups of proposals are concatenated, the cluster_type codes (incl. _cluster2's quirk: the votes get type 1 only if
    the position clusters are not empty), the per-sample mean shift with its `> 3 points` rule."""
    import ast
    os.environ["OMP_NUM_THREADS"] = "1"  # the reference forks a worker Pool: no OpenMP team may exist in this process before
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import oracle as O
    ms_ref = _load(os.path.join(REF, "torch_points3d/utils/meanshift_cluster.py"), "ref_meanshift_cluster")
    sys.modules["ref_meanshift_cluster"] = ms_ref  # its Pool workers are pickled by reference
    path = os.path.join(REF, "torch_points3d/models/panoptic/PointGroup3heads.py")
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "PointGroup3heads"][0]
    wanted = ("_cluster", "_cluster2", "_cluster3", "_cluster4", "_cluster5", "_cluster6")
    fns = [m for m in cls.body if isinstance(m, ast.FunctionDef) and m.name in wanted]

    def region_grow(pos, labels, batch, ignore_labels=[], radius=0.02, nsample=16, min_cluster_size=32):
        cl, _ = O.region_grow(pos.numpy(), labels.numpy(), batch.numpy(), [int(v) for v in ignore_labels], nsample=nsample,
                              radius=float(radius), min_cluster_size=min_cluster_size)
        return [torch.from_numpy(c) for c in cl]

    ns = {"torch": torch, "np": np, "region_grow": region_grow, "meanshift_cluster": ms_ref}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    rng = np.random.default_rng(31)
    cases, names = {}, []
    for t, (n, n_inst, sparse_pos) in enumerate([(5000, 14, False), (3000, 8, True)]):
        inst = rng.integers(0, n_inst, size=n)
        batch = np.sort(rng.integers(0, 2, size=n)).astype(np.int64)
        inst_cls = np.array([2, 3, 4, 6, 7, 8, 0, 1, 5, 2, 3, 4, 6, 7])[:n_inst]
        pred = inst_cls[inst].astype(np.int64)
        cen = rng.uniform(-6, 6, size=(n_inst, 3))
        spread = 2.5 if sparse_pos else 0.45      # sparse_pos: raw positions too far apart for region growing at nsample 16
        pos = (cen[inst] + rng.normal(0, spread, size=(n, 3)) + batch[:, None] * 0.01).astype(np.float32)
        off = (cen[inst] - pos + rng.normal(0, 0.03, size=(n, 3))).astype(np.float32)
        e_inst = rng.normal(0, 2.0, size=(n_inst, 5))
        emb = (e_inst[inst] + rng.normal(0, 0.12, size=(n, 5))).astype(np.float32)
        sem = np.full((n, 9), -5.0, np.float32)
        sem[np.arange(n), pred] = 5.0
        me = types.SimpleNamespace(raw_pos=torch.from_numpy(pos), input=types.SimpleNamespace(batch=torch.from_numpy(batch)),
                                   device="cpu", _stuff_classes=torch.tensor([0, 1, 5]),
                                   opt=types.SimpleNamespace(cluster_radius_search=0.3, bandwidth=0.6))
        name = "p%d" % t
        names.append(name)
        for k, v in dict(pos=pos, off=off, emb=emb, pred=pred, batch=batch, sem=sem).items():
            cases["%s_%s" % (k, name)] = v
        cases["radius_" + name], cases["bandwidth_" + name] = np.float64(0.3), np.float64(0.6)
        for fn_name in wanted:
            if fn_name in ("_cluster3", "_cluster4"):   # (semantic_logits, embed_logits): mean shift alone / raw positions + mean shift
                args = [torch.from_numpy(sem), torch.from_numpy(emb)]
            else:
                args = [torch.from_numpy(sem), torch.from_numpy(off)] + ([torch.from_numpy(emb)] if fn_name in ("_cluster5", "_cluster6") else [])
            cl, ct = ns[fn_name](me, *args)
            tag = "%s%s" % (name, fn_name)
            cases["offsets_" + tag] = np.cumsum([0] + [len(c) for c in cl]).astype(np.int64)
            cases["points_" + tag] = torch.cat(cl).numpy().astype(np.int64) if cl else np.zeros(0, np.int64)
            cases["types_" + tag] = np.asarray(ct.cpu().numpy() if torch.is_tensor(ct) else ct, np.uint8)
            print("proposals", tag, len(cl), "clusters, types", np.bincount(cases["types_" + tag], minlength=3).tolist())
    cases["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "proposal_cases.npz"), **cases)


def make_embed_recipes():
    """PointGroupEmbed._cluster .. _cluster16 (torch_points3d/models/panoptic/pointgroupembed.py:219-783) extracted with `ast`
    and executed with recording stand-ins for the clustering primitives (the hdbscan package is absent; the stand-ins return
    one tagged single-point proposal per (call, loop index) so that the order of the final list is visible).  Pins, per
    cluster_type: which primitive runs on which feature matrix (xyz / embeddings / both) with which numeric arguments, in
    which order, how the proposal lists are concatenated and which type codes they get.  The reference's
    meanshift_cluster.cluster_loop raises TypeError when really executed (it calls meanshift_cluster without its bandwidth,
    :57); the stand-in records the call instead -- the fixture marks those cluster types."""
    import ast
    import json
    path = os.path.join(REF, "torch_points3d/models/panoptic/pointgroupembed.py")
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "PointGroupEmbed"][0]
    wanted = ["_cluster"] + ["_cluster%d" % i for i in range(2, 17)]
    fns = [m for m in cls.body if isinstance(m, ast.FunctionDef) and m.name in wanted]
    n = 40
    rng = np.random.default_rng(5)
    pos = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32))
    emb = torch.from_numpy(rng.normal(size=(n, 5)).astype(np.float32))
    pred = np.array([2, 3, 4, 6, 7, 8, 0, 1, 5, 2] * 4)
    sem = np.full((n, 9), -5.0, np.float32)
    sem[np.arange(n), pred] = 5.0
    thing = ~np.isin(pred, [0, 1, 5])
    trace = []

    def feat_name(x):
        x = x.detach().cpu()
        for name, ref in (("xyz", pos[thing]), ("emb", emb[thing]), ("all", torch.cat((pos[thing], emb[thing]), 1)),
                          ("raw_pos(all points)", pos)):
            if x.shape == ref.shape and torch.equal(x, ref):
                return name
        raise AssertionError("unknown feature matrix %s" % (tuple(x.shape),))

    def tagged(count):
        base = 1000 * len(trace)
        return [torch.tensor([base + i]) for i in range(count)]

    class H:
        @staticmethod
        def cluster_single(x, unique_in_batch, label_batch, local_ind, type):
            trace.append(["hdbscan.cluster_single", feat_name(x), int(type)])
            return tagged(1), [type]

        @staticmethod
        def cluster_loop(x, unique_in_batch, label_batch, local_ind, low, high, loop_num):
            trace.append(["hdbscan.cluster_loop", feat_name(x), int(low), int(high), int(loop_num)])
            return tagged(loop_num), list(range(loop_num))

        @staticmethod
        def cluster_loop_fixedD(x, unique_in_batch, label_batch, local_ind, low, high, loop_num):
            trace.append(["hdbscan.cluster_loop_fixedD", feat_name(x), int(low), int(high), int(loop_num)])
            return tagged(loop_num), list(range(loop_num))

    class M:
        @staticmethod
        def cluster_single(x, unique_in_batch, label_batch, local_ind, type, bandwidth):
            trace.append(["meanshift.cluster_single", feat_name(x), int(type), "bandwidth=%g" % bandwidth])
            return tagged(1), [type]

        @staticmethod
        def cluster_loop(x, unique_in_batch, label_batch, local_ind, low, high, loop_num):
            trace.append(["meanshift.cluster_loop", feat_name(x), int(low), int(high), int(loop_num)])
            return tagged(loop_num), list(range(loop_num))

    def region_grow(p, labels, batch, ignore_labels=[], radius=0.02, nsample=16, min_cluster_size=32):
        trace.append(["region_grow", feat_name(p), "ignore=%s" % sorted(int(v) for v in ignore_labels), "radius=%g" % radius,
                      "nsample=%d" % nsample, "min_cluster_size=%d" % min_cluster_size])
        return tagged(2)

    ns = {"torch": torch, "np": np, "region_grow": region_grow, "hdbscan_cluster": H, "meanshift_cluster": M}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    me = types.SimpleNamespace(raw_pos=pos, input=types.SimpleNamespace(batch=torch.zeros(n, dtype=torch.long)), device="cpu",
                               _stuff_classes=torch.tensor([0, 1, 5]),
                               opt=types.SimpleNamespace(cluster_radius_search=0.3, bandwidth=0.6))
    out = {}
    for i, name in enumerate(wanted):
        del trace[:]
        cl, ct = ns[name](me, torch.from_numpy(sem), emb)
        out[str(i + 1)] = {"calls": [list(c) for c in trace], "proposal_tags": [int(c[0]) for c in cl],
                           "types": [int(v) for v in (ct.tolist() if torch.is_tensor(ct) else ct)],
                           "raises_in_the_reference": any(c[0] == "meanshift.cluster_loop" for c in trace)}
        print("embed recipe", i + 1, out[str(i + 1)])
    with open(os.path.join(OUT, "embed_cluster_recipes.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    if "--grid-only" in sys.argv:
        make_grid_cylinders()
        sys.exit(0)
    if "--proposals-only" in sys.argv:
        if os.environ.get("OMP_NUM_THREADS") != "1":  # the reference forks a worker Pool: torch / BLAS must not hold OpenMP teams
            os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1",
                                                                         OPENBLAS_NUM_THREADS="1"))
        make_proposals()
        sys.exit(0)
    if "--embed-recipes-only" in sys.argv:
        make_embed_recipes()
        sys.exit(0)
    if "--mask-losses-only" in sys.argv:
        make_mask_losses()
        sys.exit(0)
    if "--nms-only" in sys.argv:
        make_nms()
        sys.exit(0)
    if "--block-merging-only" in sys.argv:
        make_block_merging()
        sys.exit(0)
    if "--final-eval-only" in sys.argv:
        make_final_eval()
        sys.exit(0)
    if "--treeins-eval-only" in sys.argv:
        make_treeins_eval()
        sys.exit(0)
    if "--tracker-metrics-only" in sys.argv:
        make_tracker_metrics()
        sys.exit(0)
    make_hdbscan()
    if "--hdbscan-only" in sys.argv:
        sys.exit(0)
    make_meanshift()
    make_losses()
    make_mask_losses()
    make_nms()
    make_final_eval()
    make_treeins_eval()
    make_tracker_metrics()
    make_grid_cylinders()
    make_block_merging()
    print("proposal_cases.npz: run `python tests/golden/make_golden.py --proposals-only` (single-threaded environment)")
# tests/golden/ref_written_npm3d_like.ply (+ _values.npz): 50 vertices written by the reference's own
# torch_points3d/models/panoptic/ply.py:write_ply (fields x, y, z, scalar_class, scalar_label as float32, the way
# CloudCompare exports NPM3D) -- generated once with the snippet in the commit that added panopticsegforlargescalepointcloud_amd/io.py.
    make_embed_recipes()
