#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: end-to-end points/sec of the panoptic hot path
(hash + kernel maps -> sparse U-Net forward -> heads -> region growing + mean shift -> ScorerUnet -> NMS / instance
labels) on the synthetic 10M-point urban scene cut into 64 overlapping cylinder tiles (configs[3]).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over the whole scene: every rank runs its shard of the 64 tiles (tiles_per_batch
cylinders per launch sequence) and, for N > 1, ONE RCCL all-gather of the per-tile results (origin id, instance label and
the 9 semantic log-probabilities per point: what the tracker's scene assembly consumes) closes the step (strong scaling:
the scene is fixed, tiles are sharded).  Inputs (and the synthetic head statistics used for grouping,
see DESIGN.md "what is measured") are resident in HBM before the timed region.  Random-init weights, synthetic data.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel pp_spconv_fwd, HIP
events on the launch stream, algorithmic bytes/flops per SURVEY.md 8d) and `cpu_baseline` (CPU oracle + sklearn
MeanShift on a bounded sample; N == 1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s   MI355X_MICROARCH.md chip-level parameters
FP32_MFMA_PEAK = 157.3e12  # FLOP/s (v_mfma_f32_16x16x4_f32 = fp32 vector rate)
BF16_MFMA_PEAK = 2.5e15    # FLOP/s dense (v_mfma_f32_16x16x32_bf16; MI355X_MICROARCH.md)
X3_PRODUCTS = 6            # bf16 products k_spconv_x3 executes per fp32 product (csrc/pp_spconv3.hip)


def pipe_seconds(family, flops):
    """time the matrix pipe a kernel family runs on needs for `flops` algorithmic fp32 flops at its dense peak: the split-operand
    kernel executes 6 bf16 products per fp32 product on the 2.5 PFLOP/s pipe (ceiling 417 TFLOP/s fp32-equivalent), the fp32-MFMA
    kernels one product on the 157.3 TFLOP/s pipe"""
    return X3_PRODUCTS * flops / BF16_MFMA_PEAK if family in ("x3", "x3f", "x3_all") else flops / FP32_MFMA_PEAK


def env_int(name, default):
    return int(os.environ.get(name, default))


def build_scene(total_points, grid, voxel, seed):
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    # overlapping cylinders feed each scene voxel ~2x: size the scene so the tiles sum to ~total_points
    target = total_points / 2.0
    for _ in range(4):
        scene = syn.urban_scene(int(target), voxel=voxel, seed=seed)
        tiles, radius = syn.cylinder_tiles(scene, grid)
        fed = sum(len(t) for t in tiles)
        if abs(fed - total_points) / total_points <= 0.02:
            break
        target *= total_points / fed
    return scene, tiles, radius


def build_model(device, voxel):
    from panopticsegforlargescalepointcloud_amd.config import load_model_config
    from panopticsegforlargescalepointcloud_amd.panoptic import PointGroup3heads
    from panopticsegforlargescalepointcloud_amd import synthetic as syn

    class DS:
        feature_dimension = 4
        num_classes = syn.NPM3D_NUM_CLASSES
        stuff_classes = torch.tensor(syn.NPM3D_STUFF)

    cfg = load_model_config(os.path.join(ROOT, "conf", "panoptic_3heads.yaml"), "PointGroup-PAPER", data={"grid_size": voxel})
    torch.manual_seed(2022)  # Trainer.set_seed, torch_points3d/trainer.py:278-282
    model = PointGroup3heads(cfg, "dummy", DS, None)
    # non-trivial BN statistics so folded scale/shift are not the identity
    g = torch.Generator().manual_seed(7)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.05)
            m.running_var.copy_(1.0 + 0.1 * torch.rand(m.running_var.shape, generator=g))
    # random-init ScorerHead gives sigmoid(~0) ~ 0.5: lift the bias so proposals clear the score > 0.5 filter and the
    # NMS / painting / exchange stages see realistic work (documented in DESIGN.md "what is measured")
    with torch.no_grad():
        model.ScorerHead[0].bias.fill_(1.0)
    return model.to(device).eval(), cfg, DS


def host_cpu_state():
    """(process CPU seconds, cgroup throttled microseconds, throttled periods, hipMalloc calls of the caching allocator) --
    diagnostics for bench.py's config.host"""
    t = time.process_time()
    usec = n = 0
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            if k == "throttled_usec":
                usec = int(v)
            elif k == "nr_throttled":
                n = int(v)
    except Exception:
        pass
    segs = torch.cuda.memory_stats().get("segment.all.allocated", 0) if torch.cuda.is_available() else 0
    return t, usec, n, segs


def host_threads():
    """(threads the OpenMP oracle runs with, CPUs the container's cgroup quota allows or None) -- the oracle is throttled,
    not sped up, by more threads than the quota"""
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(round(int(q) / int(period))))
    except Exception:
        pass
    import ctypes
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
        if "OMP_NUM_THREADS" not in os.environ and quota is not None and quota < (os.cpu_count() or 1):
            gomp.omp_set_num_threads(quota)
        return int(gomp.omp_get_max_threads()), quota
    except Exception:
        return os.cpu_count(), quota


def cpu_baseline(model, cfg, DS, scene, tiles, voxel, repeats=5, warmups=2, n_tiles=16, net_tiles=4):
    """CPU baseline per SURVEY.md 8d on a batch of `n_tiles` tiles around the median size (reported baseline only):
      (1) embedding clustering = the reference's own dependency and fan-out: sklearn MeanShift(bin_seeding) on ONE sample
          per process, min(n_tiles, cores) SPAWNED processes, `pool.map` over the batch's samples exactly as
          torch_points3d/utils/meanshift_cluster.py:9-18,96-101 does (the pool persists across passes: process start-up is
          not charged);
      (2) sparse U-Net + heads + region growing + ScorerUnet = the C/OpenMP restatement (oracle/) on all the threads the
          container's CPU quota grants, timed on `net_tiles` tiles of the batch (its smallest ... largest, the median tile among
          them) and scaled by the batch's point count (the restatement is parallel over rows, so a 16-tile batch costs 16
          tiles' worth of the same passes); the median-tile-only estimate of earlier rounds is printed beside it.
    Both parts: `warmups` untimed passes, then the median of `repeats`.  value = points of the batch / (network time for
    the batch + mean-shift fan-out time) -- the reference runs the two one after the other inside forward().
    Also returns what the self-check needs to compare the GPU path with the oracle on the median tile."""
    from oracle import pipeline as opipe
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    order = np.argsort([len(x) for x in tiles])
    mid = len(tiles) // 2
    lo = max(0, min(mid - n_tiles // 2, len(tiles) - n_tiles))
    chosen = [int(t) for t in order[lo: lo + n_tiles]]
    t_med = int(order[mid])  # median-size tile
    b = syn.tile_batch(scene, tiles, [t_med])
    cls, off, emb = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(99))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    opt = {"cluster_radius_search": cfg.cluster_radius_search, "cluster_type": cfg.cluster_type, "bandwidth": cfg.bandwidth}
    try:
        import sklearn  # noqa: F401
        use_sk = True
    except Exception:
        use_sk = False
    threads, quota = host_threads()
    ignore = [-1] + [int(c) for c in syn.NPM3D_STUFF]
    # ---- (1) the embedding clustering of the batch, fanned out over processes
    samples, n_batch = [], 0
    for t in chosen:
        bt = b if t == t_med else syn.tile_batch(scene, tiles, [t])
        c_t, _, e_t = (cls, off, emb) if t == t_med else syn.synthetic_head_outputs(scene, bt["origin_id"], 0.0, np.random.default_rng(99))
        samples.append((np.ascontiguousarray(e_t[~np.isin(c_t, ignore)]), float(cfg.bandwidth)))
        n_batch += len(bt["pos"])
    procs = max(1, min(len(samples), threads))
    ms_times, ms_labels = [], None
    if use_sk:
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(processes=procs) as pool:
            for it in range(warmups + repeats):
                t0 = time.perf_counter()
                ms_labels = pool.map(opipe.sklearn_meanshift_labels, samples)
                ms_times.append(time.perf_counter() - t0)
    else:  # no sklearn on the box: the oracle's own mean shift, one sample after the other
        from oracle import oracle as O
        for it in range(warmups + repeats):
            t0 = time.perf_counter()
            ms_labels = [O.meanshift(x, [0, len(x)], bw)[0] for x, bw in samples]
            ms_times.append(time.perf_counter() - t0)
    t_ms = float(np.median(ms_times[warmups:]))
    # mean-shift proposals of the median tile, in the reference's order (labels ascending), for the network passes
    j = chosen.index(t_med)
    li = np.nonzero(~np.isin(cls, ignore))[0]
    ms_clusters = [li[ms_labels[j] == l] for l in np.unique(ms_labels[j]) if l != -1] if len(li) > 3 else []
    # ---- (2) U-Net + heads + region growing + scorer: the median tile (2 warm-ups + median of `repeats`, also what the self-check
    # compares the GPU path with) AND `net_tiles` - 1 more tiles spread over the batch's size range (1 warm-up + median of 3), so
    # that the batch figure is a sum over tiles of different sizes, not one tile scaled by 16
    def net_pass(bt, heads, clusters):
        timings = {}
        opipe.CONV_STATS["flops"] = opipe.CONV_STATS["seconds"] = 0.0
        t0 = time.perf_counter()
        out = opipe.forward(sd, bt, opt, DS.num_classes, syn.NPM3D_STUFF, override=heads, use_sklearn_meanshift=use_sk,
                            timings=timings, ms_clusters=clusters)
        want = opipe.instance_labels(out, len(bt["pos"]), bt["batch"])
        return time.perf_counter() - t0, timings, dict(opipe.CONV_STATS), out, want

    runs = []
    for it in range(warmups + repeats):
        runs.append(net_pass(b, (cls, off, emb), ms_clusters))
    out, want_labels = runs[-1][3], runs[-1][4]
    runs = sorted(runs[warmups:], key=lambda r: r[0])
    dt, timings, conv = runs[len(runs) // 2][:3]
    n = len(b["pos"])
    timed = [(t_med, n, dt)]
    others = [chosen[i] for i in np.linspace(0, len(chosen) - 1, max(net_tiles, 1)).round().astype(int)] if net_tiles > 1 else []
    for t in dict.fromkeys(others):  # (distinct, order kept)
        if t == t_med or len(timed) >= net_tiles:
            continue
        bt = syn.tile_batch(scene, tiles, [t])
        c_t, o_t, e_t = syn.synthetic_head_outputs(scene, bt["origin_id"], 0.0, np.random.default_rng(99))
        jt = chosen.index(t)
        li_t = np.nonzero(~np.isin(c_t, ignore))[0]
        cl_t = [li_t[ms_labels[jt] == l] for l in np.unique(ms_labels[jt]) if l != -1] if len(li_t) > 3 else []
        tt = sorted(net_pass(bt, (c_t, o_t, e_t), cl_t)[0] for _ in range(1 + 3))  # (the first pass is the slowest or close to it)
        timed.append((t, len(bt["pos"]), tt[1]))
    n_timed = sum(v[1] for v in timed)
    t_net_batch = sum(v[2] for v in timed) * n_batch / n_timed   # the timed tiles' passes, scaled by points to the batch
    t_net_one = dt * n_batch / n                                  # (round 5's estimate: the median tile alone, scaled)
    timings = dict(timings)
    timings["meanshift"] = t_ms  # the fan-out over the whole batch (the per-tile passes reuse its result)
    res = {"value": n_batch / (t_net_batch + t_ms), "unit": "points/sec", "cores": threads, "host_cpus": os.cpu_count(),
           "cgroup_cpu_quota": quota, "kind": "port", "tiles_timed_meanshift": len(chosen), "tiles_timed_network": len(timed),
           "network_tiles": [{"tile": int(t), "voxels": int(nv), "s_per_pass": round(sec, 3)} for t, nv, sec in timed],
           "network_scaled_to_tiles": len(chosen), "meanshift_processes": procs,
           "value_from_median_tile_only": n_batch / (t_net_one + t_ms),
           "sample": "batch of %d of %d tiles around the median size (%d voxels): %s MeanShift one sample per spawned process on %d "
                     "processes (pool.map, %.2f s per batch) + C/OpenMP oracle U-Net+heads+region_grow+scorer on %d threads "
                     "timed on %d tiles of the batch (%d voxels, smallest to largest incl. the median tile: %.2f s per pass in all, "
                     "scaled by points to the batch: %.1f s; the median tile alone scaled: %.1f s); median tile %d warm-ups + median "
                     "of %d passes, the other tiles 1 + median of 3" % (
                         len(chosen), len(tiles), n_batch, "sklearn" if use_sk else "oracle", procs, t_ms, threads, len(timed),
                         n_timed, sum(v[2] for v in timed), t_net_batch, t_net_one, warmups, repeats),
           "stages_s": {k: round(v, 3) for k, v in timings.items()}}
    res["conv_GFLOPs"] = round(conv["flops"] / max(conv["seconds"], 1e-9) / 1e9, 1)  # sparse convolutions of both U-Nets
    return res, (b, (cls, off, emb), out, want_labels)


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= atol + rtol * np.abs(b)))


def _canonical(labels):
    """instance ids renumbered by first appearance (-1 kept): label permutations compare equal"""
    labels = np.asarray(labels)
    out = np.full(labels.shape, -1, np.int64)
    m = labels >= 0
    if m.any():
        _, first, inv = np.unique(labels[m], return_index=True, return_inverse=True)
        order = np.argsort(np.argsort(first))
        out[m] = order[inv]
    return out


def _scaled_err(a, b):
    """max |a - b| as a fraction of b's magnitude (>= 1): the 1e-4 fp32 bar of north_star"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max())) if a.size else 0.0


def _near_tie_points(clusters, scores, n_points, nms_threshold=0.3, eps=1e-5, min_score=0.5, other_scores=None):
    """points whose instance label hangs on a score comparison closer than eps: two proposals overlapping by more than the NMS
    threshold (region growing and mean shift propose every object twice, with nearly the same points -- their max-pooled
    scorer features, hence scores, agree to the last bits) or a score within eps of the score filter.  Which of such twins
    survives is decided by float rounding, legitimately differently in two correct implementations.
    other_scores: the second implementation's scores of the same proposals.  A pair whose scores are EXACTLY equal in both
    (twins whose differing points never attain a channel's maximum: identical pooled features) is not ambiguous -- equal scores
    are ordered by the same rule (descending index) in both, so their labels must agree and stay in the comparison."""
    amb = np.zeros(n_points, bool)
    scores = np.asarray(scores, np.float64)
    other = None if other_scores is None else np.asarray(other_scores, np.float64)
    owner = {}
    for i, c in enumerate(clusters):
        for p in np.asarray(c).tolist():
            owner.setdefault(p, []).append(i)
    inter = {}
    for lst in owner.values():
        for a in range(len(lst)):
            for b in range(a + 1, len(lst)):
                inter[(lst[a], lst[b])] = inter.get((lst[a], lst[b]), 0) + 1
    for (i, j), n in inter.items():
        if n / (len(clusters[i]) + len(clusters[j]) - n) > nms_threshold and abs(scores[i] - scores[j]) < eps:
            if other is not None and scores[i] == scores[j] and other[i] == other[j]:
                continue
            amb[np.asarray(clusters[i])] = True
            amb[np.asarray(clusters[j])] = True
    for i in np.nonzero(np.abs(scores - min_score) < eps)[0]:
        amb[np.asarray(clusters[i])] = True
    return amb


class _SpreadScorerHead:
    """with _SpreadScorerHead(lin, scores): ... -- rescales the ScorerHead's Linear layer from the scores of a first pass so that
    the proposals' logits spread to mean 2.1 / unit variance (scores over roughly (0.55, 0.99)).  A random-init head squeezes
    every score into a ~1e-3 band, where a quarter of a tile's points hang on score comparisons closer than 1e-5 by CHANCE;
    with the logits spread, only the genuine twins (two proposals with nearly the same points, whose max-pooled features agree
    to the last bits) stay near-tied.  Same proposals, same scorer features; the layer is restored on exit."""

    def __init__(self, lin, scores):
        self.lin, self.w0, self.b0 = lin, lin.weight.detach().clone(), lin.bias.detach().clone()
        sc = scores.detach().double().clamp(1e-9, 1 - 1e-9)
        z = torch.log(sc / (1 - sc)) - float(self.b0[0])
        self.alpha = 1.0 / max(float(z.std()), 1e-9)
        self.beta = 2.1 - self.alpha * float(z.mean())

    def __enter__(self):
        with torch.no_grad():
            self.lin.weight.copy_(self.w0 * self.alpha)
            self.lin.bias.fill_(self.beta)
        return self

    def __exit__(self, *exc):
        with torch.no_grad():
            self.lin.weight.copy_(self.w0)
            self.lin.bias.copy_(self.b0)
        return False

    def oracle_scores(self, cluster_features):
        """the oracle's scores under the rescaled head: sigmoid(features . w + b) as oracle/pipeline.py computes them"""
        w = (self.w0 * self.alpha).detach().cpu().numpy().astype(np.float32)
        z = (np.asarray(cluster_features) @ w.T + np.float32(self.beta))[:, 0]
        return 1.0 / (1.0 + np.exp(-z))


NEAR_TIE_BOUND = 0.10  # share of a tile's points that may hang on a score near-tie (< 1e-5) before the label check FAILS


def self_check(runner, batches, device, oracle_case):
    """Untimed correctness checks of the benchmark's own run (the 10 M-row kernel variants are not reachable from the
    unit tests' sizes): (1) batch invariance -- a tile of the 64-tile batch gives the same result as that tile run alone
    (instance labels bit-exact, semantic / embedding outputs 1e-4); (2) the median tile run alone equals the CPU oracle
    pipeline (proposals bit-exact, scores / embeddings / semantic log-probabilities within 1e-4 of their magnitude,
    instance labels equal after canonicalisation -- the oracle's own scores, no substitution; points whose label hangs on
    a score near-tie below 1e-5 between two overlapping proposals are excluded and counted).  The measured errors are
    reported in `checks["max_err"]`."""
    checks = {}
    ids, dev_b, override, starts, _ = batches[0]
    labels, res, counts = runner.run(dev_b, len(ids), override=override)
    j = int(np.argsort(np.diff(starts))[len(ids) // 2])
    lo, hi = int(starts[j]), int(starts[j + 1])
    one = {k: (v[lo:hi] if k != "batch" else torch.zeros(hi - lo, dtype=v.dtype, device=device)) for k, v in dev_b.items()}
    ov1 = tuple(o[lo:hi] for o in override)
    l1, r1, c1 = runner.run(one, 1, override=ov1)
    ok_lab = bool(torch.equal(labels[lo:hi], l1)) and counts[j] == c1[0]
    e_sem = _scaled_err(res.semantic_logits[lo:hi].cpu().numpy(), r1.semantic_logits.cpu().numpy())
    e_emb = _scaled_err(res.embed_logits[lo:hi].cpu().numpy(), r1.embed_logits.cpu().numpy())
    ok_sem, ok_emb = e_sem < 1e-4, e_emb < 1e-4
    checks["batch_invariance"] = "pass" if (ok_lab and ok_sem and ok_emb) else \
        "FAIL(labels=%s sem=%s emb=%s)" % (ok_lab, ok_sem, ok_emb)
    checks["batch_invariance_tile"] = {"tile": int(ids[j]), "rows": hi - lo, "instances": int(c1[0])}
    checks["max_err"] = {"batch_vs_alone_semantic": e_sem, "batch_vs_alone_embeddings": e_emb}
    if oracle_case is None:
        checks["oracle"] = "skipped (--no-cpu-baseline)"
    else:
        b, ov, want, want_labels = oracle_case
        dev_one = {k: torch.from_numpy(v).to(device) for k, v in b.items()}
        lg, rg, cg = runner.run(dev_one, 1, override=tuple(torch.from_numpy(a).to(device) for a in ov))
        got = [c.cpu().numpy() for c in rg.clusters_csr.to_list()]
        ok_prop = len(got) == len(want["clusters"]) and all(np.array_equal(g, np.sort(w)) for g, w in zip(got, want["clusters"]))
        e_score = _scaled_err(rg.cluster_scores.cpu().numpy(), want["cluster_scores"]) if ok_prop else float("nan")
        e_feat = _scaled_err(rg.embed_logits.cpu().numpy(), want["embed_logits"])
        e_seml = _scaled_err(rg.semantic_logits.cpu().numpy(), want["semantic_logits"])
        ok_score, ok_feat = ok_prop and e_score < 1e-4, e_feat < 1e-4 and e_seml < 1e-4
        # instance labels against the oracle's OWN scores and NMS, with the ScorerHead's logits spread (second pass of both
        # paths: same proposals, same scorer features) so that only genuine twins stay near-tied; those points are left out,
        # reported as a fraction, and the check FAILS above NEAR_TIE_BOUND
        amb = np.zeros(len(b["pos"]), bool)
        ok_inst, e_spread, frac_unspread = False, float("nan"), float("nan")
        if ok_prop:
            from oracle import pipeline as opipe
            frac_unspread = float(_near_tie_points(want["clusters"], want["cluster_scores"], len(b["pos"])).mean())
            with _SpreadScorerHead(runner.model.ScorerHead[0], rg.cluster_scores) as sp:
                lg, rg2, cg = runner.run(dev_one, 1, override=tuple(torch.from_numpy(a).to(device) for a in ov))
                want_sp = dict(want)
                want_sp["cluster_scores"] = sp.oracle_scores(want["cluster_features"])
            e_spread = _scaled_err(rg2.cluster_scores.cpu().numpy(), want_sp["cluster_scores"])
            want_labels = opipe.instance_labels(want_sp, len(b["pos"]), b["batch"])
            amb = _near_tie_points(want_sp["clusters"], want_sp["cluster_scores"], len(b["pos"]),
                                   other_scores=rg2.cluster_scores.cpu().numpy())
            ok_inst = bool(np.array_equal(_canonical(lg.cpu().numpy()[~amb]), _canonical(want_labels[~amb]))) \
                and e_spread < 1e-4 and amb.mean() <= NEAR_TIE_BOUND
        checks["oracle"] = "pass" if (ok_prop and ok_score and ok_feat and ok_inst) else \
            "FAIL(proposals=%s scores=%s embeddings=%s instances=%s)" % (ok_prop, ok_score, ok_feat, ok_inst)
        checks["oracle_tile"] = {"rows": len(b["pos"]), "proposals": len(got), "instances": int(cg[0]),
                                 "points_on_score_near_ties": int(amb.sum()), "near_tie_fraction": round(float(amb.mean()), 4),
                                 "near_tie_bound": NEAR_TIE_BOUND, "near_tie_fraction_random_head": round(frac_unspread, 4),
                                 "score_spread": "ScorerHead logits rescaled to mean 2.1 / unit variance for the label check"}
        checks["max_err"].update({"oracle_scores": e_score, "oracle_scores_spread_head": e_spread, "oracle_embeddings": e_feat,
                                  "oracle_semantic": e_seml})
    checks["split_operand_accuracy"] = split_operand_accuracy(device, checks["max_err"])
    checks["all"] = "pass" if all(not str(v).startswith("FAIL") for v in checks.values()) else "FAIL"
    return checks


def split_operand_accuracy(device, max_err):
    """The wide layers multiply fp32 operands on the bf16 matrix pipe from exactly split operands (k_spconv_x3).
    Measured here on a 64 -> 64 layer over a 60 k-row surface map: the default path and the fp32-MFMA kernel (forced through
    pp_spconv_fwd_ex) against a float64 evaluation of the same sums on the device; rows of very different magnitude.  Passes
    when the default path's error does not exceed twice the fp32-MFMA kernel's (or 5e-7 of the output's magnitude)."""
    from panopticsegforlargescalepointcloud_amd import ops
    g = torch.Generator().manual_seed(11)
    xy = torch.randint(-200, 200, (90000, 2), generator=g)
    z = ((xy[:, 0].float() * 0.05).sin() * 6 + (xy[:, 1].float() * 0.07).cos() * 5).round().int()
    c = torch.unique(torch.cat([torch.zeros(len(xy), 1, dtype=torch.int32), xy.int(), z[:, None]], 1), dim=0).to(device)
    n = c.shape[0]
    table, _ = ops.hash_build(c)
    nbr = ops.kernel_map(c, table, 3, 1, 1)
    x = (torch.randn(n, 64, generator=g) * torch.exp(torch.randn(n, 1, generator=g))).to(device)
    W = (torch.randn(27, 64, 64, generator=g) * 0.1).to(device)
    pk = ops.pack_weight(W)
    got = ops.spconv_fwd(x, pk, nbr, n, 64, 27)
    f32 = ops.spconv_fwd(x, pk, nbr, n, 64, 27, variant=(32, 1, 1))
    ref = torch.zeros(n, 64, dtype=torch.float64, device=device)
    xd, Wd = x.double(), W.double()
    for k in range(27):
        ok = nbr[k] >= 0
        ref[ok] += xd[nbr[k][ok].long()] @ Wd[k]
    mag = float(ref.abs().max())
    e_def, e_f32 = float((got.double() - ref).abs().max()) / mag, float((f32.double() - ref).abs().max()) / mag
    max_err["conv64_default_path_vs_float64"] = e_def
    max_err["conv64_fp32_mfma_vs_float64"] = e_f32
    split_on = not torch.equal(got, f32)
    max_err["conv64_default_path_is_split_operand_kernel"] = split_on
    return "pass" if e_def <= max(2.0 * e_f32, 5e-7) else "FAIL(default path %.2e vs fp32 MFMA %.2e of the magnitude)" % (e_def, e_f32)


def self_launch(n):
    """re-run this command line as n ranks under torch.distributed.run (same interpreter, a free port on 127.0.0.1)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--event-steps", type=int, default=0,
                    help="record the per-launch HIP events of the roofline object only in the first N timed steps (0 = all of them)")
    ap.add_argument("--points", type=int, default=env_int("PP_BENCH_POINTS", 10_000_000))
    ap.add_argument("--grid", type=int, default=env_int("PP_BENCH_GRID", 8))
    ap.add_argument("--tiles-per-batch", type=int, default=env_int("PP_BENCH_TPB", 64))
    ap.add_argument("--voxel", type=float, default=0.05)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-input-prefetch", action="store_true",
                    help="build every batch's coordinate manager at the start of its own pass (no overlap with the previous batch)")
    ap.add_argument("--backbone-ahead", choices=("auto", "on", "off"), default=os.environ.get("PP_BACKBONE_AHEAD", "auto"),
                    help="launch the next batch's backbone beside this batch's grouping (scene.TileRunner backbone_ahead).  auto = on "
                         "when a batch has fewer than PP_AHEAD_MAX_VOXELS (4 M) voxels -- there the step is paced by latency-bound "
                         "kernels and host reads (1.2 M voxels, one rank's share at 8 GPUs: 22.2 -> 17.6 ms) -- and off above: at "
                         "9.8 M voxels it is worth 5 %% of the step but convolutions that share the GPU with the grouping kernels "
                         "take longer per launch, so per-launch events stop measuring the kernel (roofline.frac 0.58 -> 0.47, "
                         "profiles/r06_backbone_ahead.txt)")
    ap.add_argument("--no-checks", action="store_true", help="skip the untimed self-check (batch invariance, oracle parity)")
    ap.add_argument("--layer-table", default=None, help="write the per-shape convolution table (markdown) here")
    ap.add_argument("--stage-timing", action="store_true", help="extra (untimed) step with per-stage wall times")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL rendezvous on
        # 127.0.0.1) and pass rank 0's JSON line through; under torch.distributed.run the environment is already set
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert args.gpus in (1, world), "--gpus %d under a launcher with WORLD_SIZE=%d" % (args.gpus, world)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    dev_index = local_rank % torch.cuda.device_count()  # (== local_rank on a real multi-GPU node)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; PP_DIST_BACKEND=gloo only exists to exercise the N > 1 control flow on a 1-GPU box
        dist.init_process_group(os.environ.get("PP_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    from panopticsegforlargescalepointcloud_amd import ops, synthetic as syn
    from panopticsegforlargescalepointcloud_amd.scene import TileRunner, exchange_tile_results, shard_tiles

    t_gen = time.perf_counter()
    scene, tiles, radius = build_scene(args.points, args.grid, args.voxel, 2022)
    sizes = [len(t) for t in tiles]
    total_points = int(sum(sizes))
    shards = shard_tiles(sizes, world)
    mine = shards[rank]
    model, cfg, DS = build_model(device, args.voxel)
    # a stream of batches: the next batch's backbone + heads are launched beside this batch's grouping / scorer front end
    # (scene.TileRunner; PP_BACKBONE_AHEAD=0 or --no-backbone-ahead: one batch at a time)
    # (decided below, once the batches exist: "auto" looks at their size)

    # resident inputs: tile batches + synthetic head statistics (HBM) before the timed region
    rng = np.random.default_rng(2022 + rank)
    batches = []
    for i in range(0, len(mine), args.tiles_per_batch):
        ids = mine[i: i + args.tiles_per_batch]
        b = syn.tile_batch(scene, tiles, ids)
        cls, off, emb = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, rng)
        dev_b = {k: torch.from_numpy(v).to(device) for k, v in b.items()}
        override = (torch.from_numpy(cls).to(device), torch.from_numpy(off).to(device), torch.from_numpy(emb).to(device))
        starts = np.concatenate([[0], np.cumsum([len(tiles[t]) for t in ids])])
        batches.append((ids, dev_b, override, starts, [int(len(tiles[t])) for t in ids]))
    t_gen = time.perf_counter() - t_gen
    mode = {"1": "on", "0": "off"}.get(args.backbone_ahead, args.backbone_ahead)
    if mode not in ("auto", "on", "off"):
        raise SystemExit("--backbone-ahead / PP_BACKBONE_AHEAD: auto, on or off")
    batch_voxels = max([int(bt[1]["coords"].shape[0]) for bt in batches] or [0])
    backbone_ahead = (mode == "on" or (mode == "auto" and 0 < batch_voxels < env_int("PP_AHEAD_MAX_VOXELS", 4_000_000))) \
        and not args.no_input_prefetch
    runner = TileRunner(model, device, backbone_ahead=backbone_ahead)

    stats = {"proposals": 0, "instances": 0, "local_ms": [], "exchange_events": []}
    input_prefetch = not args.no_input_prefetch

    def step(profile=False, input_prefetch=input_prefetch):
        local = {}
        stats["proposals"] = stats["instances"] = 0
        t_local = time.perf_counter()
        for j, (ids, dev_b, override, starts, sizes) in enumerate(batches):
            # the job is a stream of tile batches (this scene's next batch, or the first batch of the next scene): the runner
            # builds the coordinate manager of the batch that follows while this one is in its grouping / scorer stages
            nxt = batches[(j + 1) % len(batches)][1] if input_prefetch else None
            nxt2 = batches[(j + 2) % len(batches)][1] if input_prefetch else None
            labels, res, counts = runner.run(dev_b, len(ids), override=override, next_batch=nxt, after_next=nxt2)
            stats["proposals"] += res.clusters_csr.n if res.clusters_csr is not None else 0
            stats["instances"] += sum(counts)
            # what the scene assembly needs from a cylinder: origin ids, instance labels, semantic vote contributions
            # (views of the batch tensors, one split per tensor)
            for t, o, l, v in zip(ids, dev_b["origin_id"].split(sizes), labels.split(sizes), res.semantic_logits.split(sizes)):
                local[t] = (o, l, v)
        # (the per-tile instance counts were just read on the host: the local part of the step is complete here)
        stats["local_ms"].append(1e3 * (time.perf_counter() - t_local))
        if world == 1:
            return local
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        full = exchange_tile_results(local)
        e1.record()
        stats["exchange_events"].append((e0, e1))
        return full

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not warm-up: two priming passes so that every lazily created resource exists before the W warm-up steps --
    # the caching allocator's pools of both HIP streams, the grow-only workspaces, and the level / kernel-map prefetch
    # plan (recorded by a model's first inference pass, replayed on the side stream from its second pass on)
    t_prime = time.perf_counter()
    step()
    ops.PROFILER = ops.LaunchProfiler()  # (counts the convolution launches of a step; its timings are discarded)
    step()
    launches_per_step = len(ops.PROFILER.records)
    ops.PROFILER = None
    sync()
    t_prime = time.perf_counter() - t_prime
    for _ in range(max(args.warmup - 1, 0)):
        step()
    # per-launch HIP events for the roofline object: created here, only recorded inside the timed region
    timed_profiler = ops.LaunchProfiler(reserve=launches_per_step * args.steps + 16)
    # a serving process freezes the objects of its set-up and keeps the cyclic collector out of the request path: a
    # generation-2 pass over the model / scene object graph is a 20 - 40 ms stall at an arbitrary point of a step
    gc.collect()
    gc.freeze()
    gc.disable()
    # the LAST warm-up step runs after that pause (~100 ms with an idle GPU, after which the first step ran 2 ms slower than the
    # others): W warm-up steps in all, the timed region starts behind a working GPU
    if args.warmup >= 1:
        step()
    ops.PROFILER = timed_profiler
    sync()
    host0 = host_cpu_state()
    stats["local_ms"], stats["exchange_events"] = [], []
    t0 = time.perf_counter()
    step_ms = []
    profiler = ops.PROFILER
    event_steps = args.event_steps if 0 < args.event_steps < args.steps else args.steps
    for it in range(args.steps):
        ts = time.perf_counter()
        ops.PROFILER = profiler if it < event_steps else None
        result = step()          # (ends with the host read of the per-tile instance counts: the step's own sync point)
        step_ms.append(round(1e3 * (time.perf_counter() - ts), 2))
    # the last step prepared a batch nobody will run (and, with backbone_ahead, ran its backbone): that work is still part of the
    # timed region (K steps = K builds = K backbones; the first timed step's backbone ran in the last warm-up step)
    runner.drain()
    held = model.Backbone.__dict__.pop("_prepared_input", None)
    if held is not None:
        held[2].take().join_prefetch()
    sync()
    dt = time.perf_counter() - t0
    host1 = host_cpu_state()
    gc.enable()
    ops.PROFILER = profiler
    prof = ops.PROFILER.summarize()
    if args.layer_table and rank == 0:
        with open(args.layer_table, "w") as f:
            f.write("# pp_spconv_fwd launches of the timed steps grouped by shape (HIP events on the launch stream)\n\n")
            f.write(ops.PROFILER.table(event_steps))
    ops.PROFILER = None
    multi = None
    if world > 1:
        # what every rank did, so that a scaling record explains itself: this rank's wall time for the K steps, the local
        # part of a step (everything before the exchange), the exchange itself (HIP events around the all-gathers) and bytes
        ex_ms = [a.elapsed_time(b) for a, b in stats["exchange_events"]]
        n_mine = float(sum(len(tiles[t]) for t in mine))
        row = torch.tensor([1e3 * dt / args.steps, float(np.mean(stats["local_ms"])), float(np.mean(ex_ms)) if ex_ms else 0.0,
                            n_mine, float(len(mine))], dtype=torch.float64, device=device)
        rows = [torch.zeros_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        rows = torch.stack(rows).cpu().numpy()
        bpp = 8 + 4 * DS.num_classes  # origin id + instance label + C semantic log-probabilities (int32 / float32)
        multi = {"per_rank_step_ms": [round(v, 3) for v in rows[:, 0]], "per_rank_local_ms": [round(v, 3) for v in rows[:, 1]],
                 "per_rank_exchange_ms": [round(v, 3) for v in rows[:, 2]], "per_rank_points": [int(v) for v in rows[:, 3]],
                 "per_rank_tiles": [int(v) for v in rows[:, 4]], "exchange_bytes_per_point": bpp,
                 "exchange_bytes_sent_per_rank": [int(v) * bpp for v in rows[:, 3]],
                 "exchange_bytes_received_per_rank": int(rows[:, 3].max()) * bpp * world,  # padded all-gather
                 "exchange_collectives_per_step": 2, "backend": os.environ.get("PP_DIST_BACKEND", "nccl")}
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # the same step WITHOUT the next batch's coordinate manager being built during the current batch: what one cold scene
    # costs (reported beside the pipelined figure, never as `value`)
    single_scene_ms = None
    if input_prefetch:
        gc.disable()  # (as in the timed region: a generation-2 pass is a 20 - 40 ms stall in one of the three steps)
        step(input_prefetch=False)
        sync()
        t1 = time.perf_counter()
        for _ in range(3):
            step(input_prefetch=False)
        sync()
        single_scene_ms = round(1e3 * (time.perf_counter() - t1) / 3, 2)
        gc.enable()

    stage_ms = None
    if args.stage_timing:
        runner.stage_timing = True
        runner.stage_ms = {}
        step()
        runner.stage_timing = False
        stage_ms = {k: round(v, 1) for k, v in runner.stage_ms.items()}

    # HBM triad to confirm the roofline denominator on this box
    n_tri = 1 << 28
    a = torch.empty(n_tri, device=device)
    b2 = torch.ones(n_tri, device=device)
    c2 = torch.ones(n_tri, device=device)
    ops.triad(a, b2, c2, 2.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.triad(a, b2, c2, 2.0)
    e1.record()
    torch.cuda.synchronize()
    triad_gbs = 5 * 3 * 4 * n_tri / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b2, c2

    if rank == 0:
        secs = prof["ms"] * 1e-3
        t_hbm = prof["bytes"] / HBM_PEAK
        t_mfma = prof["flops"] / FP32_MFMA_PEAK
        if t_hbm >= t_mfma:
            roof = {"bound": "hbm", "achieved": prof["bytes"] / secs / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s"}
        else:
            roof = {"bound": "mfma", "achieved": prof["flops"] / secs / 1e12, "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        # per kernel family: fwd3 = fp32-MFMA kernel (16-channel layers), x3 = split-operand kernel (>= 32 channels).  `frac_mfma_fp32`
        # = algorithmic flops against the fp32 MFMA peak (the figure of earlier rounds, kept for continuity: the split kernel can
        # exceed 1 there); `frac_pipe` = against the pipe the family RUNS on (pipe_seconds: 6 bf16 products per fp32 product on the
        # 2.5 PFLOP/s pipe for x3) -- the honest denominator
        fams = {}
        pipe_s = 0.0
        for fam, pf in prof.get("by_family", {}).items():
            if pf["launches"]:
                n_ = pf["launches"]
                pipe_s += pipe_seconds(fam, pf["flops"])
                fams[fam] = {"launches_per_step": n_ // max(event_steps, 1), "ms_per_step": pf["ms"] / event_steps,
                             "alg_bytes_per_launch": pf["bytes"] / n_,
                             "frac_mfma_fp32": pf["flops"] / (pf["ms"] * 1e-3) / FP32_MFMA_PEAK,
                             "frac_pipe": pipe_seconds(fam, pf["flops"]) / (pf["ms"] * 1e-3),
                             "pipe": "bf16 MFMA 2.5 PFLOP/s, 6 products per fp32 product" if fam in ("x3", "x3f") else "fp32 MFMA 157.3 TFLOP/s",
                             "frac_hbm": pf["bytes"] / (pf["ms"] * 1e-3) / HBM_PEAK}
        # "x3_all": every launch on the split-operand arithmetic (k_spconv_x3 + k_spconv_x3f) -- the "x3" family of earlier rounds
        xs = [pf for fam, pf in prof.get("by_family", {}).items() if fam in ("x3", "x3f") and pf["launches"]]
        if xs:
            n_, ms_, fl_, by_ = (sum(pf[k] for pf in xs) for k in ("launches", "ms", "flops", "bytes"))
            fams["x3_all"] = {"launches_per_step": n_ // max(event_steps, 1), "ms_per_step": ms_ / event_steps,
                              "alg_bytes_per_launch": by_ / n_, "frac_mfma_fp32": fl_ / (ms_ * 1e-3) / FP32_MFMA_PEAK,
                              "frac_pipe": pipe_seconds("x3_all", fl_) / (ms_ * 1e-3), "frac_hbm": by_ / (ms_ * 1e-3) / HBM_PEAK,
                              "pipe": "bf16 MFMA 2.5 PFLOP/s, 6 products per fp32 product"}
        roof["by_kernel_family"] = fams
        roof["frac_pipe"] = pipe_s / secs  # all launches: matrix-pipe time at the dense peak of the pipe each runs on / measured time
        # HBM bytes per launch from the committed PMC passes of this same command (profiles/collect.sh -> traffic.json; a
        # PMC pass cannot run inside the timed bench).  FETCH_SIZE tallies 64 bytes per request and a request is <= 128 bytes
        # (profiles/r05_fetch_calibration.md: known / raw = 1.06 for random 64-byte rows, 2.0 for 128 / 256 / 384-byte rows and all
        # streaming loads): the convolutions gather every row in 64-byte pieces whatever its width (a lane quad reads one piece),
        # so their gathers are counted in full; their wide streaming loads -- the kernel map, known exactly -- are counted half
        # and added back.  Bounds: the gathers at 1.0 x and at 1.059 x the raw counter.
        roof["traffic"] = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # per kernel class (traffic.json "classes"): fwd3 gathers 64-byte rows, counted in full, and streams its kernel map
                # with wide loads, counted half (+ 0.5 x map bytes); x3f / x3 fetch rows, weights and map in 128-byte requests:
                # 2 x the raw counter (round 6)
                tot_tr = tot_lo = tot_hi = 0.0
                n_tot = 0
                for fam, pf in prof.get("by_family", {}).items():
                    cj = tj.get("classes", {}).get(fam)
                    if not (cj and pf["launches"]):
                        continue
                    n_ = pf["launches"]
                    w_c, f_c, m_c = cj["write_bytes_per_launch"], cj["fetch_raw_bytes_per_launch"], pf["map_bytes"] / n_
                    if fam in ("x3f", "x3"):
                        # both split-operand kernels reach the L2 with 128-byte requests (the register-gather kernel's two 64-byte
                        # pieces of a line merge before the L2: profiles/r06_pmc_fetch_x3f.md -- L2 requests x 128 bytes = rows +
                        # staged weights on both, equal memory-side request counts), a miss is a 128-byte fill tallied at 64 bytes.
                        # (lower bound: round 5's rule, every request tallied in full + half the map)
                        tr, lo, hi = w_c + 2.0 * f_c, w_c + f_c + 0.5 * m_c, w_c + 2.0 * f_c
                    else:
                        tr, lo, hi = w_c + f_c + 0.5 * m_c, w_c + f_c + 0.5 * m_c, w_c + 1.059 * f_c + 0.5 * m_c
                    if fam in fams:
                        fams[fam].update({"traffic_per_launch": tr, "traffic_over_algorithmic": tr / (pf["bytes"] / n_)})
                    tot_tr, tot_lo, tot_hi, n_tot = tot_tr + tr * n_, tot_lo + lo * n_, tot_hi + hi * n_, n_tot + n_
                if n_tot:
                    roof["traffic"] = tot_tr / n_tot
                    roof["traffic_bounds"] = [tot_lo / n_tot, tot_hi / n_tot]
                    roof["map_bytes_per_launch"] = prof["map_bytes"] / max(prof["launches"], 1)
                    roof["traffic_over_algorithmic"] = roof["traffic"] / (prof["bytes"] / max(prof["launches"], 1))
                roof["traffic_source"] = ("profiles/traffic.json (rocprofv3 --pmc passes of this command, per kernel class: WRITE_SIZE + "
                                          "FETCH_SIZE raw + 0.5 x kernel-map bytes for k_spconv_fwd3, WRITE_SIZE + 2 x FETCH_SIZE raw for "
                                          "k_spconv_x3f / k_spconv_x3 whose L2 requests are whole 128-byte lines; calibration "
                                          "profiles/r05_fetch_calibration.md, profiles/r06_pmc_fetch_x3f.md)")
            except Exception:
                pass
        roof["alg_bytes_per_launch"] = prof["bytes"] / max(prof["launches"], 1)
        x3_on = os.environ.get("PP_CONV_X3", "1") != "0"
        roof.update({"kernel": "k_spconv_x3f + k_spconv_x3 + k_spconv_fwd3 (pp_spconv_fwd)" if x3_on else "k_spconv_fwd3 (pp_spconv_fwd)",
                     # fp32 operands and fp32-accurate results everywhere; which matrix pipe multiplies them
                     "mfma_path": ("family x3 = layers with >= 32 input channels (and the 16 -> >= 48 ones): v_mfma_f32_16x16x32_bf16 on "
                                   "operands split exactly into three bfloat16 terms, six products, fp32 accumulation -- k_spconv_x3f "
                                   "(rows gathered as full 128-byte lines through LDS) where the input is whole 32-channel groups on a "
                                   "dense or 8-wide map, k_spconv_x3 (register gathers) on 48 / 80 / 112-channel inputs; family fwd3 = "
                                   "the 16-channel-input layers: v_mfma_f32_16x16x4_f32.  peak = the fp32 MFMA peak either way "
                                   "(frac); frac_pipe = against the pipe each family runs on") if x3_on else
                                  "v_mfma_f32_16x16x4_f32 on every layer (PP_CONV_X3=0)", "launches_per_step": prof["launches"] // max(event_steps, 1), "event_steps": event_steps,
                     "avg_launch_us": 1e3 * prof["ms"] / max(prof["launches"], 1),
                     "alg_GB_per_step": prof["bytes"] / event_steps / 1e9, "alg_TFLOP_per_step": prof["flops"] / event_steps / 1e12,
                     "hbm_GBps": prof["bytes"] / secs / 1e9, "mfma_TFLOPs": prof["flops"] / secs / 1e12,
                     "share_of_step_time": secs / (dt * event_steps / args.steps), "triad_GBps_measured": triad_gbs})
        out = {
            "metric": "points/sec end-to-end (sparse-conv fwd + clustering), 10M-pt scene",
            "value": total_points * args.steps / dt, "unit": "points/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic urban scene, %d overlapping cylinder tiles (r=%.1f m, 5 cm voxels), %d voxels fed "
                                   "(%d unique scene voxels); setting IV: U-Net fwd + heads + region_grow(shifted) + MeanShift(embed) "
                                   "+ ScorerUnet + NMS" % (len(tiles), radius, total_points, len(scene.pos)),
                       "tiles": len(tiles), "tiles_per_batch": args.tiles_per_batch, "points": total_points,
                       "input_prefetch": input_prefetch,  # next batch's coordinate manager built during the current batch
                       # next batch's backbone + heads launched beside this batch's grouping (own stream): --backbone-ahead auto = on
                       # below PP_AHEAD_MAX_VOXELS (4 M) voxels per batch
                       "backbone_ahead": backbone_ahead, "backbone_ahead_mode": mode, "batch_voxels": batch_voxels,
                       "single_scene_ms": single_scene_ms,  # a step that builds its own coordinate manager first (3 steps, untimed region)
                       "grouping_inputs": "synthetic head statistics (SURVEY.md 8d)", "parallelism": "tile-sharded x%d" % world,
                       "proposals_per_step": stats["proposals"], "instances_per_step": stats["instances"],
                       "setup_s": round(t_gen, 1), "priming_s": round(t_prime, 2), "stage_ms": stage_ms, "multi_gpu": multi,
                       # the host side of the timed region (the step has ~25 host reads of data-dependent sizes): CPU time
                       # this process used, and how long the container's CPU quota throttled it
                       "host": {"cpu_s": round(host1[0] - host0[0], 3), "wall_s": round(dt, 3), "step_ms": step_ms,
                                "cgroup_throttled_ms": round((host1[1] - host0[1]) / 1e3, 1),
                                "cgroup_throttled_periods": host1[2] - host0[2], "device_mallocs": host1[3] - host0[3], "torch_threads": torch.get_num_threads()}},
            "roofline": roof,
        }
        oracle_case = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], oracle_case = cpu_baseline(model, cfg, DS, scene, tiles, args.voxel)
        if not args.no_checks:
            out["config"]["checks"] = self_check(runner, batches, device, oracle_case)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
