"""GPU: scene assembly (block merging) and final evaluation on the device (csrc/pp_eval.hip) vs the NumPy restatements
(scene.block_merging / SceneAssembler, panoptic/metrics.panoptic_evaluation) and vs the numbers the reference's own
final_eval logs (tests/golden/final_eval_cases.npz).  Integer work: bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from panopticsegforlargescalepointcloud_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_histogram2d_and_pair_counts(ops):
    rng = np.random.default_rng(0)
    for n, na, nb in [(0, 3, 4), (1, 1, 1), (100_000, 10, 10), (300_000, 700, 10), (200_000, 3000, 40)]:
        a = rng.integers(-1, na, size=n)
        b = rng.integers(-1, nb, size=n)
        want = np.zeros((na, nb), np.int64)
        m = (a >= 0) & (b >= 0)
        np.add.at(want, (a[m], b[m]), 1)
        got = ops.histogram2d(dev(a), dev(b), na, nb)  # host table; skipped rows (a < 0 or b < 0) are allowed here
        assert np.array_equal(got, want)
        if (~m).any():
            with pytest.raises(Exception, match="negative label"):
                ops.histogram2d(dev(a), dev(b), na, nb, allow_skipped=False)
        if n:
            pa, pb, cnt = (t.cpu().numpy() for t in ops.pair_counts(dev(a), dev(b), nb, capacity=64))  # forces the retry path
            wa, wb = np.nonzero(want)
            assert np.array_equal(pa, wa) and np.array_equal(pb, wb) and np.array_equal(cnt, want[wa, wb])
    # sorted labels (the realistic case: long runs of equal pairs -> one atomic per run)
    a = np.repeat(np.arange(50), 4000)
    b = np.repeat(np.arange(100), 2000)
    pa, pb, cnt = (t.cpu().numpy() for t in ops.pair_counts(dev(a), dev(b), 100))
    assert len(pa) == 100 and np.all(cnt == 2000) and np.array_equal(pa, np.arange(100) // 2)
    with pytest.raises(Exception, match="outside"):  # out-of-range labels raise instead of being dropped
        ops.histogram2d(dev(np.array([0, 5])), dev(np.array([0, 0])), 3, 3)
    from panopticsegforlargescalepointcloud_amd.panoptic.metrics import panoptic_evaluation_device
    sem = dev(np.array([0, 1, 9, 2]))  # class 9 with num_classes = 9: the NumPy form's bincount fails on it as well
    with pytest.raises(Exception, match="outside"):
        panoptic_evaluation_device(sem, dev(np.array([-1, -1, 0, 0])), dev(np.array([0, 1, 2, 2])), dev(np.array([-1, -1, 0, 0])))


def test_panoptic_evaluation_device_matches_numpy_and_reference_log(ops):
    from panopticsegforlargescalepointcloud_amd.panoptic.metrics import panoptic_evaluation, panoptic_evaluation_device
    z = np.load(os.path.join(GOLD, "final_eval_cases.npz"))
    cases = [(z["pred_sem_" + n], z["pred_ins_" + n], z["gt_sem_" + n], z["gt_ins_" + n]) for n in z["names"].tolist()]
    rng = np.random.default_rng(5)
    n = 400_000  # a scene-sized case: 300 predicted / 250 ground-truth instances with noisy overlap
    gt_ins = rng.integers(-1, 250, size=n)
    gt_sem = np.where(gt_ins >= 0, np.array([2, 3, 4, 6, 7, 8])[gt_ins % 6], rng.choice([-1, 0, 1, 5], size=n))
    pred_ins = np.where(rng.random(n) < 0.85, gt_ins + (gt_ins >= 0) * 7, rng.integers(-1, 300, size=n))
    pred_sem = np.where(rng.random(n) < 0.9, gt_sem, rng.integers(0, 9, size=n))
    pred_sem = np.where(pred_sem < 0, 0, pred_sem)
    cases.append((pred_sem, pred_ins, gt_sem, gt_ins))
    for ps, pi, gs, gi in cases:
        want = panoptic_evaluation(ps, pi, gs, gi)
        got = panoptic_evaluation_device(dev(np.asarray(ps, np.int64)), dev(np.asarray(pi, np.int64)),
                                         dev(np.asarray(gs, np.int64)), dev(np.asarray(gi, np.int64)))
        assert sorted(got) == sorted(want)
        for k in want:
            np.testing.assert_array_equal(np.asarray(got[k]), np.asarray(want[k]), err_msg=k)  # same tables, same float ops
    # and the device path against the reference's own log lines directly
    for name in z["names"].tolist():
        r = panoptic_evaluation_device(*(dev(z[k + name].astype(np.int64)) for k in ("pred_sem_", "pred_ins_", "gt_sem_", "gt_ins_")))
        for mine, theirs in {"mIoU": "Semantic_Segmentation_mIoU", "F1": "Instance_Segmentation_F1_score",
                             "meanPQ": "Instance_Segmentation_meanPQ", "mMWCov": "Instance_Segmentation_mMWCov"}.items():
            np.testing.assert_allclose(r[mine], z["log_%s_%s" % (name, theirs)], rtol=1e-6, atol=1e-7)


def _blocks(rng, n_scene, n_blocks, per_block, inst_per_block):
    """overlapping blocks (cylinders): contiguous id windows with random instance partitions"""
    out = []
    for b in range(n_blocks):
        start = int(b * (n_scene - per_block) / max(n_blocks - 1, 1))
        origin = np.sort(rng.choice(np.arange(start, start + per_block), size=int(per_block * 0.8), replace=False)).astype(np.int64)
        labels = np.full(len(origin), -1, np.int32)
        k = int(rng.integers(0, inst_per_block + 1))
        if k:
            cuts = np.sort(rng.choice(len(origin), size=2 * k, replace=False))
            ids = rng.permutation(k + 2)[:k]  # non-contiguous instance ids: unused ids still burn a label in the reference
            for i in range(k):
                labels[cuts[2 * i]: cuts[2 * i + 1]] = ids[i]
        out.append((origin, labels))
    return out


def test_block_merging_device_matches_numpy(ops):
    from panopticsegforlargescalepointcloud_amd.scene import SceneAssembler, SceneAssemblerGPU
    rng = np.random.default_rng(8)
    for n_scene, n_blocks, per_block, k in [(3000, 6, 1000, 5), (60_000, 25, 6000, 30), (400, 3, 200, 0), (50_000, 12, 9000, 60)]:
        blocks = _blocks(rng, n_scene, n_blocks, per_block, k)
        cpu = SceneAssembler(n_scene, 9)
        gpu = SceneAssemblerGPU(n_scene, 9, "cuda")
        for origin, labels in blocks:
            logits = rng.normal(size=(len(origin), 9)).astype(np.float32)
            cpu.add_block(origin, labels, logits)
            gpu.add_block(dev(origin), dev(labels), dev(logits))
        gpu.finish()
        assert np.array_equal(gpu.ins_pre.cpu().numpy(), cpu.ins_pre)
        assert gpu.max_instance == cpu.max_instance
        assert np.array_equal(gpu.prediction_count.cpu().numpy(), cpu.prediction_count)
        np.testing.assert_allclose(gpu.votes.cpu().numpy(), cpu.votes, rtol=1e-6, atol=1e-6)


def test_block_merging_device_matches_the_reference(ops):
    """the device assembler against the output of the reference's own block_merging, block after block
    (tests/golden/block_merging_cases.npz, make_golden.py)"""
    from panopticsegforlargescalepointcloud_amd.scene import SceneAssemblerGPU
    z = np.load(os.path.join(GOLD, "block_merging_cases.npz"))
    for name in z["names"].tolist():
        n_scene = int(z["n_scene_" + name])
        offs, origin, labels = z["block_offsets_" + name], z["origin_" + name], z["labels_" + name]
        gpu = SceneAssemblerGPU(n_scene, 2, "cuda")
        for b in range(len(offs) - 1):
            gpu.add_block(dev(origin[offs[b]: offs[b + 1]]), dev(labels[offs[b]: offs[b + 1]].astype(np.int32)))
            assert np.array_equal(gpu.ins_pre.cpu().numpy(), z["after_" + name][b]), (name, b)
            assert gpu.max_instance == int(z["max_instance_" + name][b])
        gpu.finish()


def test_block_merging_rules_device(ops):
    """the hand-written cases of tests/test_host_logic.py::test_block_merging_rules on the device, step by step"""
    from panopticsegforlargescalepointcloud_amd.scene import SceneAssembler, SceneAssemblerGPU
    cpu, gpu = SceneAssembler(20, 3), SceneAssemblerGPU(20, 3, "cuda")
    steps = [(np.arange(10), [0, 0, 0, 0, 0, 1, 1, 1, -1, -1]),          # untouched region: labels + max_instance
             (np.arange(3, 13), [0, 0, 0, 0, 0, 0, 0, 1, 1, 1]),         # merge by best IoU / new instance (max_instance + 1)
             (np.arange(0, 8), [0, 0, 1, 1, 2, 2, 3, 3]),                # fully labelled block: nothing changes
             (np.arange(12, 20), [-1] * 8),                              # no instance in the block
             (np.arange(8, 20), [5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5])]   # sparse instance ids: unused ids burn labels
    for origin, labels in steps:
        cpu.add_block(origin, np.asarray(labels))
        gpu.add_block(dev(origin), dev(np.asarray(labels, np.int32)))
        assert gpu.ins_pre.cpu().tolist() == cpu.ins_pre.tolist() and gpu.max_instance == cpu.max_instance
    gpu.finish()
    assert cpu.ins_pre[10:13].tolist() == [3, 3, 3]  # the reference allocates max_instance + 1


def test_treeins_evaluation_device_matches_reference_log(ops):
    """FOR-instance final evaluation (two instance predictions, datasets/panoptic/treeins.py:99-497) on device tensors ==
    the reference's own log (tests/golden/treeins_eval_cases.npz) == the NumPy form."""
    import test_host_logic as thl
    from panopticsegforlargescalepointcloud_amd.panoptic.metrics import panoptic_evaluation_treeins
    z = np.load(os.path.join(GOLD, "treeins_eval_cases.npz"))
    for name in z["names"].tolist():
        args = [z[k + name].astype(np.int64) for k in ("pred_sem_", "pre_emb_", "pre_off_", "gt_sem_", "gt_ins_")]
        r = panoptic_evaluation_treeins(*(dev(a) for a in args))
        thl.check_treeins_result(z, name, r)
        h = panoptic_evaluation_treeins(*args)
        for sec in ("offset", "embed"):
            for k in h[sec]:
                np.testing.assert_array_equal(np.asarray(r[sec][k]), np.asarray(h[sec][k]), err_msg="%s %s %s" % (name, sec, k))


def test_tracker_batch_metrics_match_reference(ops):
    """compute_acc / compute_eval == the reference tracker's own _compute_acc / _compute_eval
    (metrics/panoptic_tracker_pointgroup_npm3d.py:678-879) on the batches of tests/golden/tracker_metric_cases.npz:
    (tp, fp, acc) and (cov, wcov, mPrecision, mRecall, F1), through pp_instance_iou, pp_histogram2d and pp_pair_counts."""
    import types
    from panopticsegforlargescalepointcloud_amd.panoptic.metrics import compute_acc, compute_eval
    z = np.load(os.path.join(GOLD, "tracker_metric_cases.npz"))
    for name in z["names"].tolist():
        off = z["cl_offsets_" + name]
        pts = dev(z["cl_points_" + name].astype(np.int64))
        clusters = [pts[off[i]:off[i + 1]] for i in range(len(off) - 1)]
        labels = types.SimpleNamespace(instance_labels=dev(z["inst_" + name].astype(np.int64)), y=dev(z["y_" + name].astype(np.int64)),
                                       num_instances=dev(z["num_instances_" + name].astype(np.int64)))
        batch, pred = dev(z["batch_" + name].astype(np.int64)), dev(z["pred_" + name].astype(np.int64))
        acc = compute_acc(clusters, pred, labels, batch, labels.num_instances, 0.5)
        np.testing.assert_allclose(np.asarray([float(v) for v in acc]), z["acc_" + name], rtol=1e-6, atol=1e-7, err_msg=name)
        ev = compute_eval(clusters, pred, labels, batch, labels.num_instances, 9, 0.5)
        np.testing.assert_allclose(np.asarray([float(v) for v in ev]), z["eval_" + name], rtol=1e-5, atol=1e-6, err_msg=name)
