"""CPU tests of the host-side logic: config loading (the reference YAML schema), model assembly / state_dict layout,
tile sharding, NMS + painting, block merging, synthetic scenes.  No GPU, no HIP calls."""
import os

import numpy as np
import pytest
import torch

import bruteforce as bf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_YAML = "/root/reference/conf/models/panoptic/area4_ablation_3heads_5.yaml"


class DS:
    feature_dimension = 4
    num_classes = 9
    stuff_classes = torch.tensor([0, 1, 5])


def _build(path):
    from panopticsegforlargescalepointcloud_amd.config import load_model_config
    from panopticsegforlargescalepointcloud_amd.panoptic import PointGroup3heads
    cfg = load_model_config(path, "PointGroup-PAPER", data={"grid_size": 0.12})
    torch.manual_seed(0)
    return PointGroup3heads(cfg, "dummy", DS, None), cfg


@pytest.mark.parametrize("path", [os.path.join(ROOT, "conf", "panoptic_3heads.yaml"), REF_YAML])
def test_model_from_reference_yaml_schema(path):
    if not os.path.exists(path):
        pytest.skip("reference tree not present on this box")
    m, cfg = _build(path)
    assert abs(cfg.cluster_radius_search - 1.5 * 0.12) < 1e-9 and cfg.cluster_type == 5 and cfg.prepare_epoch == 30
    sd = m.state_dict()
    conv = sum(v.numel() for k, v in sd.items() if k.startswith("Backbone.") and k.endswith(".kernel"))
    bn = sum(v.numel() for k, v in sd.items() if k.startswith("Backbone.") and k.endswith(("bn.weight", "bn.bias")))
    scorer = sum(v.numel() for k, v in sd.items() if k.startswith("ScorerUnet.") and k.endswith(".kernel"))
    assert (conv, bn, scorer) == (10403520, 10176, 931840)
    # ME layout and names: kernel [27, Cin, Cout]; 1x1 shortcut [Cin, Cout]; BN under .bn
    assert tuple(sd["Backbone.down_modules.0.conv_in.0.kernel"].shape) == (27, 4, 16)
    assert tuple(sd["Backbone.down_modules.1.blocks.0.downsample.0.kernel"].shape) == (16, 32)
    assert tuple(sd["Backbone.up_modules.1.conv_in.0.kernel"].shape) == (27, 192, 192)
    assert "Backbone.up_modules.6.blocks.1.block.4.bn.running_var" in sd
    assert tuple(sd["Semantic.0.0.0.weight"].shape) == (16, 16) and "Semantic.0.0.1.batch_norm.weight" in sd
    assert tuple(sd["Offset.1.weight"].shape) == (3, 16) and tuple(sd["Embed.1.weight"].shape) == (5, 16)
    assert tuple(sd["ScorerHead.0.weight"].shape) == (1, 16)
    assert m._stuff_classes.tolist() == [-1, 0, 1, 5]
    # kaiming fan_out init of conv kernels (applications/minkowski.py:104-111): std = sqrt(2 / (27 * Cout))
    k = sd["Backbone.down_modules.3.blocks.1.block.0.kernel"]
    assert abs(float(k.std()) - (2.0 / (27 * 64)) ** 0.5) < 2e-3


REF_DIR = "/root/reference/conf/models/panoptic"
# (file, class name, heads, cluster_type, scorer_type, use_score_net) -- SURVEY.md App. B, README.md:185
SETTINGS = [("area4_ablation_19.yaml", "PointGroupEmbed", ("Semantic", "Embed"), 7, None, False),
            ("area4_ablation_14.yaml", "PointGroup", ("Semantic", "Offset"), 1, None, False),
            ("area4_ablation_15.yaml", "PointGroup", ("Semantic", "Offset"), 2, "unet", True),
            ("area4_ablation_3heads_5.yaml", "PointGroup3heads", ("Semantic", "Offset", "Embed"), 5, "unet", True),
            ("area4_ablation_3heads_6.yaml", "PointGroup3heads", ("Semantic", "Offset", "Embed"), 6, "unet", True)]


def _check_setting(cfg, cls_name, heads, cluster_type, scorer_type, use_score_net):
    from panopticsegforlargescalepointcloud_amd.panoptic import instantiate_model
    torch.manual_seed(0)
    m = instantiate_model(cfg, DS)
    assert type(m).__name__ == cls_name and m.HEADS == heads
    assert cfg.cluster_type == cluster_type and cfg.use_score_net is use_score_net
    # "scorer_type: None" must resolve to Python None (model_definition_resolver.py:44-48), else _compute_score would
    # look for a scorer called "None" instead of using the semantic-certainty score
    assert cfg.scorer_type == scorer_type and m._scorer_type == scorer_type
    assert cluster_type in m._cluster_fns()
    sd = m.state_dict()
    conv = sum(v.numel() for k, v in sd.items() if k.startswith("Backbone.") and k.endswith(".kernel"))
    assert conv == 10403520
    for h in ("Offset", "Embed"):
        assert any(k.startswith(h + ".") for k in sd) == (h in heads)
    assert abs(cfg.cluster_radius_search - 1.5 * 0.05) < 1e-9 and cfg.prepare_epoch == 30


@pytest.mark.parametrize("fname,cls_name,heads,cluster_type,scorer_type,use_score_net", SETTINGS)
def test_all_published_reference_yamls_load(fname, cls_name, heads, cluster_type, scorer_type, use_score_net):
    from panopticsegforlargescalepointcloud_amd.config import load_model_config
    path = os.path.join(REF_DIR, fname)
    if not os.path.exists(path):
        pytest.skip("reference tree not present on this box")
    cfg = load_model_config(path, "PointGroup-PAPER", data={"grid_size": 0.05})
    _check_setting(cfg, cls_name, heads, cluster_type, scorer_type, use_score_net)


@pytest.mark.parametrize("name,spec", list(zip(["setting-I", "setting-II", "setting-III", "setting-IV", "setting-V"], SETTINGS)))
def test_build_owned_settings_yaml(name, spec):
    from panopticsegforlargescalepointcloud_amd.config import load_model_config
    cfg = load_model_config(os.path.join(ROOT, "conf", "panoptic_settings.yaml"), name, data={"grid_size": 0.05})
    _check_setting(cfg, *spec[1:])


def _structure_matches_fixture(device):
    """the build-owned networks vs tests/golden/structure_fixture.json, which tests/test_reference_binding.py derives from
    the REFERENCE's api_modules.py / applications/minkowski.py executing on the shim (build container only)"""
    import json
    from panopticsegforlargescalepointcloud_amd.applications import Minkowski
    from panopticsegforlargescalepointcloud_amd.config import load_model_config
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "structure_fixture.json")))["networks"]
    for name in ["setting-I", "setting-II", "setting-III", "setting-IV", "setting-V"]:
        cfg = load_model_config(os.path.join(ROOT, "conf", "panoptic_settings.yaml"), name, data={"grid_size": 0.05})
        for part, input_nc, conf in [("Backbone", 4, cfg.backbone.config), ("ScorerUnet", 16, cfg.scorer_unet)]:
            net = Minkowski("unet", input_nc=input_nc, num_layers=4, config=conf).to(device)
            assert {k: list(v.shape) for k, v in net.state_dict().items()} == fx[part]
            assert list(net.state_dict()) == sorted(fx[part], key=list(net.state_dict()).index)


def test_structure_fixture_cpu():
    _structure_matches_fixture("cpu")


@pytest.mark.gpu
def test_structure_fixture_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    _structure_matches_fixture("cuda")


def test_config_resolver_keeps_non_expressions():
    from panopticsegforlargescalepointcloud_amd.config import resolve
    cfg = {"a": "2*in_feat", "b": "max", "c": "ResNetDown", "d": ["FEAT", "in_feat"], "e": "1.5 * 0.05"}
    resolve(cfg, {"in_feat": 16, "FEAT": 4})
    assert cfg == {"a": 32, "b": "max", "c": "ResNetDown", "d": [4, 16], "e": 0.07500000000000001}


def test_shard_tiles_balanced_and_complete():
    from panopticsegforlargescalepointcloud_amd.scene import shard_tiles
    sizes = np.random.default_rng(0).integers(100_000, 200_000, size=64)
    for w in (1, 2, 4, 8):
        sh = shard_tiles(sizes, w)
        assert sorted(sum(sh, [])) == list(range(64))
        loads = [sizes[s].sum() for s in sh]
        assert max(loads) / min(loads) < 1.08
        assert all(s == sorted(s) for s in sh)


def test_synthetic_scene_and_tiles():
    import bench
    from panopticsegforlargescalepointcloud_amd import synthetic as syn
    scene, tiles, r = bench.build_scene(120_000, 2, 0.05, 2022)
    assert abs(sum(len(t) for t in tiles) - 120_000) / 120_000 < 0.05
    assert len(np.unique(scene.coords, axis=0)) == len(scene.coords)  # one point per voxel
    covered = np.zeros(len(scene.pos), bool)
    for t in tiles:
        covered[t] = True
    assert covered.mean() > 0.97  # cylinders overlap and cover the scene (corners excepted)
    b = syn.tile_batch(scene, tiles, [0, 3])
    cc = np.concatenate([b["batch"][:, None], b["coords"]], 1)
    assert len(np.unique(cc, axis=0)) == len(cc) and b["x"].shape[1] == 4
    cls, off, emb = syn.synthetic_head_outputs(scene, b["origin_id"], 0.0, np.random.default_rng(0))
    thing = scene.inst[b["origin_id"]] > 0
    assert np.all(np.isin(cls[thing], syn.THING_CLASSES)) and np.all(off[~thing] == 0)
    # shifted points pile up around the instance centre
    sh = scene.pos[b["origin_id"]][thing] + off[thing] - scene.inst_center[scene.inst[b["origin_id"]][thing]]
    assert np.abs(sh).mean() < 0.06


def test_host_nms_matches_reference_golden():
    """the host restatement of non_max_suppression + the size / score filters vs the reference's own implementation
    (tests/golden/nms_cases.npz); the device path is checked against the same vectors in tests/test_nms_gpu.py."""
    from panopticsegforlargescalepointcloud_amd.panoptic.structures import non_max_suppression
    z = np.load(os.path.join(ROOT, "tests", "golden", "nms_cases.npz"))
    offs = z["cluster_offsets"]
    clusters = [z["cluster_points"][offs[i]: offs[i + 1]] for i in range(len(offs) - 1)]
    n = int(max(c.max() for c in clusters)) + 1
    dense = np.zeros((len(clusters), n), np.float32)
    for i, c in enumerate(clusters):
        dense[i, c] = 1
    m = dense @ dense.T
    sz = np.diag(m).copy()
    ious = m / (sz[:, None] + sz[None, :] - m)
    for tag, (thr, mn, ms) in {"default": (0.3, 100, 0.5), "tracker": (0.3, 10, 0.5), "loose": (0.6, 0, 0.0)}.items():
        pick = non_max_suppression(ious, z["scores"], thr)
        ids = [i for i in pick if sz[i] > mn and z["scores"][i] > ms]
        assert ids == z["ids_" + tag].tolist()
        assert [int(sz[i]) for i in ids] == z["sizes_" + tag].tolist()


def test_oracle_painting_matches_the_reference_tracker(oracle):
    """oracle/pipeline.instance_labels (the checker of the device path) vs the labels the reference's own tracker paints
    (tests/golden/nms_cases.npz: paint_tracker, generated by executing get_instances + get_cur_ins_pre_label)"""
    from oracle import pipeline as opipe
    z = np.load(os.path.join(ROOT, "tests", "golden", "nms_cases.npz"))
    offs = z["cluster_offsets"]
    n = int(z["n"])
    clusters = [z["cluster_points"][offs[i]: offs[i + 1]] for i in range(len(offs) - 1)]
    got = opipe.instance_labels({"clusters": clusters, "cluster_scores": z["scores"]}, n, np.zeros(n, np.int64))
    assert np.array_equal(got.astype(np.int64), z["paint_tracker"])


@pytest.mark.parametrize("use_sklearn", [True, False])
def test_oracle_proposal_generation_matches_the_reference_model(oracle, use_sklearn):
    """oracle/pipeline.group (the checker of the model's _cluster* functions) vs the proposals the reference's OWN
    PointGroup3heads._cluster / _cluster2 / _cluster3 / _cluster4 / _cluster5 / _cluster6 produce (tests/golden/proposal_cases.npz, generated by
    executing those methods with the reference's mean-shift module; make_golden.py): same proposals in the same order, same
    cluster_type codes, with sklearn's MeanShift and with the oracle's own mean shift behind it."""
    from oracle import pipeline as opipe
    z = np.load(os.path.join(ROOT, "tests", "golden", "proposal_cases.npz"))
    for name in z["names"].tolist():
        pos, off, emb, pred, batch = (z["%s_%s" % (k, name)] for k in ("pos", "off", "emb", "pred", "batch"))
        for fn, ct in (("_cluster", 1), ("_cluster2", 2), ("_cluster3", 3), ("_cluster4", 4), ("_cluster5", 5), ("_cluster6", 6)):
            opt = {"cluster_type": ct, "cluster_radius_search": float(z["radius_" + name]), "bandwidth": float(z["bandwidth_" + name])}
            clusters, types = opipe.group(pos, batch, pred, off, emb, opt, [0, 1, 5], use_sklearn_meanshift=use_sklearn)
            tag = name + fn
            offs, pts = z["offsets_" + tag], z["points_" + tag]
            assert len(clusters) == len(offs) - 1, tag
            for i, c in enumerate(clusters):
                assert np.array_equal(np.sort(c), np.sort(pts[offs[i]: offs[i + 1]])), (tag, i)
            assert np.array_equal(np.asarray(types, np.uint8), z["types_" + tag]), tag


def test_block_merging_rules():
    from panopticsegforlargescalepointcloud_amd.scene import SceneAssembler
    asm = SceneAssembler(20, 3)
    # block 0: instances {0: pts 0-4, 1: pts 5-7}
    asm.add_block(np.arange(10), np.array([0, 0, 0, 0, 0, 1, 1, 1, -1, -1]))
    assert asm.max_instance == 2 and asm.ins_pre[:8].tolist() == [0] * 5 + [1] * 3
    # block 1 overlaps pts 3-12: its instance 0 = pts 3-9 (IoU with old 0: 2/10 > 0.1 -> merge), instance 1 = pts 10-12 (new)
    asm.add_block(np.arange(3, 13), np.array([0, 0, 0, 0, 0, 0, 0, 1, 1, 1]))
    assert asm.ins_pre[5:8].tolist() == [1, 1, 1]  # already labelled points keep their label
    assert asm.ins_pre[8:10].tolist() in ([0, 0], [1, 1])  # merged into the best-IoU old instance
    assert asm.ins_pre[10:13].tolist() == [3, 3, 3] and asm.max_instance == 3  # reference allocates max_instance + 1
    assert asm.prediction_count[3:10].tolist() == [2] * 7
    # a fully labelled block changes nothing
    before = asm.ins_pre.copy()
    asm.add_block(np.arange(0, 8), np.array([0, 0, 1, 1, 2, 2, 3, 3]))
    assert np.array_equal(before, asm.ins_pre)


def test_block_merging_matches_the_reference_block_after_block():
    """scene.block_merging / SceneAssembler against the output of the reference's OWN block_merging
    (tests/golden/block_merging_cases.npz, generated by running the tracker's method, make_golden.py): scene labels and
    max_instance after every block, incl. its label-allocation quirks (unused block-local ids burn labels, max_instance + 1)."""
    from panopticsegforlargescalepointcloud_amd.scene import SceneAssembler, block_merging
    z = np.load(os.path.join(ROOT, "tests", "golden", "block_merging_cases.npz"))
    for name in z["names"].tolist():
        n_scene = int(z["n_scene_" + name])
        offs, origin, labels = z["block_offsets_" + name], z["origin_" + name], z["labels_" + name]
        asm = SceneAssembler(n_scene, 2)
        cur, mx = np.full(n_scene, -1, np.int64), 0
        for b in range(len(offs) - 1):
            o, lab = origin[offs[b]: offs[b + 1]], labels[offs[b]: offs[b + 1]]
            cur, mx = block_merging(o, lab, cur, mx)
            asm.add_block(o, lab)
            assert np.array_equal(cur, z["after_" + name][b]) and mx == int(z["max_instance_" + name][b]), (name, b)
            assert np.array_equal(asm.ins_pre, z["after_" + name][b]) and asm.max_instance == mx


def test_panoptic_quality_metric():
    from panopticsegforlargescalepointcloud_amd.panoptic.metrics import thing_panoptic_quality
    gt_sem = np.array([2] * 10 + [2] * 10 + [3] * 10 + [0] * 10)
    gt_ins = np.array([1] * 10 + [2] * 10 + [3] * 10 + [0] * 10)
    # perfect prediction
    r = thing_panoptic_quality(gt_sem, gt_ins - 1, gt_sem, gt_ins, [2, 3])
    assert abs(r["PQ"] - 1.0) < 1e-12
    # instance 2 split 6/4 (IoU 0.6 -> TP, IoU 0.4 -> FP), instance 3 missed
    pred = np.array([0] * 10 + [1] * 6 + [2] * 4 + [-1] * 10 + [-1] * 10)
    r = thing_panoptic_quality(gt_sem, pred, gt_sem, gt_ins, [2, 3])
    c2 = r["per_class"][2]
    assert c2["n_pred"] == 3 and c2["n_gt"] == 2
    assert abs(c2["precision"] - 2 / 3) < 1e-12 and abs(c2["recall"] - 1.0) < 1e-12 and abs(c2["SQ"] - 0.8) < 1e-12
    assert r["per_class"][3]["PQ"] == 0.0 and abs(r["PQ"] - (0.8 * 0.8 + 0.0) / 2) < 1e-12


def test_panoptic_evaluation_matches_reference_final_eval():
    """panoptic_evaluation vs the numbers logged by the reference's own final_eval (tests/golden/make_golden.py)."""
    from panopticsegforlargescalepointcloud_amd.panoptic.metrics import panoptic_evaluation
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "final_eval_cases.npz"))
    pairs = {"oAcc": "Semantic_Segmentation_oAcc", "mAcc": "Semantic_Segmentation_mAcc", "mIoU": "Semantic_Segmentation_mIoU",
             "MUCov": "Instance_Segmentation_MUCov", "mMUCov": "Instance_Segmentation_mMUCov",
             "MWCov": "Instance_Segmentation_MWCov", "mMWCov": "Instance_Segmentation_mMWCov",
             "Precision": "Instance_Segmentation_Precision", "mPrecision": "Instance_Segmentation_mPrecision",
             "Recall": "Instance_Segmentation_Recall", "mRecall": "Instance_Segmentation_mRecall",
             "F1": "Instance_Segmentation_F1_score", "RQ": "Instance_Segmentation_RQ", "SQ": "Instance_Segmentation_SQ",
             "PQ": "Instance_Segmentation_PQ", "meanRQ": "Instance_Segmentation_meanRQ", "meanSQ": "Instance_Segmentation_meanSQ",
             "meanPQ": "Instance_Segmentation_meanPQ", "PQ_things": "Instance_Segmentation_PQ_things_",
             "meanRQ_things": "Instance_Segmentation_meanRQ_things_", "meanSQ_things": "Instance_Segmentation_meanSQ_things_",
             "meanPQ_things": "Instance_Segmentation_meanPQ_things_", "PQ_stuff": "Instance_Segmentation_PQ_stuff_",
             "meanPQ_stuff": "Instance_Segmentation_meanPQ_stuff_"}
    for name in z["names"].tolist():
        r = panoptic_evaluation(z["pred_sem_" + name], z["pred_ins_" + name], z["gt_sem_" + name], z["gt_ins_" + name])
        for mine, theirs in pairs.items():
            want = z["log_%s_%s" % (name, theirs)]
            got = np.atleast_1d(np.asarray(r[mine], np.float64))
            if mine in ("RQ", "SQ", "PQ"):   # numpy wraps the 9-element arrays in the log: the first line was parsed
                got = got[: len(want)]
            assert got.shape == want.shape, (name, mine, got, want)
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7, err_msg="%s %s" % (name, mine))


TREEINS_KEYS = {"MUCov": "MUCov", "mMUCov": "mMUCov", "MWCov": "MWCov", "mMWCov": "mMWCov", "Precision": "Precision",
                "mPrecision": "mPrecision", "Recall": "Recall", "mRecall": "mRecall", "F1": "F1_score", "RQ": "RQ", "meanRQ": "meanRQ",
                "SQ": "SQ", "meanSQ": "meanSQ", "PQ": "PQ", "meanPQ": "meanPQ", "PQStar": "PQ_star", "meanPQStar": "mean_PQ_star",
                "PQ_things": "PQ_things_", "meanPQ_things": "meanPQ_things_", "meanRQ_things": "meanRQ_things_",
                "meanSQ_things": "meanSQ_things_", "PQ_stuff": "PQ_stuff_", "meanPQ_stuff": "meanPQ_stuff_",
                "meanRQ_stuff": "meanRQ_stuff_", "meanSQ_stuff": "meanSQ_stuff_"}


def check_treeins_result(z, name, r):
    """r = panoptic_evaluation_treeins(...) against the reference's log of the same case (nan where the reference logs nan)"""
    for mine, theirs in (("oAcc", "oAcc"), ("mAcc", "mAcc"), ("IoU", "IoU"), ("mIoU", "mIoU")):
        np.testing.assert_allclose(np.atleast_1d(np.asarray(r[mine], np.float64)), z["log_%s_Semantic_Segmentation_%s" % (name, theirs)],
                                   rtol=1e-6, atol=1e-7, err_msg="%s %s" % (name, mine))
    for section in ("offset", "embed"):
        for mine, theirs in TREEINS_KEYS.items():
            want = z["log_%s_%s_Instance_Segmentation_%s" % (name, section, theirs)]
            got = np.atleast_1d(np.asarray(r[section][mine], np.float64))
            assert got.shape == want.shape, (name, section, mine, got, want)
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7, equal_nan=True, err_msg="%s %s %s" % (name, section, mine))


def test_treeins_evaluation_matches_reference_final_eval():
    """panoptic_evaluation_treeins vs the numbers the reference's own FOR-instance final_eval logs
    (datasets/panoptic/treeins.py:99-497; tests/golden/make_golden.py --treeins-eval-only): both instance predictions, incl.
    the case where the embedding branch finds nothing and the case without stuff instances."""
    from panopticsegforlargescalepointcloud_amd.panoptic.metrics import panoptic_evaluation_treeins
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "treeins_eval_cases.npz"))
    for name in z["names"].tolist():
        r = panoptic_evaluation_treeins(z["pred_sem_" + name], z["pre_emb_" + name], z["pre_off_" + name], z["gt_sem_" + name],
                                        z["gt_ins_" + name])
        check_treeins_result(z, name, r)


def test_ply_io_and_checkpoint_layout(tmp_path):
    from panopticsegforlargescalepointcloud_amd import io as pio
    gold = os.path.join(os.path.dirname(__file__), "golden")
    # a file written by the reference's write_ply reads back exactly
    v = np.load(os.path.join(gold, "ref_written_npm3d_like_values.npz"))
    d = pio.read_ply(os.path.join(gold, "ref_written_npm3d_like.ply"))
    assert d.dtype.names == ("x", "y", "z", "scalar_class", "scalar_label")
    assert np.array_equal(np.stack([d["x"], d["y"], d["z"]], 1), v["xyz"]) and np.array_equal(d["scalar_class"], v["cls"])
    xyz, sem, ins = pio.read_npm3d(os.path.join(gold, "ref_written_npm3d_like.ply"))
    assert sem.dtype == torch.int64 and np.array_equal(sem.numpy(), v["cls"].astype(np.int64) - 1)
    assert np.array_equal(ins.numpy(), v["lab"].astype(np.int64) + 1)
    # round trip incl. [n,3] blocks, int64 -> int32 labels, ascii and big-endian inputs
    rng = np.random.default_rng(4)
    pos = rng.normal(size=(20, 3)).astype(np.float32)
    lab = rng.integers(-1, 9, size=20)
    p = pio.write_ply(str(tmp_path / "a"), [pos, lab, lab.astype(np.int16)], ["x", "y", "z", "preds", "gt"])
    r = pio.read_ply(p)
    assert np.array_equal(r["x"], pos[:, 0]) and r["preds"].dtype == np.int32 and np.array_equal(r["gt"], lab.astype(np.int16))
    asc = tmp_path / "b.ply"
    asc.write_text("ply\nformat ascii 1.0\ncomment hi\nelement vertex 2\nproperty float x\nproperty int l\nend_header\n1.5 3\n-2 4\n")
    r = pio.read_ply(str(asc))
    assert r["x"].tolist() == [1.5, -2.0] and r["l"].tolist() == [3, 4]
    be = tmp_path / "c.ply"
    with open(be, "wb") as f:
        f.write(b"ply\nformat binary_big_endian 1.0\nelement vertex 2\nproperty double x\nproperty ushort l\nend_header\n")
        np.array([(1.25, 7), (2.5, 9)], dtype=[("x", ">f8"), ("l", ">u2")]).tofile(f)
    r = pio.read_ply(str(be))
    assert r["x"].tolist() == [1.25, 2.5] and r["l"].tolist() == [7, 9]
    with pytest.raises(ValueError):
        pio.write_ply(str(tmp_path / "bad"), [pos], ["x", "y"])
    # checkpoint in the reference trainer's layout
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    net(torch.randn(5, 4)).sum().backward()
    opt.step()
    ck = str(tmp_path / "PointGroup-PAPER.pt")
    pio.save_checkpoint(ck, net, opt, weight_name="best_miou", run_config={"model_name": "PointGroup-PAPER"})
    raw = torch.load(ck, weights_only=False)
    assert set(raw) >= {"models", "optimizer", "schedulers", "stats", "run_config", "dataset_properties"}
    assert raw["optimizer"][0] == "Adam" and list(raw["models"]) == ["best_miou"]
    net2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    opt2 = torch.optim.Adam(net2.parameters(), lr=1e-3)
    with pytest.raises(KeyError):
        pio.load_checkpoint(ck, net2, weight_name="latest")
    missing, unexpected = pio.load_checkpoint(ck, net2, weight_name="best_miou", optimizer=opt2)
    assert not missing and not unexpected
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))


def test_gt_layout_histogram_matches_oracle():
    """torch_points_kernels.gt_layout (one (batch, id) histogram) against the oracle's per-element loops, including
    ids with no point, batch elements without instances and unsorted batch vectors."""
    from oracle import oracle
    from panopticsegforlargescalepointcloud_amd.torch_points_kernels import gt_layout
    rng = np.random.default_rng(3)
    for n, nb, kmax in [(1, 1, 0), (50, 1, 3), (4000, 4, 9), (3000, 6, 1), (2000, 3, 0)]:
        batch = rng.integers(0, nb, n)
        batch[0] = nb - 1
        gt = rng.integers(0, kmax + 1, n) * (rng.random(n) < 0.6)
        gt[(batch == 1) & (gt == 2)] = 0  # a hole in the id range of one element
        off, sizes = gt_layout(torch.from_numpy(gt), torch.from_numpy(batch))
        want_off, want_sizes = oracle.gt_layout(gt, batch)
        assert off.dtype == torch.int32 and sizes.dtype == torch.int32
        np.testing.assert_array_equal(off.numpy(), want_off)
        np.testing.assert_array_equal(sizes.numpy(), want_sizes)


def test_sparseconv3d_backend_surface():
    """the SparseConv3d.nn seam (reference modules/SparseConv3d/nn/__init__.py:21): the six names, their constructor
    defaults and what the consumer touches (`.bn.weight`, `.kernel` shapes) -- no GPU needed to build the modules."""
    import inspect
    from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as ME, sparseconv3d_nn as snn
    assert set(snn.__all__) == {"cat", "Conv3d", "Conv3dTranspose", "ReLU", "SparseTensor", "BatchNorm"}
    for cls, base in [(snn.Conv3d, ME.MinkowskiConvolution), (snn.Conv3dTranspose, ME.MinkowskiConvolutionTranspose)]:
        sig = inspect.signature(cls.__init__).parameters
        assert [p for p in sig][1:] == ["in_channels", "out_channels", "kernel_size", "stride", "dilation", "bias"]
        assert (sig["kernel_size"].default, sig["stride"].default, sig["dilation"].default, sig["bias"].default) == (3, 1, 1, False)
        m = cls(8, 16)
        assert isinstance(m, base) and tuple(m.kernel.shape) == (27, 8, 16) and m.bias is None
        assert tuple(cls(8, 16, kernel_size=1).kernel.shape) == (8, 16)
    bn = snn.BatchNorm(16)
    assert isinstance(bn, ME.MinkowskiBatchNorm) and bn.bn.weight.shape == (16,) and "BatchNorm1d" in repr(bn)
    assert isinstance(snn.ReLU(inplace=True), ME.MinkowskiReLU)
    with pytest.raises(Exception):  # no CPU execution path: the tensor has to live on a HIP device
        snn.SparseTensor(torch.zeros(4, 3), torch.zeros(4, 3, dtype=torch.int32), torch.zeros(4, dtype=torch.int64))


def test_pointgroupembed_recipes_match_the_trace_of_the_reference_functions():
    """tests/golden/embed_cluster_recipes.json: PointGroupEmbed._cluster .. _cluster16 of the reference executed with
    recording stand-ins for the clustering primitives (make_golden.make_embed_recipes).  The recipe table of
    panoptic/variants.py must name the same primitive on the same feature matrix with the same arguments, and combine the
    proposal sets in the same order with the same type codes."""
    import json
    import types as T
    from panopticsegforlargescalepointcloud_amd.panoptic.variants import PointGroupEmbed
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "embed_cluster_recipes.json")))
    assert sorted(int(k) for k in fx) == sorted(PointGroupEmbed.RECIPES) == list(range(1, 17))
    for ct, rec in fx.items():
        recipe = PointGroupEmbed.RECIPES[int(ct)]
        # the reference's calls, in OUTPUT order (tags = 1000 * call number + index inside the call)
        call_of = []
        for tag in rec["proposal_tags"]:
            if tag // 1000 not in call_of:
                call_of.append(tag // 1000)
        ref_calls = [rec["calls"][c - 1] for c in call_of]
        assert len(ref_calls) == len(recipe) == len(rec["calls"])
        parts = []
        for step, call in zip(recipe, ref_calls):
            kind = step[0]
            if kind == "H":
                assert call == ["hdbscan.cluster_single", step[1], step[2]]
                parts.append([(T.SimpleNamespace(n=1, tag=(call, 0)), step[2])])
            elif kind == "HL":
                assert call == ["hdbscan.cluster_loop", step[1], step[2], step[3], step[4]]
                parts.append([(T.SimpleNamespace(n=1, tag=(call, i)), i) for i in range(step[4])])
            elif kind == "HF":
                assert call[:2] == ["hdbscan.cluster_loop_fixedD", step[1]] and call[4] == step[2]
                parts.append([(T.SimpleNamespace(n=1, tag=(call, i)), i) for i in range(step[2])])
            elif kind == "M":
                assert call[:3] == ["meanshift.cluster_single", step[1], step[2]] and call[3].startswith("bandwidth=")
                parts.append([(T.SimpleNamespace(n=1, tag=(call, 0)), step[2])])
            elif kind == "ML":
                assert call[:2] == ["meanshift.cluster_loop", step[1]] and call[4] == step[2] and rec["raises_in_the_reference"]
                parts.append([(T.SimpleNamespace(n=1, tag=(call, i)), i) for i in range(step[2])])
            else:
                assert call == ["region_grow", "raw_pos(all points)", "ignore=[0, 1, 5]", "radius=0.3", "nsample=16", "min_cluster_size=10"]
                parts.append([(T.SimpleNamespace(n=2, tag=(call, 0)), step[1])])
        proposals, typed = PointGroupEmbed._order(int(ct), parts)
        # output order of the proposal sets = the reference's list order
        got_order = [rec["calls"].index(p.tag[0]) + 1 for p in proposals for _ in range(p.n)]
        assert got_order == [t // 1000 for t in rec["proposal_tags"]], ct
        assert [t for c, t in typed for _ in range(c.n)] == rec["types"], ct


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-runs itself as N ranks under torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1, a free port) and passes the command line through"""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(4)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_conv_kernel_family_rule():
    """which kernel a convolution shape runs on, as bench.py's per-family figures attribute it: the library's own dispatch rule
    (pp_spconv_kernel_family, a host function -- no GPU needed).  Split-operand arithmetic (bf16 matrix pipe): inputs of whole
    32-channel groups on k_spconv_x3f ("x3f": full-line gathers through LDS, from one column tile per wave up); 48 / 80 / 112-channel
    inputs, two sources of 16 (2 k + 1) channels and 16-channel inputs with >= 3 column tiles on k_spconv_x3 ("x3").  The fp32-MFMA
    kernel ("fwd3"): 16-channel inputs on <= 2 column tiles, 1x1 layers, the 4-channel input layer, inputs of 4 GiB or more, and
    everything under PP_CONV_X3=0 (read once per process: checked in a child); PP_CONV_X3F=0 puts the x3f shapes back on x3 and
    the one-column-tile ones on fwd3"""
    import subprocess
    import sys
    from panopticsegforlargescalepointcloud_amd import ops
    fam = ops.LaunchProfiler.kernel_family
    want = {(64, 64, 27): "x3f", (48, 48, 27): "x3", (128, 48, 27): "x3f", (160, 64, 27): "x3f", (96, 96, 27): "x3f", (192, 80, 27): "x3f",
            (112, 112, 27): "x3", (96, 32, 27): "x3f", (64, 32, 27): "x3f", (32, 32, 27): "x3f", (16, 16, 27): "fwd3",
            (64, 16, 27): "x3f", (32, 16, 27): "x3f", (48, 16, 27): "fwd3", (4, 16, 27): "fwd3", (96, 112, 1): "fwd3", (32, 64, 27): "x3f",
            (16, 32, 27): "fwd3", (16, 48, 27): "x3", (80, 80, 27): "x3"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("PP_CONV_X3", "1") != "0" and not os.environ.get("PP_CONV_X3_MIN_NTW") and os.environ.get("PP_CONV_X3F", "1") != "0":
        for (cin, cout, K), f in want.items():
            assert fam(cin, cout, K) == f, (cin, cout, K)
        assert fam(64, 32, 27, c1=32) == "x3f" and fam(48, 32, 27, c1=16) == "fwd3"  # two sources (ME.cat fused): equal widths only
        assert fam(96, 32, 27, c1=48) == "x3"   # ... and a source boundary inside a 32-channel group stays on the register gathers
        assert fam(64, 64, 27, n_in=1 << 26) == "fwd3"  # >= 4 GiB of input rows: not addressable by the buffer descriptors
        code = ("from panopticsegforlargescalepointcloud_amd import ops; f = ops.LaunchProfiler.kernel_family; "
                "print(f(64, 64, 27), f(64, 16, 27), f(48, 48, 27))")
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PP_CONV_X3F="0"), capture_output=True, text=True, cwd=root)
        assert out.stdout.split()[-3:] == ["x3", "fwd3", "x3"], out.stdout + out.stderr
    code = "from panopticsegforlargescalepointcloud_amd import ops; print(ops.LaunchProfiler.kernel_family(64, 64, 27))"
    env = dict(os.environ, PP_CONV_X3="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip().endswith("fwd3"), out.stdout + out.stderr
