"""GPU: proposal overlaps, NMS and painting on the device (csrc/pp_nms.hip) vs the dense mask product, the oracle
pipeline and the reference's own get_instances output (tests/golden/nms_cases.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from panopticsegforlargescalepointcloud_amd import ops as o
    return o


def _fake_results(ops, rng, n_tiles=3, per_tile=2000):
    """proposals over `n_tiles` batch elements incl. duplicates and partial overlaps + scores"""
    from panopticsegforlargescalepointcloud_amd import ops
    clusters, batch = [], np.repeat(np.arange(n_tiles), per_tile)
    for t in range(n_tiles):
        base = t * per_tile
        for _ in range(12):
            c0 = int(rng.integers(0, per_tile - 300))
            size = int(rng.integers(5, 250))
            clusters.append(np.sort(base + c0 + rng.permutation(300)[:size]))
        clusters.append(clusters[-1].copy())  # exact duplicate (region growing and mean shift often agree)
    perm = rng.permutation(len(clusters))
    clusters = [clusters[i] for i in perm]
    scores = rng.uniform(0.3, 1.0, len(clusters)).astype(np.float32)
    csr = ops.ClusterCSR.from_list([torch.from_numpy(c) for c in clusters], "cuda")
    return clusters, scores, csr, batch




def _pairs_np(pairs):
    k = int(pairs.n_pairs.item())
    pairs.check()
    a, b, inter = (t[:k].cpu().numpy() for t in (pairs.a, pairs.b, pairs.inter))
    o = np.lexsort((b, a))
    return a[o], b[o], inter[o]


def test_pairs_nms_and_painting_match_dense_product_and_oracle(ops):
    from oracle import pipeline as opipe
    from panopticsegforlargescalepointcloud_amd.panoptic.structures import PanopticResults
    from panopticsegforlargescalepointcloud_amd.scene import instance_labels_per_tile
    for seed in (3, 4, 5):
        rng = np.random.default_rng(seed)
        clusters, scores, csr, batch = _fake_results(ops, rng)
        n = len(batch)
        a, b, inter = _pairs_np(ops.proposal_pairs(csr, n))
        dense = np.zeros((len(clusters), n), np.int64)
        for i, c in enumerate(clusters):
            dense[i, c] = 1
        full = dense @ dense.T
        wa, wb = np.nonzero(np.triu(full, 1))
        assert np.array_equal(a, wa) and np.array_equal(b, wb) and np.array_equal(inter, full[wa, wb])
        poe = ops.proposal_pairs(csr, n).prop_of_entry.cpu().numpy()
        assert np.array_equal(poe, np.repeat(np.arange(len(clusters)), [len(c) for c in clusters]))
        res = PanopticResults(semantic_logits=torch.zeros(n, 9), offset_logits=None, embed_logits=None, clusters=None,
                              cluster_scores=torch.from_numpy(scores).cuda(), mask_scores=None, cluster_type=None, clusters_csr=csr)
        labels, counts = instance_labels_per_tile(res, torch.from_numpy(batch).cuda(), 3)
        want = opipe.instance_labels({"clusters": clusters, "cluster_scores": scores}, n, batch)
        assert np.array_equal(labels.cpu().numpy(), want)
        assert counts == [len(np.unique(want[batch == t][want[batch == t] >= 0])) for t in range(3)]
        # no ScoreNet: every proposal is an instance, painted in proposal order per tile
        res0 = res._replace(cluster_scores=None)
        labels0, counts0 = instance_labels_per_tile(res0, torch.from_numpy(batch).cuda(), 3)
        want0 = np.full(n, -1, np.int32)
        tile = np.asarray([batch[c[0]] for c in clusters])
        for t in range(3):
            for r, i in enumerate(np.nonzero(tile == t)[0]):
                want0[clusters[i]] = np.maximum(want0[clusters[i]], r)
        assert np.array_equal(labels0.cpu().numpy(), want0) and counts0 == [int((tile == t).sum()) for t in range(3)]


def test_get_instances_matches_reference_golden(ops):
    """PanopticResults.get_instances on the device kernels vs the reference's own implementation."""
    from panopticsegforlargescalepointcloud_amd.panoptic.structures import PanopticResults
    z = np.load(os.path.join(ROOT, "tests", "golden", "nms_cases.npz"))
    offs = z["cluster_offsets"]
    clusters = [torch.from_numpy(z["cluster_points"][offs[i]: offs[i + 1]]).cuda() for i in range(len(offs) - 1)]
    n = int(z["cluster_points"].max()) + 1
    res = PanopticResults(semantic_logits=torch.zeros(n, 9, device="cuda"), offset_logits=None, embed_logits=None, clusters=clusters,
                          cluster_scores=torch.from_numpy(z["scores"]).cuda(), mask_scores=None, cluster_type=None)
    for tag, (thr, mn, ms) in {"default": (0.3, 100, 0.5), "tracker": (0.3, 10, 0.5), "loose": (0.6, 0, 0.0)}.items():
        ids, cl = res.get_instances(nms_threshold=thr, min_cluster_points=mn, min_score=ms)
        assert ids == z["ids_" + tag].tolist()
        assert [int(c.numel()) for c in cl] == z["sizes_" + tag].tolist()
    empty = PanopticResults(semantic_logits=torch.zeros(4, 9, device="cuda"), offset_logits=None, embed_logits=None, clusters=[],
                            cluster_scores=None, mask_scores=None, cluster_type=None)
    assert empty.get_instances() == ([], [])


def test_painting_matches_the_reference_tracker(ops):
    """NMS + painting on the device vs the labels the reference's own tracker paints (get_instances with the tracker's
    min_cluster_points, then get_cur_ins_pre_label executed by make_golden.py: `paint_tracker`), and the ScoreNet-less
    case (`paint_noscore`: every proposal, proposal order)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "nms_cases.npz"))
    offs = z["cluster_offsets"]
    n = int(z["n"])
    clusters = [torch.from_numpy(z["cluster_points"][offs[i]: offs[i + 1]]) for i in range(len(offs) - 1)]
    csr = ops.ClusterCSR.from_list(clusters, "cuda")
    batch = torch.zeros(n, dtype=torch.int64, device="cuda")
    labels, counts, _, pairs = ops.nms_paint(csr, n, batch, 1, torch.from_numpy(z["scores"]).cuda(), 0.3, 10, 0.5)
    pairs.check()
    assert np.array_equal(labels.cpu().numpy().astype(np.int64), z["paint_tracker"])
    assert counts.tolist() == [len(z["ids_tracker"])]
    labels0, counts0, _, _ = ops.nms_paint(csr, n, batch, 1, None, 0.3, 10, 0.5)
    assert np.array_equal(labels0.cpu().numpy().astype(np.int64), z["paint_noscore"]) and counts0.tolist() == [len(clusters)]


def test_equal_scores_and_many_sources(ops):
    """duplicated proposals (identical sets, identical scores) keep exactly one copy; a point in 3 proposals yields 3 pairs;
    more than 8 proposals on one point is reported."""
    pts = [np.arange(0, 50), np.arange(0, 50), np.arange(40, 90), np.arange(0, 50), np.arange(200, 260)]
    csr = ops.ClusterCSR.from_list([torch.from_numpy(p) for p in pts], "cuda")
    a, b, inter = _pairs_np(ops.proposal_pairs(csr, 300))
    assert list(zip(a.tolist(), b.tolist(), inter.tolist())) == [(0, 1, 50), (0, 2, 10), (0, 3, 50), (1, 2, 10), (1, 3, 50), (2, 3, 10)]
    scores = torch.tensor([0.9, 0.9, 0.8, 0.9, 0.7]).cuda()
    labels, counts, rank, _ = ops.nms_paint(csr, 300, None, 1, scores, 0.3, 10, 0.5)
    r = rank.cpu().numpy()
    assert counts.tolist() == [3] and sorted(r[[0, 1, 3]].tolist()) == [-1, -1, 2] and r[2] == 1 and r[4] == 0
    lab = labels.cpu().numpy()
    assert (lab[:50] == 2).all() and (lab[50:90] == 1).all() and (lab[200:260] == 0).all() and (lab[90:200] == -1).all()
    many = ops.ClusterCSR.from_list([torch.arange(5) for _ in range(9)], "cuda")
    with pytest.raises(NotImplementedError):
        ops.proposal_pairs(many, 5).check()
