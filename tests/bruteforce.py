"""Independent brute-force NumPy/torch definitions used to cross-check the CPU oracle
(the reference ships no vectors for MinkowskiEngine / torch-points-kernels -- SURVEY.md 8c)."""
import numpy as np


def surface_coords(rng, n_batch=2, n=500, extent=24, dup=False):
    """surface-like int32 COO rows (b,x,y,z), unique unless dup=True; random row order."""
    rows = []
    for b in range(n_batch):
        xy = rng.integers(-extent, extent, size=(n * 2, 2))
        z = np.round(0.15 * xy[:, 0] + 0.1 * np.sin(xy[:, 1] / 3.0) * 4 + rng.normal(0, 0.6, n * 2)).astype(np.int64)
        c = np.concatenate([xy, z[:, None]], 1)
        c = np.unique(c, axis=0)
        c = c[rng.permutation(len(c))[:n]]
        rows.append(np.concatenate([np.full((len(c), 1), b), c], 1))
    coords = np.concatenate(rows).astype(np.int32)
    if dup:
        extra = coords[rng.integers(0, len(coords), size=len(coords) // 10)]
        coords = np.concatenate([coords, extra])
        coords = coords[rng.permutation(len(coords))]
    return coords


def stride_coords_ref(coords, ts):
    q = coords.copy()
    q[:, 1:] = np.floor_divide(coords[:, 1:], ts) * ts
    uniq, first, inv = np.unique(q, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(uniq), np.int64)
    rank[order] = np.arange(len(uniq))
    return q[np.sort(first)], rank[inv.reshape(-1)]


def kernel_map_ref(out_coords, in_coords, ksize, step, sign):
    table = {tuple(c): i for i, c in reversed(list(enumerate(in_coords.tolist())))}
    K = ksize ** 3
    nbr = np.full((K, len(out_coords)), -1, np.int32)
    for o, c in enumerate(out_coords.tolist()):
        for k in range(K):
            if ksize == 3:
                d = (k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1)
            else:
                d = (0, 0, 0)
            key = (c[0], c[1] + sign * d[0] * step, c[2] + sign * d[1] * step, c[3] + sign * d[2] * step)
            nbr[k, o] = table.get(key, -1)
    return nbr


def region_grow_ref(pos, labels, batch, ignore_labels, nsample, radius, min_cluster_size):
    """Literal SURVEY.md App. C algorithm with O(N^2) neighbour lists (float32 distances, fma order as oracle)."""
    pos = np.asarray(pos, np.float32)
    clusters = []
    r2 = np.float32(radius) * np.float32(radius)
    for l in sorted(np.unique(labels).tolist()):
        if l in ignore_labels:
            continue
        idx = np.nonzero(labels == l)[0]
        p = pos[idx]
        b = batch[idx]
        M = len(idx)
        nbr = np.full((M, nsample), -1, np.int64)
        for a in range(M):
            d = p - p[a]
            dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
            # same op order as the oracle: fma(dz,dz, fma(dy,dy, dx*dx)) evaluated in float64-exact products
            t0 = (dx.astype(np.float32) * dx.astype(np.float32)).astype(np.float32)
            t1 = (dy.astype(np.float64) * dy.astype(np.float64) + t0.astype(np.float64)).astype(np.float32)
            d2 = (dz.astype(np.float64) * dz.astype(np.float64) + t1.astype(np.float64)).astype(np.float32)
            cand = np.nonzero((d2 < r2) & (b == b[a]))[0][:nsample]
            nbr[a, : len(cand)] = cand
        visited = np.zeros(M, bool)
        for s in range(M):
            if visited[s]:
                continue
            stack = [s]
            cl = [s]
            visited[s] = True
            while stack:
                k = stack.pop()
                for j in nbr[k]:
                    if j < 0:
                        break
                    if not visited[j]:
                        visited[j] = True
                        stack.append(j)
                        cl.append(j)
            if len(cl) >= min_cluster_size:
                clusters.append(np.sort(idx[np.asarray(cl)]))
    return clusters


def canon_partition(labels):
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


def canon_clusters(clusters):
    cl = [np.sort(np.asarray(c, np.int64)) for c in clusters]
    return sorted([c.tolist() for c in cl], key=lambda c: (c[0], len(c)))


def scaled_err(name, got, want):
    """measured parity of a float output: max |got - want| as a fraction of the output's magnitude (north_star: 1e-4 fp32);
    printed so that the GPU test log carries the numbers"""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want)
    scale = max(1.0, float(np.abs(want).max()))
    print("parity %-26s max abs err %.3e  output magnitude %.3f  -> %.3e of it   (rms err %.2e)" % (
        name, err.max(), scale, err.max() / scale, np.sqrt((err ** 2).mean())))
    return err.max() / scale


class spread_scorer_head:
    """with spread_scorer_head(lin, scores): ... -- rescales the ScorerHead's Linear layer from the scores of a first pass so
    that the proposals' logits spread to mean 2.1 / unit variance (scores over roughly (0.55, 0.99)): a random-init head
    squeezes all scores into a ~1e-3 band where float rounding decides the NMS / paint order.  Restores the layer on exit."""

    def __init__(self, lin, scores):
        import torch
        self.lin, self.w0, self.b0 = lin, lin.weight.detach().clone(), lin.bias.detach().clone()
        sc = scores.detach().double().clamp(1e-9, 1 - 1e-9)
        z = torch.log(sc / (1 - sc)) - float(self.b0[0])  # w . f of every proposal
        self.alpha = 1.0 / max(float(z.std()), 1e-9)
        self.beta = 2.1 - self.alpha * float(z.mean())

    def __enter__(self):
        import torch
        with torch.no_grad():
            self.lin.weight.copy_(self.w0 * self.alpha)
            self.lin.bias.fill_(self.beta)
        return self

    def __exit__(self, *exc):
        import torch
        with torch.no_grad():
            self.lin.weight.copy_(self.w0)
            self.lin.bias.copy_(self.b0)
        return False


def near_tie_points(clusters, scores, n_points, nms_threshold=0.3, eps=1e-5, min_score=0.5, other_scores=None):
    """Points whose instance label is decided by a score comparison closer than `eps`: two overlapping proposals
    (IoU > nms_threshold; region growing and mean shift often return NEARLY the same point set for one object, whose
    max-pooled scorer features -- hence scores -- then agree to the last bits whatever the scorer head's scale) with
    |score_i - score_j| < eps, or a score within eps of the `min_score` filter.  Which of such a pair survives the NMS is
    decided by float rounding, legitimately differently in two correct implementations; everything else must agree.
    other_scores: the second implementation's scores; a pair that is EXACTLY tied in both is ordered by the same rule in both
    (descending index) and stays in the comparison."""
    amb = np.zeros(n_points, bool)
    other = None if other_scores is None else np.asarray(other_scores, np.float64)
    if not clusters:
        return amb
    scores = np.asarray(scores, np.float64)
    owner = [[] for _ in range(n_points)]
    for i, c in enumerate(clusters):
        for p in np.asarray(c).tolist():
            owner[p].append(i)
    pairs = {}
    for lst in owner:
        for a in range(len(lst)):
            for b in range(a + 1, len(lst)):
                pairs[(lst[a], lst[b])] = pairs.get((lst[a], lst[b]), 0) + 1
    for (i, j), inter in pairs.items():
        iou = inter / (len(clusters[i]) + len(clusters[j]) - inter)
        if iou > nms_threshold and abs(scores[i] - scores[j]) < eps:
            if other is not None and scores[i] == scores[j] and other[i] == other[j]:
                continue
            amb[np.asarray(clusters[i])] = True
            amb[np.asarray(clusters[j])] = True
    for i in np.nonzero(np.abs(scores - min_score) < eps)[0]:
        amb[np.asarray(clusters[i])] = True
    return amb
