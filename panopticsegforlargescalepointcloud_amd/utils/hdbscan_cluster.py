"""Drop-in for torch_points3d/utils/hdbscan_cluster.py on the MI355X HDBSCAN kernels (csrc/pp_hdbscan.hip).

Same functions and return conventions as the reference wrapper:
  hdbscan_cluster(prediction) -> LongTensor labels (-1 = noise)                                   (reference :8-13)
  cluster_single(embed_logits_u, unique_in_batch, label_batch, local_ind, type)                    (reference :117-167)
  cluster_loop(embed_logits_u, unique_in_batch, label_batch, local_ind, low, high, loop_num)       (reference :15-64)
  cluster_loop_fixedD(...)                                                                          (reference :66-115)
      -> (List[LongTensor] of point indices, List[int] cluster types)
All samples of the batch are clustered in one launch sequence on the GPU; no multiprocessing.Pool.
The reference fixes min_cluster_size=15, min_samples=5, cluster_selection_epsilon=0.006; COUNT_SELF selects the
core-distance convention (False: the point itself is not one of its min_samples neighbours -- the behaviour recalled for
hdbscan 0.8.27's Boruvka path, which `algorithm='best'` takes for this data; True: sklearn / hdbscan's Prim paths).
"""
import numpy as np
import torch

from .. import ops

MIN_CLUSTER_SIZE = 15
MIN_SAMPLES = 5
CLUSTER_SELECTION_EPSILON = 0.006
COUNT_SELF = False


def hdbscan_cluster(prediction):
    x = torch.as_tensor(prediction, dtype=torch.float32)
    if not x.is_cuda:
        x = x.cuda()
    labels, _ = ops.hdbscan(x.contiguous(), [0, x.shape[0]], MIN_CLUSTER_SIZE, MIN_SAMPLES, CLUSTER_SELECTION_EPSILON,
                            COUNT_SELF, min_points_exclusive=-1)
    return labels.long()


def cluster_csr(x, label_batch, local_ind, min_points_exclusive):
    """Device-resident core: (ops.ClusterCSR over local_ind values) for one feature matrix; clusters are ordered by
    sample, then by label -- the order the reference appends them in."""
    dev = x.device
    label_batch = label_batch.to(dev).long()
    local_ind = local_ind.to(dev).long()
    m = x.shape[0]
    if m == 0:
        return ops.ClusterCSR(torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int64, device=dev), 0)
    if bool((label_batch[1:] < label_batch[:-1]).any()):
        order = torch.sort(label_batch, stable=True)[1]
        x, label_batch, local_ind = x[order], label_batch[order], local_ind[order]
    _, counts = torch.unique_consecutive(label_batch, return_counts=True)
    offs = [0] + torch.cumsum(counts, 0).tolist()
    labels, ncl = ops.hdbscan(x.detach().float().contiguous(), offs, MIN_CLUSTER_SIZE, MIN_SAMPLES,
                              CLUSTER_SELECTION_EPSILON, COUNT_SELF, min_points_exclusive=min_points_exclusive)
    base = torch.cumsum(ncl, 0) - ncl
    sample_of_point = torch.repeat_interleave(torch.arange(len(offs) - 1, device=dev), counts)
    key = torch.where(labels >= 0, labels + base[sample_of_point].to(torch.int32), labels)
    n_groups = int(ncl.sum().item())
    goffs, out, total = ops.group_by_key(key.contiguous(), n_groups, ids=local_ind.contiguous())
    return ops.ClusterCSR(goffs, out[: ops.group_by_key_check(total)], n_groups)


def cluster_single(embed_logits_logits_u, unique_in_batch, label_batch, local_ind, type):
    csr = cluster_csr(embed_logits_logits_u, label_batch, local_ind, 3)
    clusters = csr.to_list()
    return clusters, [type] * len(clusters)


def loop_csr(x, label_batch, local_ind, picks):
    """[(ops.ClusterCSR, loop index)] for the feature subsets of sizes `picks`; the subsets are drawn as the reference
    draws them (torch.multinomial on the CPU generator, :32 / :83), one HDBSCAN launch sequence per subset over all
    batch elements with more than 5 points"""
    parts = []
    for loop_i, k in enumerate(picks):
        feature_choose = torch.multinomial(torch.ones(x.shape[-1]), int(k), replacement=False)
        parts.append((cluster_csr(x[:, feature_choose.to(x.device)], label_batch, local_ind, 5), loop_i))
    return parts


def loop_picks(low, high, loop_num):
    """subset sizes of cluster_loop: numpy's global generator, as in the reference (:28)"""
    return np.random.randint(low=low, high=high + 1, size=loop_num)


def _loop(x, label_batch, local_ind, picks):
    final_result, cluster_type = [], []
    for csr, loop_i in loop_csr(x, label_batch, local_ind, picks):
        clusters = csr.to_list()
        final_result += clusters
        cluster_type += [loop_i] * len(clusters)
    return final_result, cluster_type


def cluster_loop(embed_logits_logits_u, unique_in_batch, label_batch, local_ind, low, high, loop_num):
    return _loop(embed_logits_logits_u, label_batch, local_ind, loop_picks(low, high, loop_num))


def cluster_loop_fixedD(embed_logits_logits_u, unique_in_batch, label_batch, local_ind, low, high, loop_num):
    return _loop(embed_logits_logits_u, label_batch, local_ind, [5] * loop_num)
