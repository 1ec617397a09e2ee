"""Drop-in for torch_points3d/utils/meanshift_cluster.py on the MI355X mean-shift kernels.

Same functions and return conventions as the reference wrapper:
  meanshift_cluster(prediction, bandwidth) -> LongTensor labels                 (reference :9-18)
  cluster_single(embed_logits_u, unique_in_batch, label_batch, local_ind, type, bandwidth)
      -> (List[LongTensor] of point indices, List[int] cluster types)          (reference :72-123)
  cluster_loop(embed_logits_u, unique_in_batch, label_batch, local_ind, low, high, loop_num, bandwidth=None)
      -> the same pair over loop_num random 5-feature subsets                    (reference :20-70)
      The reference's loop hands `meanshift_cluster` to Pool.map with the samples alone (:57) although the function
      takes (prediction, bandwidth) (:9): it raises TypeError there, i.e. no reference run of it exists.  Here it takes
      the bandwidth as a keyword and raises the same TypeError when it is not given.
but all samples (cylinders) of the batch are clustered together on the GPU -- no multiprocessing.Pool, no
device->host->worker round trip.  Indices are returned on the input's device.
"""
import torch

from .. import ops


def meanshift_cluster(prediction, bandwidth):
    x = torch.as_tensor(prediction, dtype=torch.float32)
    if not x.is_cuda:
        x = x.cuda()
    labels, _, _ = ops.meanshift(x.contiguous(), [0, x.shape[0]], bandwidth, min_points_exclusive=-1)
    return labels.long()


def cluster_single_csr(embed_logits_u, label_batch, local_ind, bandwidth, min_points_exclusive=3):
    """Device-resident core: returns (ops.ClusterCSR over local_ind values, clusters per sample tensor)."""
    dev = embed_logits_u.device
    label_batch = label_batch.to(dev).long()
    local_ind = local_ind.to(dev).long()
    m = embed_logits_u.shape[0]
    if m == 0:
        return ops.ClusterCSR(torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int64, device=dev), 0)
    # samples must be contiguous; PyG batches are sorted, but do not rely on it: the run heads of a sorted batch vector
    # are strictly increasing (checked on the host from the read that also brings the run boundaries).  Runs, run boundaries
    # and the run of every point come from the library's own scan (ops.run_lengths; round 5: torch.unique_consecutive / cumsum /
    # repeat_interleave dispatched rocPRIM kernels here)
    def runs_of(lb):
        heads_d, starts_d, run_id, n_runs = ops.run_lengths(lb.contiguous())
        nr = int(n_runs.item())
        vals = torch.cat([heads_d[:nr], starts_d[: nr + 1].long()]).tolist()
        return vals[:nr], vals[nr:], run_id
    heads, offs, sample_of_point = runs_of(label_batch)
    if any(b <= a for a, b in zip(heads, heads[1:])):
        order = torch.sort(label_batch, stable=True)[1]
        embed_logits_u, label_batch, local_ind = embed_logits_u[order], label_batch[order], local_ind[order]
        heads, offs, sample_of_point = runs_of(label_batch)
    labels, ncl, _ = ops.meanshift(embed_logits_u.detach().float().contiguous(), offs, bandwidth,
                                   min_points_exclusive=min_points_exclusive)
    # global cluster id = (clusters of earlier samples) + label ; samples ascending, labels ascending
    base, n_total = ops.exclusive_scan(ncl.to(torch.int32).contiguous(), want_total=True)
    key = torch.where(labels >= 0, labels + base[sample_of_point.long()], labels)
    # The number of clusters is on the device.  Group with an upper bound instead of reading it first (a sample of p points has
    # at most p clusters), and read it together with the counts the grouping returns: one host read where there were two
    bound = m
    goffs, out, total = ops.group_by_key(key.contiguous(), bound, ids=local_ind.contiguous())
    # sklearn can leave a centre without points; torch.unique in the reference wrapper skips such labels
    sizes = goffs[1:] - goffs[:-1]
    keep = sizes > 0
    n_groups, n_keep, kept, bad = torch.cat([n_total.view(1), keep.sum().view(1).to(torch.int32), total]).tolist()
    if bad:
        raise ops._lib.PanopticHipError("group_by_key: %d keys outside [0, n_groups)" % bad)
    if n_keep == n_groups:
        return ops.ClusterCSR(goffs[: n_groups + 1], out[:kept], n_groups)
    sizes, keep = sizes[:n_groups], keep[:n_groups]
    new_offs = torch.cat([torch.zeros(1, dtype=torch.int32, device=dev), torch.cumsum(sizes[keep], 0).to(torch.int32)])
    return ops.ClusterCSR(new_offs, out[:kept], n_keep)


def cluster_single(embed_logits_logits_u, unique_in_batch, label_batch, local_ind, type, bandwidth):
    csr = cluster_single_csr(embed_logits_logits_u, label_batch, local_ind, bandwidth)
    clusters = csr.to_list()
    return clusters, [type] * len(clusters)


def loop_csr(x, label_batch, local_ind, loop_num, bandwidth):
    """[(ops.ClusterCSR, loop index)]: loop_num subsets of 5 features drawn with torch.multinomial on the CPU generator
    (reference :33-36; `pick_num = 5`, low / high unused), batch elements with more than 5 points"""
    parts = []
    for loop_i in range(loop_num):
        feature_choose = torch.multinomial(torch.ones(x.shape[-1]), 5, replacement=False)
        parts.append((cluster_single_csr(x[:, feature_choose.to(x.device)], label_batch, local_ind, bandwidth,
                                         min_points_exclusive=5), loop_i))
    return parts


def cluster_loop(embed_logits_logits_u, unique_in_batch, label_batch, local_ind, low, high, loop_num, bandwidth=None):
    if bandwidth is None:
        raise TypeError("meanshift_cluster() missing 1 required positional argument: 'bandwidth'")
    final_result, cluster_type = [], []
    for csr, loop_i in loop_csr(embed_logits_logits_u, label_batch, local_ind, loop_num, bandwidth):
        clusters = csr.to_list()
        final_result += clusters
        cluster_type += [loop_i] * len(clusters)
    return final_result, cluster_type
