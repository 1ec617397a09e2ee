"""torch_points_kernels-compatible surface (region_grow, instance_iou) on the MI355X kernels.

Signatures as used by the reference:
  region_grow(pos, labels, batch, ignore_labels, nsample=16, radius=0.02, min_cluster_size=32) -> List[LongTensor]
      call sites torch_points3d/models/panoptic/PointGroup3heads.py:166-174,185-205,296-304,340-357
  instance_iou(List[LongTensor], gt_instances, batch) -> FloatTensor[nProp, sum nGT]
      call sites torch_points3d/core/losses/panoptic_losses.py:37, metrics/panoptic_tracker_pointgroup_npm3d.py:681
Differences from the CUDA library, both invisible after the canonicalisation the parity bar allows: the points of a
cluster come back in ascending index instead of DFS order.
Use as a drop-in:  sys.modules["torch_points_kernels"] = panopticsegforlargescalepointcloud_amd.torch_points_kernels
"""
import torch

from . import ops


def region_grow_csr(pos, labels, batch, ignore_labels=None, nsample=16, radius=0.02, min_cluster_size=32, num_classes=None):
    """Device-resident variant: returns (ops.ClusterCSR, point_cluster int32 [N]) without building Python lists."""
    if ignore_labels is None:
        ignore_labels = torch.zeros(0, dtype=torch.int64)
    ignore_labels = torch.as_tensor(ignore_labels)
    if batch is None:
        batch = torch.zeros(pos.shape[0], dtype=torch.int64, device=pos.device)
    if num_classes is None:
        num_classes = int(labels.max().item()) + 1 if labels.numel() else 1
    return ops.region_grow_csr(pos.float(), labels.long(), batch.long(), ignore_labels, nsample, radius, min_cluster_size,
                               max(int(num_classes), 1))


def region_grow(pos, labels, batch, ignore_labels=[], nsample=16, radius=0.02, min_cluster_size=32):
    csr, _ = region_grow_csr(pos, labels, batch, ignore_labels, nsample, radius, min_cluster_size)
    return csr.to_list()


def gt_layout(gt_instances, batch):
    """Per-sample GT instance counts (cumulative) and sizes, the column layout of instance_iou.
    One histogram over (batch, id) gives both: ids are 1..k per batch element (0 = none), k = the largest id present."""
    dev = batch.device
    if batch.numel() == 0:
        return torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev)
    nb, g = (int(v) + 1 for v in torch.stack([batch.max(), gt_instances.max()]).tolist())
    hist = torch.bincount(batch * g + gt_instances, minlength=nb * g).view(nb, g)
    ids = torch.arange(g, device=dev)
    k = ((hist > 0) * ids).amax(1)
    offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(k, 0)])
    sizes = hist[(ids > 0) & (ids <= k.view(-1, 1))]  # row-major: batch element by batch element, ids 1..k
    return offs.to(torch.int32), sizes.to(torch.int32)


def instance_iou_csr(csr, gt_instances, batch=None):
    if batch is None:
        batch = torch.zeros_like(gt_instances)
    gt_instances = gt_instances.long()
    batch = batch.long()
    gt_off, gt_sizes = gt_layout(gt_instances, batch)
    return ops.instance_iou_csr(csr, gt_instances, batch, gt_off.contiguous(), gt_sizes.contiguous())


def instance_iou(instance_idx, gt_instances, batch=None):
    csr = ops.ClusterCSR.from_list(list(instance_idx), gt_instances.device)
    return instance_iou_csr(csr, gt_instances, batch)
