"""Loss and metric surface of the reference's model/loss.py, backed by the HIP Chamfer kernels.

    compute_truncated_chamfer_distance(x, y, ..., trunc=0.2)   (/root/reference/model/loss.py:94-258)
    compute_flow_metrics(flow, flow_gt, overlap=None)          (loss.py:431-471)
    scene_flow_metrics(pred, labels, strict=0.025, relax=0.05) (loss.py:382-403)
    landmark_cost(x, y)                                        (loss.py:348-351)

The nearest-neighbour search and the L1 reduction run in libndp_hip.so (no pytorch3d, no torch
fallback for GPU tensors).  Normals / Pointclouds inputs are not part of any caller of the hot
path and are rejected.
"""
import torch

from . import ops


def _lengths(points, lengths, name):
    if not torch.is_tensor(points):
        raise ValueError("The input pointclouds should be torch.Tensor of shape (minibatch, num_points, 3).")
    if points.ndim != 3:
        raise ValueError("Expected points to be of shape (N, P, D)")
    n, p = points.shape[0], points.shape[1]
    if lengths is None:
        return [p] * n
    if lengths.ndim != 1 or lengths.shape[0] != n:
        raise ValueError("Expected lengths to be of shape (N,)")
    return [int(v) for v in lengths.tolist()]


def compute_truncated_chamfer_distance(x, y, x_lengths=None, y_lengths=None, x_normals=None, y_normals=None,
                                       weights=None, trunc=0.2, batch_reduction="mean", point_reduction="mean"):
    """Bidirectional L1 Chamfer on exact 1-NN with truncation in SQUARED units.  Differentiable
    with respect to x (the only use on the hot path: registration.py:195,212)."""
    if batch_reduction is not None and batch_reduction not in ["mean", "sum"]:
        raise ValueError('batch_reduction must be one of ["mean", "sum"] or None')
    if point_reduction not in ["mean", "sum"]:
        raise ValueError('point_reduction must be one of ["mean", "sum"]')
    if x_normals is not None or y_normals is not None:
        raise NotImplementedError("normals are not used by the NDP path and are not supported")
    xl = _lengths(x, x_lengths, "x")
    yl = _lengths(y, y_lengths, "y")
    N, _, D = x.shape
    if y.shape[0] != N or y.shape[2] != D:
        raise ValueError("y does not have the correct shape.")
    if D != 3:
        raise ValueError("only 3-D point clouds are supported")
    if weights is not None:
        if weights.size(0) != N:
            raise ValueError("weights must be of shape (N,).")
        if not (weights >= 0).all():
            raise ValueError("weights cannot be negative.")
        if weights.sum() == 0.0:
            z = (x.sum((1, 2)) * weights).sum() * 0.0
            return z if batch_reduction in ["mean", "sum"] else (x.sum((1, 2)) * weights) * 0.0
    per_batch = []
    for b in range(N):
        xb, yb = x[b, :xl[b]], y[b, :yl[b]]
        v = ops.chamfer_distance(xb, yb, trunc, point_sum=point_reduction == "sum")   # sum_x[/len_x] + sum_y[/len_y] (loss.py:233-235)
        if weights is not None:
            v = v * weights[b]
        per_batch.append(v)
    out = torch.stack(per_batch)
    if batch_reduction is None:
        return out
    out = out.sum()
    if batch_reduction == "mean":
        out = out / (weights.sum() if weights is not None else N)
    return out


def landmark_cost(x, y):
    return torch.mean(torch.sum((x - y) ** 2, dim=-1))


def scene_flow_metrics(pred, labels, strict=0.025, relax=0.05):
    """EPE3D (x100 -> cm), AccS, AccR, Outlier (x100 -> %) on CPU, as upstream."""
    pred, labels = pred.detach().float().cpu(), labels.detach().float().cpu()
    err = torch.linalg.vector_norm(pred - labels, dim=1)
    rel = err / (torch.linalg.vector_norm(labels, dim=1) + 1e-20)
    epe = err.mean().item()
    acc_s = ((err < strict) | (rel < strict)).float().mean().item()
    acc_r = ((err < relax) | (rel < relax)).float().mean().item()
    outlier = (rel > 0.3).float().mean().item()
    return epe * 100, acc_s * 100, acc_r * 100, outlier * 100


def _metrics_on_device(flow, flow_gt, overlap):
    sums = ops.flow_metrics(flow.detach().float().contiguous(), flow_gt.detach().float().contiguous(), overlap)
    info = {}
    for tag, row in zip(("full", "vis", "occ") if overlap is not None else ("full",), sums):
        cnt = row[4].item()
        if cnt == 0:                                              # upstream: mean over an empty selection
            vals = [float("nan")] * 4
        else:
            vals = [float(torch.tensor(row[0].item() / cnt, dtype=torch.float32)) * 100] + [100.0 * row[k].item() / cnt for k in (1, 2, 3)]
        info.update({f"{tag}-epe": vals[0], f"{tag}-AccS": vals[1], f"{tag}-AccR": vals[2], f"{tag}-outlier": vals[3]})
    return info


def compute_flow_metrics(flow, flow_gt, overlap=None):
    """loss.py:431-471.  GPU tensors are reduced on the device (k_flow_metrics: one launch, 15 sums back); CPU tensors take
    the torch path upstream takes."""
    if torch.is_tensor(flow) and flow.is_cuda and flow.shape[0] > 0:
        return _metrics_on_device(flow, flow_gt.to(flow.device), overlap)
    info = {}
    subsets = [("full", None)]
    if overlap is not None:
        subsets += [("vis", overlap), ("occ", ~overlap)]
    for tag, mask in subsets:
        f, g = (flow, flow_gt) if mask is None else (flow[mask], flow_gt[mask])
        epe, acc_s, acc_r, outlier = scene_flow_metrics(f, g)
        info.update({f"{tag}-epe": epe, f"{tag}-AccS": acc_s, f"{tag}-AccR": acc_r, f"{tag}-outlier": outlier})
    return info
