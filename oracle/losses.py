"""TEST INFRASTRUCTURE (oracle).  CPU restatement (numpy fp64) of the Lovasz-Softmax criterion of
pointcept/models/losses/lovasz.py as the ScanNet PTv3 config uses it (mode="multiclass", per_image=False,
classes="present", ignore_index=-1; scannet/semseg-pt-v3m1-0-base.py:49-52):

  _flatten_probas  (:149-166)  drop the ignored points
  _lovasz_softmax_flat (:118-146)  for every class present in the labels: fg = [label == c], errors = |fg - p_c|,
                               sort descending, dot(sorted errors, _lovasz_grad(sorted fg)); mean over those classes
  _lovasz_grad     (:22-33)    jaccard = 1 - (gts - cumsum(fg)) / (gts + cumsum(1 - fg)); first differences

Pinned by tests/golden/lovasz.npz = losses and gradients produced by the reference module itself
(tests/golden/make_golden.py).  The gradient treats the sort permutation as constant, as autograd does.
"""
from __future__ import annotations

import numpy as np


def softmax(x: np.ndarray) -> np.ndarray:
    z = x.astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def lovasz_grad(fg_sorted: np.ndarray) -> np.ndarray:
    gts = fg_sorted.sum()
    inter = gts - np.cumsum(fg_sorted)
    union = gts + np.cumsum(1.0 - fg_sorted)
    jac = 1.0 - inter / union
    jac[1:] = jac[1:] - jac[:-1]
    return jac


def lovasz_softmax(logits: np.ndarray, labels: np.ndarray, ignore_index: int = -1):
    """-> (loss, dloss/dlogits [N, C]) in fp64."""
    n, c = logits.shape
    valid = labels != ignore_index
    dlogits = np.zeros((n, c), dtype=np.float64)
    if not valid.any():
        return 0.0, dlogits
    p = softmax(logits[valid])
    lab = labels[valid]
    present = np.unique(lab)
    gp = np.zeros_like(p)
    losses = []
    for cls in present:
        fg = (lab == cls).astype(np.float64)
        err = np.abs(fg - p[:, cls])
        perm = np.argsort(-err, kind="stable")
        g = lovasz_grad(fg[perm])
        losses.append(float(np.dot(err[perm], g)))
        gp[perm, cls] = g * np.where(fg[perm] > 0, -1.0, 1.0)
    gp /= len(present)
    dz = p * (gp - (gp * p).sum(axis=1, keepdims=True))
    dlogits[valid] = dz
    return float(np.mean(losses)), dlogits


def lovasz_softmax_torch(logits, labels, ignore_index: int = -1):
    """Same criterion as a differentiable torch expression (fp32/fp64 follows `logits`), for the oracle MODEL's loss
    (SegmentorV2(criteria=("ce", "lovasz"))): lovasz.py:118-146 / :22-33 / :149-166 line by line, sort permutation
    constant under autograd as in the reference."""
    import torch

    valid = labels != ignore_index
    probas = torch.softmax(logits.float() if logits.dtype in (torch.float16, torch.bfloat16) else logits, dim=1)[valid]
    lab = labels[valid]
    if probas.numel() == 0:
        return (probas * 0.0).sum()
    losses = []
    for c in lab.unique():
        fg = (lab == c).to(probas.dtype)
        errors = (fg - probas[:, c]).abs()
        errors_sorted, perm = torch.sort(errors, 0, descending=True)
        fg_sorted = fg[perm]
        gts = fg_sorted.sum()
        inter = gts - fg_sorted.cumsum(0)
        union = gts + (1.0 - fg_sorted).cumsum(0)
        jac = 1.0 - inter / union
        if jac.numel() > 1:
            jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
        losses.append(torch.dot(errors_sorted, jac))
    return torch.stack(losses).mean()
