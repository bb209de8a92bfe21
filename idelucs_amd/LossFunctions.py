"""idelucs_amd.LossFunctions -- the two training losses, restated for a device-resident step.

Same names/arguments/values as reference idelucs/LossFunctions.py (IID_loss :20-46, compute_joint
:49-62, info_nce_loss :65-98).  The reference builds its InfoNCE masks and labels on the CPU
(`torch.eye`, `torch.arange` without device, :75-76,:83) and boolean-indexes device tensors with
them -- a host sync plus D2H/H2D every step -- and materialises a [B, C, C] outer product for the
IIC joint (:57).  Here both are written as GEMM + row-wise reductions on the device: no masks, no
sync, no [B, C, C] intermediate (82 MB per step at C = 200).
"""
import sys

import torch
import torch.nn.functional as F


def compute_joint(x_out, x_tf_out):
    """Reference LossFunctions.py:49-62: P = sum_b x_out[b,:,None] * x_tf_out[b,None,:] (as one
    [C,B]x[B,C] GEMM), symmetrised and normalised to sum 1."""
    bn, k = x_out.size()
    assert x_tf_out.size(0) == bn and x_tf_out.size(1) == k
    p_i_j = x_out.t() @ x_tf_out
    p_i_j = (p_i_j + p_i_j.t()) / 2.0
    return p_i_j / p_i_j.sum()


def IID_loss(x_out, x_tf_out, lamb=1.0, EPS=sys.float_info.epsilon):
    """Reference LossFunctions.py:20-46 (IIC mutual-information loss).  Entries below EPS are
    REPLACED by the constant EPS (in-place assignment at :36-38), so no gradient flows through
    them -- hence torch.where, not clamp."""
    _, k = x_out.size()
    p_i_j = compute_joint(x_out, x_tf_out)
    p_i = p_i_j.sum(dim=1, keepdim=True)       # marginals of the UN-clamped joint (:32-33)
    p_j = p_i_j.sum(dim=0, keepdim=True)
    eps = torch.full((), EPS, dtype=p_i_j.dtype, device=p_i_j.device)
    p_i_j = torch.where(p_i_j < EPS, eps, p_i_j)
    p_j = torch.where(p_j < EPS, eps, p_j)
    p_i = torch.where(p_i < EPS, eps, p_i)
    loss = -p_i_j * (torch.log(p_i_j) - lamb * torch.log(p_j) - lamb * torch.log(p_i))
    return loss.sum()


def info_nce_loss(z1, z2, temperature):
    """Reference LossFunctions.py:65-98 (SimCLR NT-Xent): rows r of cat(z1, z2), L2-normalised;
    positive of r is (r + B) mod 2B; cross-entropy over all j != r; mean over the 2B rows.
    Re-ordering the logits as [positive | negatives] with label 0 (:88-94) does not change the value."""
    b = z1.shape[0]
    f = F.normalize(torch.cat((z1, z2), 0).float(), dim=1)
    s = (f @ f.t()) / temperature
    r = torch.arange(2 * b, device=s.device)
    pos = s[r, (r + b) % (2 * b)]
    s = s.masked_fill(r.unsqueeze(0) == r.unsqueeze(1), float("-inf"))
    return (torch.logsumexp(s, dim=1) - pos).mean()
