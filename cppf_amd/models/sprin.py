"""Drop-in for the reference's models/sprin.py (the pieces PointEncoder is built from).

Module names, constructor signatures and parameter names follow the reference
(`conv_kernel` -> nn.Sequential of Linear/LayerNorm/ReLU, models/sprin.py:64-72;
`GlobalInfoProp.linear`, :75-84; `SparseSO3Conv.{kernel,outnet,layer_norm}`, :87-107), so
`point_encoder_epoch*.pth` checkpoints load unchanged (nocs/inference.py:87).  The `forward` methods here
are the torch composite used when autograd needs a graph (train.py:64); inference goes through
`cppf_amd.models.model.PointEncoder`, which runs the whole stack in HIP (csrc/sprin.hip).
"""
import numpy as np
import torch
import torch.nn as nn

__all__ = ["rifeat", "conv_kernel", "GlobalInfoProp", "SparseSO3Conv", "pack_point_encoder"]


def _unit_cos(u, v, nu, nv):
    """cosine between two edge vectors with the reference's 1e-7 guard (models/sprin.py:56-58)"""
    return (u * v).sum(-1, keepdim=True) / (nu * nv + 1e-7)


def rifeat(points_r, points_s):
    """Six rotation-invariant numbers per neighbour (models/sprin.py:40-61): the triangle (neighbour r, centre s,
    neighbour mean m) described by its three edge lengths |m-r|, |r-s|, |s-m| and the cosines at its corners.
    points_r [B,N,K,3], points_s [B,N,1,3] -> [B,N,K,6]."""
    if points_r.shape[1] != points_s.shape[1]:
        points_r = points_r.expand(-1, points_s.shape[1], -1, -1)
    centroid = points_r.mean(dim=-2, keepdim=True)
    e_mr, e_rs, e_sm = centroid - points_r, points_r - points_s, points_s - centroid
    n_mr = torch.linalg.vector_norm(e_mr, dim=-1, keepdim=True)
    n_rs = torch.linalg.vector_norm(e_rs, dim=-1, keepdim=True)
    n_sm = torch.linalg.vector_norm(e_sm, dim=-1, keepdim=True).expand_as(n_rs)
    return torch.cat([n_mr, n_rs, n_sm, _unit_cos(e_mr, e_rs, n_mr, n_rs), _unit_cos(e_rs, e_sm, n_rs, n_sm),
                      _unit_cos(e_sm, e_mr, n_sm, n_mr)], dim=-1)


def conv_kernel(iunit, ounit, *hunits):
    """nn.Sequential of (Linear, LayerNorm, ReLU) per hidden width, then Linear -- the layout whose indices
    (0, 1, 3, 4, ...) the reference checkpoints use (models/sprin.py:64-72)."""
    widths = [iunit, *hunits]
    stack = []
    for w_in, w_out in zip(widths[:-1], widths[1:]):
        stack.extend((nn.Linear(w_in, w_out), nn.LayerNorm(w_out), nn.ReLU()))
    stack.append(nn.Linear(widths[-1], ounit))
    return nn.Sequential(*stack)


class GlobalInfoProp(nn.Module):
    """models/sprin.py:75-84: append the per-channel maximum over all points of a linear map of the features."""

    def __init__(self, n_in, n_global):
        super().__init__()
        self.linear = nn.Linear(n_in, n_global)

    def forward(self, feat):
        pooled = self.linear(feat).amax(dim=-2, keepdim=True)
        return torch.cat([feat, pooled.expand(*feat.shape[:-1], pooled.shape[-1])], dim=-1)


class SparseSO3Conv(nn.Module):
    """models/sprin.py:87-107: per-neighbour kernel MLP on the rotation-invariant features, rank-`rank` contraction with
    the neighbour features, output linear, LayerNorm."""

    def __init__(self, rank, n_in, n_out, *kernel_interns, layer_norm=True):
        super().__init__()
        self.rank = rank
        self.kernel = conv_kernel(6, rank, *kernel_interns)
        self.outnet = nn.Linear(rank * n_in, n_out)
        self.layer_norm = nn.LayerNorm(n_out) if layer_norm else None

    def forward(self, feat_points, feat, eval_points):
        weights = self.kernel(rifeat(feat_points, eval_points.unsqueeze(-2)))          # [B,N,K,rank]
        mixed = torch.einsum("bnkr,bnki->bnri", weights.reshape(*feat.shape[:-1], self.rank), feat)
        out = self.outnet(mixed.flatten(-2))
        return out if self.layer_norm is None else self.layer_norm(out)


def pack_point_encoder(sd, num_layers):
    """state_dict (numpy values) -> (packed f32 array in the layout of include/cppf.h, descriptor dict)."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    parts, desc = [], None
    for l in range(num_layers):
        pre = f"spconvs.{l}.kernel."
        lins = sorted(i for i in {int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre)}
                      if g(f"{pre}{i}.weight").ndim == 2)
        hidden = []
        for i in lins[:-1]:
            W = g(f"{pre}{i}.weight")
            parts += [W.ravel(), g(f"{pre}{i}.bias"), g(f"{pre}{i + 1}.weight"), g(f"{pre}{i + 1}.bias")]
            hidden.append(int(W.shape[0]))
        Wk = g(f"{pre}{lins[-1]}.weight")
        parts += [Wk.ravel(), g(f"{pre}{lins[-1]}.bias")]
        rank = int(Wk.shape[0])
        Wo = g(f"spconvs.{l}.outnet.weight")
        n_out, n_in = int(Wo.shape[0]), int(Wo.shape[1]) // rank
        if f"spconvs.{l}.layer_norm.weight" not in sd:
            raise ValueError("SparseSO3Conv(layer_norm=False) has no device kernel")
        parts += [np.ascontiguousarray(Wo.T).ravel(), g(f"spconvs.{l}.outnet.bias"),
                  g(f"spconvs.{l}.layer_norm.weight"), g(f"spconvs.{l}.layer_norm.bias")]
        Wa = g(f"aggrs.{l}.linear.weight")
        parts += [Wa.ravel(), g(f"aggrs.{l}.linear.bias")]
        if l == 0:
            desc = dict(hidden=hidden, rank=rank, n_nbr_feats=n_in, n_out=n_out, n_glob=int(Wa.shape[0]),
                        num_layers=num_layers)
    return np.concatenate(parts).astype(np.float32), desc
