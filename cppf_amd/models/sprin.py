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


def rifeat(points_r, points_s):
    """models/sprin.py:40-61: six rotation-invariant features of (neighbour, centre, neighbour mean)."""
    if points_r.shape[1] != points_s.shape[1]:
        points_r = points_r.expand(-1, points_s.shape[1], -1, -1)
    r_mean = points_r.mean(-2, keepdim=True)
    l1, l2, l3 = r_mean - points_r, points_r - points_s, points_s - r_mean
    l1n, l2n = l1.norm(dim=-1, keepdim=True), l2.norm(dim=-1, keepdim=True)
    l3n = l3.norm(dim=-1, keepdim=True).expand_as(l2n)
    th1 = (l1 * l2).sum(-1, keepdim=True) / (l1n * l2n + 1e-7)
    th2 = (l2 * l3).sum(-1, keepdim=True) / (l2n * l3n + 1e-7)
    th3 = (l3 * l1).sum(-1, keepdim=True) / (l3n * l1n + 1e-7)
    return torch.cat([l1n, l2n, l3n, th1, th2, th3], -1)


def conv_kernel(iunit, ounit, *hunits):
    layers = []
    for unit in hunits:
        layers += [nn.Linear(iunit, unit), nn.LayerNorm(unit), nn.ReLU()]
        iunit = unit
    layers.append(nn.Linear(iunit, ounit))
    return nn.Sequential(*layers)


class GlobalInfoProp(nn.Module):
    def __init__(self, n_in, n_global):
        super().__init__()
        self.linear = nn.Linear(n_in, n_global)

    def forward(self, feat):
        tran = self.linear(feat)
        glob = tran.max(-2, keepdim=True)[0].expand(*feat.shape[:-1], tran.shape[-1])
        return torch.cat([feat, glob], -1)


class SparseSO3Conv(nn.Module):
    def __init__(self, rank, n_in, n_out, *kernel_interns, layer_norm=True):
        super().__init__()
        self.kernel = conv_kernel(6, rank, *kernel_interns)
        self.outnet = nn.Linear(rank * n_in, n_out)
        self.rank = rank
        self.layer_norm = nn.LayerNorm(n_out) if layer_norm else None

    def forward(self, feat_points, feat, eval_points):
        r_inv_s = rifeat(feat_points, eval_points.unsqueeze(-2))
        kern = self.kernel(r_inv_s).reshape(*feat.shape[:-1], self.rank)
        conv = self.outnet(torch.einsum("bnkr,bnki->bnri", kern, feat).flatten(-2))
        return conv if self.layer_norm is None else self.layer_norm(conv)


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
