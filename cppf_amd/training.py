"""The training step of the reference (train.py:53-92) on posed synthetic objects: targets of utils/dataset.py:27-60,229-246
computed on the device, the loss of train.py:68-87, Adam over both encoders -- forward AND backward of the point encoder and
of the pair encoder run on the HIP kernels (csrc/sprin*.hip, csrc/pair_mlp*.hip through the autograd Functions of
cppf_amd/models/model.py).  Used by scripts/train_synthetic.py (which produced tests/golden/trained_*.npz), by the
train -> infer -> pose-recovered test (tests/test_gpu_trained.py) and by bench.py's trained regime."""
import numpy as np
import torch
import torch.nn.functional as F

from . import synthetic as syn
from .models.model import PointEncoder, PPFEncoder


def real2prob(val, max_val, num_bins):
    """utils/util.py:121-146 (non-circular): a value in [0, max_val] -> weights on its two neighbouring bins.  torch, any device."""
    interval = max_val / (num_bins - 1)
    x = val / interval
    low = torch.clamp(torch.floor(x).long(), max=num_bins - 2)
    w_low = 1.0 - (x - low)
    res = torch.zeros((*val.shape, num_bins), dtype=val.dtype, device=val.device)
    res.scatter_(-1, low[..., None], w_low[..., None])
    res.scatter_(-1, (low + 1)[..., None], (1.0 - w_low)[..., None])
    return res


def targets(pc, normals, idx, center, R, half_extents, cfg):
    """utils/dataset.py:27-60 (generate_target) + :229-246 for an object whose centre / axes / half extents are known, as soft
    bin distributions: (tr [P,2,tr_bins], rot [P,2,rot_bins], aux [P,2], scale [3]) on pc.device (pc in the WORLD frame).
    generate_target's `right_sym` branch (utils/dataset.py:49-50) is not restated: every config of the reference leaves it False
    (config/category/*.yaml), and a config that sets it is refused rather than silently trained without it."""
    if getattr(cfg, "right_sym", False):
        raise NotImplementedError("right_sym categories are not supported by cppf_amd.training.targets (no reference config uses them)")
    dev = pc.device
    c = torch.as_tensor(center, dtype=torch.float32, device=dev)
    Rm = torch.as_tensor(R, dtype=torch.float32, device=dev)
    a = pc[idx[:, 0]] - c
    b = pc[idx[:, 1]] - c
    d = a - b
    u = d / (d.norm(dim=-1, keepdim=True) + 1e-7)
    proj = (a * u).sum(-1)
    dist2o = (a - proj[:, None] * u).norm(dim=-1)
    up = Rm[:, 1]
    right = Rm[:, 2] if cfg.z_right else Rm[:, 0]
    th_up = torch.arccos(torch.clamp(u @ up, -1, 1))
    if cfg.up_sym:
        th_up = torch.minimum(th_up, torch.arccos(torch.clamp(-(u @ up), -1, 1)))
    th_right = torch.arccos(torch.clamp(u @ right, -1, 1))
    n = normals[idx[:, 0]].clone()
    n[(n * u).sum(-1) < 0] *= -1
    aux = torch.stack([(n @ up > 0), (n @ right > 0)], -1).float()
    v0, v1 = cfg.vote_range
    tr = torch.stack([real2prob(torch.clamp(proj + v0, 0, 2 * v0), 2 * v0, cfg.tr_num_bins),
                      real2prob(torch.clamp(dist2o, 0, v1), v1, cfg.tr_num_bins)], 1)
    rot = torch.stack([real2prob(th_up, np.pi, cfg.rot_num_bins), real2prob(th_right, np.pi, cfg.rot_num_bins)], 1)
    scale = torch.as_tensor(np.log(np.asarray(half_extents)) - np.log(np.asarray(cfg.scale_mean)), dtype=torch.float32, device=dev)
    return tr, rot, aux, scale


def loss_fn(preds, tr, rot, aux, scale, cfg):
    """train.py:68-87.  preds [1,P,out_dim]"""
    tb, rb = cfg.tr_num_bins, cfg.rot_num_bins
    kld = lambda logit, tgt: F.kl_div(F.log_softmax(logit, -1), tgt, reduction="batchmean")
    preds_tr = preds[..., :2 * tb].reshape(-1, 2, tb)
    loss = kld(preds_tr[:, 0], tr[:, 0]) + kld(preds_tr[:, 1], tr[:, 1])
    loss = loss + kld(preds[0, :, 2 * tb:2 * tb + rb], rot[:, 0])
    loss = loss + F.binary_cross_entropy_with_logits(preds[0, :, -5], aux[:, 0])
    loss = loss + F.mse_loss(preds[..., -3:], scale[None, None].expand_as(preds[..., -3:]))
    if cfg.regress_right:
        loss = loss + kld(preds[0, :, 2 * tb + rb:2 * tb + 2 * rb], rot[:, 1])
        loss = loss + F.binary_cross_entropy_with_logits(preds[0, :, -4], aux[:, 1])
    return loss


def new_encoders(cfg, dev, seed=0):
    """the two networks of train.py:34-35"""
    torch.manual_seed(seed)
    penc = PointEncoder(k=cfg.knn, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32).to(dev)
    enc = PPFEncoder(cfg.ppffcs, cfg.out_dim).to(dev)
    return penc, enc


def train(category, dev, steps=400, n_points=1024, n_pairs=60000, lr=2e-3, seed=0, log=None, encoders=None, seed0=10000):
    """`steps` steps of train.py's loop body, one freshly generated posed object per step (batch size 1 like the reference,
    train.py:32).  Returns (point_encoder, ppf_encoder, losses)."""
    cfg = syn.CATEGORIES[category]
    penc, enc = encoders or new_encoders(cfg, dev, seed)
    penc.train()
    enc.train()
    opt = torch.optim.Adam([*penc.parameters(), *enc.parameters()], lr=lr)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, steps, eta_min=lr * 0.05)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    n_rng = np.random.default_rng(seed)       # n_points = (lo, hi): a different cloud size every step (voxelised clouds vary)
    losses = []
    for it in range(steps):
        n_it = int(n_rng.integers(n_points[0], n_points[1] + 1)) if isinstance(n_points, (tuple, list)) else int(n_points)
        ob = syn.make_posed_object(category, n_it, seed0 + it)
        pcs = torch.from_numpy(ob["pc"][None]).to(dev)
        nrms = torch.from_numpy(ob["normals"][None]).to(dev)
        idx = torch.randint(0, n_it, (n_pairs, 2), device=dev, generator=gen)              # utils/dataset.py:25
        tr, rot, aux, scale = targets(pcs[0], nrms[0], idx, ob["center"], ob["R"], ob["half_extents"], cfg)
        opt.zero_grad()
        with torch.no_grad():
            dist = torch.cdist(pcs, pcs)                                                     # train.py:61-62
        feat = penc(pcs, nrms, dist)
        preds = enc(pcs, nrms, feat, idxs=idx)
        loss = loss_fn(preds, tr, rot, aux, scale, cfg)
        loss.backward()
        opt.step()
        sched.step()
        if it % 20 == 0 or it == steps - 1:
            losses.append(float(loss.item()))
            if log:
                log(f"{category} step {it:4d} loss {losses[-1]:.4f}")
    penc.eval()
    enc.eval()
    return penc, enc, losses


def infer(penc, enc, ob, dev, n_pairs=100000, seed=0, sphere=None):
    """nocs/inference.py:177-339 on one posed object: kNN + SPRIN features, then cppf_amd.inference.estimate_pose"""
    from .inference import estimate_pose
    from .utils.util import fibonacci_sphere, num_sphere_bins
    cfg = ob["cfg"]
    if sphere is None:
        sphere = np.array(fibonacci_sphere(num_sphere_bins(1.5)))
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n = ob["pc"].shape[0]
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, n, (n_pairs, 2)).astype(np.int64)                                  # :177
    u_tr, u_rot = rng.random((n_pairs, 2), dtype=np.float32), rng.random((n_pairs, 2), dtype=np.float32)
    with torch.no_grad():
        pc, nrm = d(ob["pc"]), d(ob["normals"])
        feat = penc(pc[None], nrm[None])[0]                                                  # :180-181
        return estimate_pose(enc, pc, nrm, feat, d(idx), d(u_tr), d(u_rot), cfg, sphere, pc_host=ob["pc"])


def pose_errors(pose, ob):
    """(centre error in cells, up-axis error in degrees -- modulo sign when the geometry cannot tell, right-axis error in
    degrees modulo sign or None, largest relative scale error)"""
    cfg = ob["cfg"]
    t_err = float(np.max(np.abs(pose["T"] - ob["center"])) / cfg.res)
    up_true = ob["R"][:, 1]
    cu = float(np.clip(pose["up"] @ up_true, -1, 1))
    up_err = float(np.degrees(np.arccos(cu)))
    up_err_mod = float(np.degrees(np.arccos(abs(cu))))
    r_err = None
    if cfg.regress_right:
        right_true = ob["R"][:, 2] if cfg.z_right else ob["R"][:, 0]
        r_err = float(np.degrees(np.arccos(abs(float(np.clip(pose["right"] @ right_true, -1, 1))))))
    s_err = float(np.max(np.abs(pose["scale"] / (2 * ob["half_extents"]) - 1)))
    return dict(t_cells=t_err, up_deg=up_err, up_deg_mod_sign=up_err_mod, right_deg_mod_sign=r_err, scale_rel=s_err)


def save_weights(path, penc, enc, meta=None):
    sd = {"penc." + k: v.detach().cpu().numpy() for k, v in penc.state_dict().items()}
    sd.update({"enc." + k: v.detach().cpu().numpy() for k, v in enc.state_dict().items()})
    np.savez_compressed(path, **sd, **({"meta." + k: np.asarray(v) for k, v in (meta or {}).items()}))


def load_weights(path, cfg, dev):
    z = np.load(path)
    penc, enc = new_encoders(cfg, torch.device("cpu"))
    penc.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("penc.")})
    enc.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("enc.")})
    return penc.to(dev).eval(), enc.to(dev).eval()
