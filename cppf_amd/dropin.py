"""What the two-line switch gives a user of the reference's own script: the per-instance body of nocs/inference.py:177-339 with
`from models.model import ...` / `from models.voting import ...` pointed at cppf_amd and every `cp.asarray(x)` replaced by a torch
tensor -- the SAME stages in the SAME order with the SAME host round trips (preds_tr.cpu().numpy(), grid_obj.get(),
output_ocs.get(), the numpy mask, candidates through a torch mm, .item() losses) and the reference's launch tuples.  This is
adoption level 1 of INTEGRATION.md; `cppf_amd.inference.estimate_pose` (one stream, one read-back) and the captured
PosePipeline / BatchPoseRunner are levels 2 and 3.  bench.py times all three on the same instances
(`dropin_flow_reference_defaults`); tests/test_gpu_dropin_flow.py checks that the levels agree."""
import numpy as np
import torch

from .models.voting import backvote_kernel, ppf_kernel, rot_voting_kernel

THREADS = 512


def _sample_bins(logits, scale, shift=0.0):
    """softmax -> torch.multinomial(., 1) -> bin value (nocs/inference.py:185-188, 245-256)"""
    k = torch.multinomial(torch.softmax(logits, -1), 1).float()[:, 0]
    return k / (logits.shape[-1] - 1) * scale - shift


def _blocks(n):
    return ((n + THREADS - 1) // THREADS, 1, 1)


def reference_style_instance(point_encoder, ppf_encoder, pc, pc_normal, cfg, sphere_pts, n_pairs=100000, num_rots=72,
                             angle_tol=1.5, adaptive=True, rng=None):
    """pc, pc_normal: numpy f32[N,3] on the host, like the script holds them (:141-142).  Returns dict(T, up, right, R, scale, n_surv)."""
    rng = rng or np.random
    dev = torch.device("cuda")
    tb, rb = cfg.tr_num_bins, cfg.rot_num_bins
    pcs, pc_normals = torch.from_numpy(pc[None]).cuda(), torch.from_numpy(pc_normal[None]).cuda()          # :174-175
    point_idxs = rng.randint(0, pc.shape[0], (n_pairs, 2))                                                 # :177
    with torch.no_grad():
        dist = torch.cdist(pcs, pcs)                                                                       # :180
        sprin_feat = point_encoder(pcs, pc_normals, dist)
        preds = ppf_encoder(pcs, pc_normals, sprin_feat, idxs=point_idxs)                                  # :182
    preds_tr = torch.stack([_sample_bins(preds[0, :, :tb], 2 * cfg.vote_range[0], cfg.vote_range[0]),
                            _sample_bins(preds[0, :, tb:2 * tb], cfg.vote_range[1])], -1)                  # :185-188
    # ---- centre vote: grid on the device, read back, arg-max on the host (:191-211)
    corners = np.stack([pc.min(0), pc.max(0)])                                                             # :194-196
    grid_obj = torch.zeros(tuple(int(v) for v in ((corners[1] - corners[0]) / cfg.res).astype(np.int32) + 1), dtype=torch.float32,
                           device=dev)
    tr_dev = torch.from_numpy(preds_tr.cpu().numpy()).cuda().contiguous()                                  # the script's host bounce (:200)
    idx32 = torch.from_numpy(point_idxs).cuda().to(torch.int32)
    pc_dev, corner_dev = torch.from_numpy(pc).cuda(), torch.from_numpy(corners[0]).cuda()
    ppf_kernel(((pc.shape[0] ** 2 + 511) // 512, 1, 1), (512, 1, 1),
               (pc_dev, tr_dev, torch.ones(pc.shape[0], device=dev), idx32, grid_obj, corner_dev, np.float32(cfg.res), n_pairs,
                num_rots, grid_obj.shape[0], grid_obj.shape[1], grid_obj.shape[2], bool(adaptive)))
    votes = grid_obj.cpu().numpy()                                                                         # grid_obj.get()
    cand = np.array(np.unravel_index(int(np.argmax(votes)), votes.shape))
    T_est = corners[0] + cand * cfg.res
    # ---- back-vote filter, mask + compaction on the host (:216-231)
    output_ocs = torch.zeros((n_pairs, 3), dtype=torch.float32, device=dev)
    backvote_kernel(_blocks(n_pairs), (THREADS, 1, 1),
                    (pc_dev, tr_dev, output_ocs, idx32, corner_dev, np.float32(cfg.res), n_pairs, num_rots, grid_obj.shape[0],
                     grid_obj.shape[1], grid_obj.shape[2], torch.from_numpy(T_est.astype(np.float32)).cuda(), np.float32(3 * cfg.res)))
    mask = np.any(output_ocs.cpu().numpy() != 0, -1)                                                       # output_ocs.get()
    point_idxs = point_idxs[mask]
    n_surv = point_idxs.shape[0]
    if n_surv == 0:
        return dict(T=T_est, up=None, right=None, R=None, scale=None, n_surv=0)
    # ---- second pass on the survivors (:234-256)
    with torch.no_grad():
        preds = ppf_encoder(pcs, pc_normals, sprin_feat, idxs=point_idxs)
        preds_tr2 = torch.stack([_sample_bins(preds[0, :, :tb], 2 * cfg.vote_range[0], cfg.vote_range[0]),
                                 _sample_bins(preds[0, :, tb:2 * tb], cfg.vote_range[1])], -1)
        angles = [_sample_bins(preds[0, :, 2 * tb:2 * tb + rb], np.pi), _sample_bins(preds[0, :, 2 * tb + rb:2 * tb + 2 * rb], np.pi)]
        aux = [preds[0, :, -5], preds[0, :, -4]]
        preds_scale = preds[0, :, -3:]
    # ---- orientation votes (:258-303)
    bce = torch.nn.BCEWithLogitsLoss()
    kept32 = torch.from_numpy(point_idxs).cuda().to(torch.int32)
    tr2_dev = torch.from_numpy(preds_tr2.cpu().numpy()).cuda().contiguous()
    sph = torch.tensor(np.asarray(sphere_pts).T, dtype=torch.float32).cuda()
    dirs = []
    for j in range(2 if cfg.regress_right else 1):
        candidates = torch.zeros((n_surv, num_rots, 3), dtype=torch.float32, device=dev)
        rot_voting_kernel(_blocks(n_surv), (THREADS, 1, 1),
                          (pc_dev, tr2_dev, torch.from_numpy(angles[j].cpu().numpy()).cuda(), candidates, kept32, corner_dev,
                           np.float32(cfg.res), n_surv, num_rots, grid_obj.shape[0], grid_obj.shape[1], grid_obj.shape[2]))
        start = np.arange(0, n_surv * num_rots, num_rots)
        rng.shuffle(start)
        sub = (start[:10000, None] + np.arange(num_rots)[None]).reshape(-1)
        cos = candidates.reshape(-1, 3)[torch.from_numpy(sub).cuda()].mm(sph)
        hits = (cos > np.cos(np.deg2rad(angle_tol))).sum(0).cpu().numpy()                                 # :282-283
        best_dir = np.array(sphere_pts[int(np.argmax(hits))])
        d_ab = pc[point_idxs[:, 0]] - pc[point_idxs[:, 1]]                                                # :287-293
        u_ab = d_ab / (np.linalg.norm(d_ab, axis=-1) + 1e-7)[:, None]
        n_first = pc_normal[point_idxs[:, 0]].copy()
        n_first[(n_first * u_ab).sum(-1) < 0] *= -1
        with torch.no_grad():                                                                             # :295-298
            side = torch.from_numpy(((n_first * best_dir).sum(-1) > 0).astype(np.float32)).cuda()
            up_loss, down_loss = bce(aux[j], side).item(), bce(aux[j], 1. - side).item()
        dirs.append(-best_dir if down_loss < up_loss else best_dir)
    up = dirs[0]
    right = dirs[1] - np.dot(up, dirs[1]) * up if cfg.regress_right else np.array([0.0, -up[2], up[1]])    # :305-312
    right = right / (np.linalg.norm(right) + 1e-9)
    R = np.stack([np.cross(up, right), up, right], -1) if cfg.z_right else np.stack([right, up, np.cross(right, up)], -1)
    scale = np.exp(preds_scale.mean(0).cpu().numpy()) * np.asarray(cfg.scale_mean) * 2                     # :335
    return dict(T=T_est, up=up, right=right, R=R, scale=scale, n_surv=n_surv)
