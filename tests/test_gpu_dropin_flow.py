"""The reference's own per-instance glue (nocs/inference.py:177-303,335), statement by statement and with
its host round trips, running on the drop-in modules -- only the imports and the `cp.asarray(x)` -> torch
tensor substitutions of INTEGRATION.md differ.  `torch.multinomial` is replaced by the arg-max bin so the
flow is deterministic; the fused device pipeline in arg-max mode (u < 0) must then give the same pose."""
import numpy as np
import pytest
import torch

import cppf_amd.synthetic as syn
from cppf_amd.models.model import PPFEncoder                                   # was: from models.model import PPFEncoder
from cppf_amd.models.voting import backvote_kernel, ppf_kernel, rot_voting_kernel   # was: from models.voting import ...
from cppf_amd.utils.util import fibonacci_sphere                               # was: from utils.util import ...

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cat", ["bottle", "mug"])
def test_reference_style_instance_loop(dev, cat):
    from cppf_amd.inference import estimate_pose
    ob = syn.make_object(cat, 1024, 77)
    cfg = ob["cfg"]
    pc, pc_normal = ob["pc"], ob["normals"]
    torch.manual_seed(0)
    ppf_encoder = PPFEncoder(ppffcs=[84, 32, 32, 16], out_dim=2 * cfg.tr_num_bins + 2 * cfg.rot_num_bins + 2 + 3)
    with torch.no_grad():
        ppf_encoder.final.weight *= 6
    ppf_encoder = ppf_encoder.cuda().eval()
    sprin_feat = torch.from_numpy(ob["feat"][None]).cuda()                     # stands in for point_encoder(...)
    angle_tol, num_rots, n_threads = 1.5, 72, 512
    num_samples = int(4 * np.pi / (angle_tol / 180 * np.pi))
    sphere_pts = np.array(fibonacci_sphere(num_samples))
    bcelogits = torch.nn.BCEWithLogitsLoss()
    dev_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()         # was: cp.asarray(a)

    pcs = torch.from_numpy(pc[None]).cuda()
    pc_normals = torch.from_numpy(pc_normal[None]).cuda()
    point_idxs = syn.make_pairs(1024, 32, 77)                                  # np.random.randint(0, N, (P, 2))
    all_idxs = point_idxs.copy()
    with torch.no_grad():
        preds = ppf_encoder(pcs, pc_normals, sprin_feat, idxs=point_idxs)
    preds_tr = torch.softmax(preds[..., :2 * cfg.tr_num_bins].reshape(-1, 2, cfg.tr_num_bins), -1)
    preds_tr = torch.cat([preds_tr[:, 0].argmax(-1, keepdim=True), preds_tr[:, 1].argmax(-1, keepdim=True)], -1).float()[None]
    preds_tr[0, :, 0] = preds_tr[0, :, 0] / (cfg.tr_num_bins - 1) * 2 * cfg.vote_range[0] - cfg.vote_range[0]
    preds_tr[0, :, 1] = preds_tr[0, :, 1] / (cfg.tr_num_bins - 1) * cfg.vote_range[1]

    # vote for center
    block_size = (pc.shape[0] ** 2 + 512 - 1) // 512
    corners = np.stack([np.min(pc, 0), np.max(pc, 0)])
    grid_res = ((corners[1] - corners[0]) / cfg.res).astype(np.int32) + 1
    grid_obj = dev_t(np.zeros(grid_res, dtype=np.float32))
    ppf_kernel(
        (block_size, 1, 1), (512, 1, 1),
        (dev_t(pc).float(), dev_t(preds_tr[0].cpu().numpy()).float(), dev_t(np.ones((pc.shape[0],))).float(),
         dev_t(point_idxs).int(), grid_obj, dev_t(corners[0]), np.float32(cfg.res),
         point_idxs.shape[0], num_rots, grid_obj.shape[0], grid_obj.shape[1], grid_obj.shape[2], True))
    grid_obj = grid_obj.cpu().numpy()                                          # was: grid_obj.get()
    cand = np.array(np.unravel_index([np.argmax(grid_obj, axis=None)], grid_obj.shape)).T[::-1]
    cand_world = corners[0] + cand * cfg.res
    T_est = cand_world[-1]

    # back vote filtering
    block_size = (point_idxs.shape[0] + n_threads - 1) // n_threads
    pred_center = T_est
    output_ocs = torch.zeros((point_idxs.shape[0], 3), dtype=torch.float32, device="cuda")
    backvote_kernel(
        (block_size, 1, 1), (n_threads, 1, 1),
        (dev_t(pc), dev_t(preds_tr[0].cpu().numpy()), output_ocs, dev_t(point_idxs).int(), dev_t(corners[0]),
         np.float32(cfg.res), point_idxs.shape[0], num_rots, grid_obj.shape[0], grid_obj.shape[1], grid_obj.shape[2],
         dev_t(pred_center).float(), np.float32(3 * cfg.res)))
    oc = output_ocs.cpu().numpy()
    mask = np.any(oc != 0, -1)
    point_idxs = point_idxs[mask]

    with torch.no_grad():
        preds = ppf_encoder(pcs, pc_normals, sprin_feat, idxs=point_idxs)
        preds_up = preds[..., 2 * cfg.tr_num_bins:2 * cfg.tr_num_bins + cfg.rot_num_bins]
        preds_right = preds[..., 2 * cfg.tr_num_bins + cfg.rot_num_bins:2 * cfg.tr_num_bins + 2 * cfg.rot_num_bins]
        preds_up_aux, preds_right_aux, preds_scale = preds[..., -5], preds[..., -4], preds[..., -3:]
        preds_up = torch.softmax(preds_up[0], -1).argmax(-1, keepdim=True).float()[None]
        preds_up[0] = preds_up[0] / (cfg.rot_num_bins - 1) * np.pi
        preds_right = torch.softmax(preds_right[0], -1).argmax(-1, keepdim=True).float()[None]
        preds_right[0] = preds_right[0] / (cfg.rot_num_bins - 1) * np.pi

    final_directions = []
    for j, (direction, aux) in enumerate(zip([preds_up, preds_right], [preds_up_aux, preds_right_aux])):
        if j == 1 and not cfg.regress_right:
            continue
        candidates = torch.zeros((point_idxs.shape[0], num_rots, 3), dtype=torch.float32, device="cuda")
        block_size = (point_idxs.shape[0] + 512 - 1) // 512
        rot_voting_kernel(
            (block_size, 1, 1), (512, 1, 1),
            (dev_t(pc), dev_t(preds_tr[0].cpu().numpy()), direction[0, :, 0].contiguous(), candidates,
             dev_t(point_idxs).int(), dev_t(corners[0]).float(), np.float32(cfg.res),
             point_idxs.shape[0], num_rots, grid_obj.shape[0], grid_obj.shape[1], grid_obj.shape[2]))
        sph_cp = torch.tensor(sphere_pts.T, dtype=torch.float32).cuda()
        start = np.arange(0, point_idxs.shape[0] * num_rots, num_rots)         # (no shuffle: first 10 000 survivors)
        sub_sample_idx = (start[:10000, None] + np.arange(num_rots)[None]).reshape(-1)
        cands = candidates.reshape(-1, 3)[torch.LongTensor(sub_sample_idx).cuda()]
        cos = cands.mm(sph_cp)
        counts = torch.sum(cos > np.cos(angle_tol / 180 * np.pi), 0).cpu().numpy()
        best_dir = np.array(sphere_pts[np.argmax(counts)])
        ab = pc[point_idxs[:, 0]] - pc[point_idxs[:, 1]]
        distsq = np.sum(ab ** 2, -1)
        ab_normed = ab / (np.sqrt(distsq) + 1e-7)[..., None]
        pairwise_normals = pc_normal[point_idxs[:, 0]]
        pairwise_normals[np.sum(pairwise_normals * ab_normed, -1) < 0] *= -1
        with torch.no_grad():
            target = torch.from_numpy((np.sum(pairwise_normals * best_dir, -1) > 0).astype(np.float32)).cuda()
            up_loss = bcelogits(aux[0], target).item()
            down_loss = bcelogits(aux[0], 1. - target).item()
        final_directions.append(-best_dir if down_loss < up_loss else best_dir)
    up = final_directions[0]
    pred_scale = np.exp(preds_scale[0].mean(0).cpu().numpy()) * cfg.scale_mean * 2

    # the fused device pipeline on the same inputs, arg-max bins (u < 0)
    neg = torch.full((all_idxs.shape[0], 2), -1.0, device="cuda")
    with torch.no_grad():
        r = estimate_pose(ppf_encoder, pcs[0], pc_normals[0], sprin_feat[0], torch.from_numpy(all_idxs).cuda(), neg, neg,
                          cfg, sphere_pts, pc_host=pc)
    assert mask.sum() > 20
    assert r["n_surv"] == int(mask.sum())
    np.testing.assert_allclose(r["T"], T_est, atol=1e-9)
    np.testing.assert_allclose(r["up"], up, atol=1e-9)
    np.testing.assert_allclose(r["scale"], pred_scale, rtol=2e-5)             # fp32 mean in torch vs fp64 sums here
    if cfg.regress_right:
        np.testing.assert_allclose(r["best_right"] * np.sign(r["best_right"] @ final_directions[1]), final_directions[1],
                                   atol=1e-9)
