"""Everything the benchmark measures besides the contract's timed region: the dominant kernels between HIP events (the two
rooflines of the printed line) and the secondary figures of a single-GPU run (other head sets, trained weights, the other
BASELINE.json configurations, the reference-default batch, a real depth frame, the stages around the path).  bench.py calls
rooflines(ctx) on every run and collect(ctx, roof) when secondaries are on; ctx carries the timed region's objects and bench.py's
own entry points (run_center_config, run_c4, ...)."""
import os
import time

import numpy as np
import torch

import cppf_amd.synthetic as syn
from bench_util import (CONFIGS, FLOP_PER_PAIR, FLOP_PER_PAIR_CENTRE, FLOP_PER_PAIR_CENTRE_EXECUTED, FLOP_PER_PAIR_EXECUTED, NUM_ROTS,
                        PEAK_F32_MFMA, PEAK_LDS_ATOMICS, ROOT, TRAINED_WEIGHTS, bracket, event_median, events_per_chain, make_center_set,
                        make_stepper, make_trained_set, pmc_traffic, repeated, settle, vote_width)
from cppf_amd._torch_util import lane_streams
from cppf_amd.inference import PoseWorkspace
from cppf_amd.models import voting


def rooflines(ctx):
    """per-kernel durations (HIP events on the stream the C ABI launches on), eagerly right after the timed region with the same
    rotating buffers: the dominant kernel alone between two events -> {"line": {"roofline", "roofline_vote"}, + what collect() reuses}"""
    args, enc, dev, m, secondary = ctx.args, ctx.enc, ctx.dev, ctx.m, ctx.secondary
    objs, pipes, P, steps = m["objs"], m["pipes"], m["P"], args.steps
    o0 = objs[0]
    cfg, dims = o0["cfg"], o0["dims"]
    # ---- per-kernel durations (HIP events on the stream the C ABI launches on), eagerly right after the timed region with the
    # same rotating buffers: the dominant kernel alone between two events
    n_ev = max(steps, 5)
    res_sec = torch.zeros((steps, 16), dtype=torch.uint8, device=dev)
    wss = [PoseWorkspace(dev, P, o["dims"], 1) for o in objs]
    settle()

    def mlp_fn(o, u_rot=True):
        p_ = o["pipe"]
        return lambda: enc.forward_decode(p_.pc, p_.nrm, p_.feat, p_.idx, p_.u_tr, o["cfg"].vote_range,
                                          p_.u_rot if u_rot else None, o["cfg"].tr_num_bins, o["cfg"].rot_num_bins)

    def vote_fn(o, ws, outputs):
        p_ = o["pipe"]
        return lambda: voting.vote_argmax(p_.pc, outputs, None, p_.idx, ws.grid, p_.corner, o["cfg"].res, NUM_ROTS, True,
                                          ws.out_idx, ws.out_val, accumulate=False)

    with torch.no_grad():
        t_mlp_all = bracket([mlp_fn(o, True) for o in objs], n_ev) if (args.all_heads or secondary) else None     # ms
        t_mlp_tr = bracket([mlp_fn(o, False) for o in objs], n_ev) if (not args.all_heads or secondary) else None
        t_mlp = t_mlp_all if args.all_heads else t_mlp_tr          # the pair stage of a single chain
        # ... and of the timed regions' chains when the pair lists of several objects share a launch (per LAUNCH of B lists)
        n_lists = m["mlp_batch"]
        t_mlp_launch = t_mlp
        if n_lists > 1:
            from cppf_amd.models.model import forward_decode_batch

            def mlp_batch_fn(group):
                items = []
                for o in group:
                    p_ = o["pipe"]
                    it = dict(encoder=enc, pc=p_.pc, pc_normal=p_.nrm, feat=p_.feat, idxs=p_.idx, u_tr=p_.u_tr,
                              vote_range=o["cfg"].vote_range)
                    if args.all_heads:
                        it["u_rot"] = p_.u_rot
                    items.append(it)
                return lambda: forward_decode_batch(items, group[0]["cfg"].tr_num_bins, group[0]["cfg"].rot_num_bins)
            groups = [objs[i:i + n_lists] for i in range(0, len(objs) - len(objs) % n_lists, n_lists)]
            t_mlp_launch = bracket([mlp_batch_fn(g) for g in groups], max(n_ev // n_lists, 3))
        t_vote = bracket([vote_fn(o, ws, o["pipe"].outputs) for o, ws in zip(objs, wss)], n_ev)

    def landed_samples(outs):
        """samples that land in the grid, averaged over the objects: a sample deposits trilinear weights that sum to 1 (probs are all
        ones), so it is the grid's total mass (fp64 sum of the exact fixed-point grid)"""
        tot = 0.0
        for o, ws, ou in zip(objs, wss, outs):
            vote_fn(o, ws, ou)()
            tot += float(ws.grid.double().sum().item())
        return tot / len(objs)

    def vote_regime(t_ms, landed):
        rate = landed * 8 / (t_ms * 1e-3) / 1e12
        return {"stage_ms": t_ms, "landed_samples": round(landed), "lane_atomics": round(landed) * 8, "achieved": rate,
                "frac": rate / PEAK_LDS_ATOMICS}

    G_cells = int(np.prod(dims))
    tr_vote = [pmc_traffic("v3_vote_kernel<true>"), pmc_traffic("v3_reduce_kernel")]
    # the timed regions' votes: with chains of B objects ONE vote + ONE reduce launch per chain (v3_*_batch_kernel): per-launch bytes / B
    if m["mlp_batch"] > 1 and not args.no_vote_batch:
        tr_vote_w = [pmc_traffic("v3_vote_batch_kernel", "pmc_traffic_timed_width"),
                     pmc_traffic("v3_reduce_batch_kernel", "pmc_traffic_timed_width")]
        tr_vote_w = [None if v is None else v / m["mlp_batch"] for v in tr_vote_w]
    else:
        tr_vote_w = [pmc_traffic("v3_vote_kernel<true>", "pmc_traffic_timed_width"),
                     pmc_traffic("v3_reduce_kernel", "pmc_traffic_timed_width")]
    alg_bytes = 24 * P + 4 * G_cells          # (mu, nu) 8 B + int64 pair 16 B read per pair, the grid written once
    batched = m["mlp_batch"] > 1 and not args.no_vote_batch
    timed = None
    if all(tr_vote_w):          # the timed regions launch the vote narrower (fewer partial tiles per object): their own PMC passes
        timed = {"vote_workgroups_per_object": m.get("vote_batch_workgroups") or objs[0]["pipe"].vote_workgroups,
                 "launch": ("v3_vote_batch_kernel + v3_reduce_batch_kernel, per object" if batched else
                            "v3_vote_kernel<true> + v3_reduce_kernel"),
                 "bytes": tr_vote_w[0] + tr_vote_w[1], "ratio": (tr_vote_w[0] + tr_vote_w[1]) / alg_bytes}
    vote_roofline = {
        "bound": "lds_atomics", "kernel": "v3_vote_kernel<true> (+ v3_reduce_kernel in the time)", "unit": "T lane-atomics/s",
        "peak": PEAK_LDS_ATOMICS,
        "benchmark_inputs": vote_regime(t_vote, landed_samples([o["pipe"].outputs for o in objs])),
        "traffic": (tr_vote[0] + tr_vote[1]) if all(tr_vote) else None, "algorithmic_bytes": alg_bytes,
        "traffic_ratio": ((tr_vote[0] + tr_vote[1]) / alg_bytes) if all(tr_vote) else None,
        "traffic_timed_regions": timed,
        "note": "the vote is bound by LDS read-modify-writes, not by HBM or MFMA (SURVEY.md 8d): achieved = samples that land in the grid "
                "x 8 trilinear corners (one returning ds_add_u32 each) / time of vote + reduce kernels (HIP events around the C-ABI "
                "call; the call is both kernels) / the measured ds_add_rtn_u32 ceiling of the chip "
                "(profiles/r1_atomics_microbench.txt).  Conservative: the reduce kernel's share of the time does no atomics "
                "(kernel-only durations: profiles/r*_vote_regimes_ktrace.txt).  traffic = HBM bytes of the two kernels per call from "
                "the committed PMC passes (benchmark inputs) against the algorithmic 24 B/pair + the grid: the surplus is the partial "
                "tiles -- one per vote workgroup, 256 at full width -- written by the vote and read back by the reduce kernel.  Times "
                "and `traffic` are of the launch at full width (one workgroup per CU, one instance alone on the chip); "
                "`traffic_timed_regions`: the narrower launch the timed regions use"}
    if args.config not in ("c2", "c1"):       # the committed PMC passes are of the default (c2) command
        vote_roofline["traffic"] = vote_roofline["traffic_ratio"] = vote_roofline["traffic_timed_regions"] = None


    def finalize_vote(vr):
        """the printed `roofline_vote`: achieved / frac of the regime a deployed model produces when it was measured"""
        reg = vr.get("known_answer_inputs") or vr["benchmark_inputs"]
        return dict(vr, achieved=reg["achieved"], frac=reg["frac"],
                    regime_of_achieved="known_answer_inputs" if "known_answer_inputs" in vr else "benchmark_inputs")

    flop_pair = FLOP_PER_PAIR if args.all_heads else FLOP_PER_PAIR_CENTRE
    flop_exec = FLOP_PER_PAIR_EXECUTED if args.all_heads else FLOP_PER_PAIR_CENTRE_EXECUTED
    heads = "true" if args.all_heads else "false"
    traffic = (pmc_traffic("pair_mlp_batch_kernel<%s>" % heads) if n_lists > 1 else None)
    if traffic is None:
        single = pmc_traffic("pair_mlp_kernel<false, true, %s>" % heads)
        traffic = None if single is None else single * n_lists
    rate = flop_exec * P * n_lists / (t_mlp_launch * 1e-3) / 1e12
    # dominant kernel = the fused pair encoder (one launch between the two events): exact-fp32 MFMA.  frac is BOUNDED: the MFMA FLOP
    # the kernel EXECUTES over the fp32-MFMA peak (the reference's algorithmic FLOP -- 80 of the 188 MFMAs per tile are hoisted to a
    # per-point table and never executed per pair -- over the same time is kept as `algorithmic_tflops`, a rate).  With --mlp-batch
    # B > 1 the timed regions launch the pair kernel once per B objects: the launch the roofline is about is that one.
    roofline = {"bound": "mfma", "kernel": ("pair_mlp_batch_kernel<%s>" if n_lists > 1 else "pair_mlp_kernel<false,true,%s>") % heads,
                "achieved": rate, "peak": PEAK_F32_MFMA, "unit": "TFLOP/s", "frac": rate / PEAK_F32_MFMA, "traffic": traffic,
                "executed_flop_per_pair": flop_exec, "launch_ms": t_mlp_launch, "lists_per_launch": n_lists,
                "pairs_per_launch": P * n_lists, "single_list_launch_ms": t_mlp,
                "single_list_frac": flop_exec * P / (t_mlp * 1e-3) / 1e12 / PEAK_F32_MFMA,
                "algorithmic_tflops": flop_pair * P * n_lists / (t_mlp_launch * 1e-3) / 1e12, "algorithmic_flop_per_pair": flop_pair,
                "note": f"achieved = MFMA FLOP the kernel issues ({flop_exec} per pair: 2 x 16x16x4 x the tile's MFMAs / 16 pairs) "
                        "x pairs / duration of the pair-encoder stage (point_proj_kernel per list + ONE pair kernel launch for "
                        "`lists_per_launch` lists, HIP events on the launch stream, launches back to back on one stream, inputs "
                        "rotating over the objects; traffic = the launch's PMC bytes, or the single-list launch's x lists); "
                        f"algorithmic_tflops = the reference's layers for the outputs this pass produces ({flop_pair} FLOP per "
                        "pair) over the same time -- larger, because the two 40-wide feature blocks of layer 0 are projected "
                        "once per POINT; fp32 MFMA shares the VALU datapath on gfx950 (32 cycles per MFMA, 4 per VALU "
                        "instruction, no co-issue), so the rest of the pipe's time is the in-register PPF / residual / decode "
                        "VALU work; profiles/r*_kernel_trace_stats_one_stream.txt holds the rocprofv3 durations of the same "
                        "command with --streams 1, whose averages agree"}
    return dict(line={"roofline": roofline, "roofline_vote": finalize_vote(vote_roofline)}, finalize_vote=finalize_vote,
                vote_roofline=vote_roofline, wss=wss, res_sec=res_sec, vote_fn=vote_fn, vote_regime=vote_regime,
                landed_samples=landed_samples, t_mlp_all=t_mlp_all, t_mlp_tr=t_mlp_tr, t_vote=t_vote)


def collect(ctx, R):
    """the secondaries of a single-GPU run -> (fields of the full record, {config: (arg-maxes, n_obj, CPU job)} still to be checked
    against the oracle by the CPU worker)"""
    args, enc, sd, dev, rank, world, m, secondary = ctx.args, ctx.enc, ctx.sd, ctx.dev, ctx.rank, ctx.world, ctx.m, ctx.secondary
    run_center_config, run_c4, workload_text, step_argmaxes = ctx.run_center_config, ctx.run_c4, ctx.workload_text, ctx.step_argmaxes
    objs, pipes, P, steps = m["objs"], m["pipes"], m["P"], args.steps
    o0, pipe = objs[0], pipes[0]
    cfg, dims = o0["cfg"], o0["dims"]
    d = lambda a: torch.from_numpy(a).to(dev)
    wss, res_sec, vote_fn, vote_regime, landed_samples = R["wss"], R["res_sec"], R["vote_fn"], R["vote_regime"], R["landed_samples"]
    vote_roofline, t_mlp_all, t_mlp_tr, t_vote = R["vote_roofline"], R["t_mlp_all"], R["t_mlp_tr"], R["t_vote"]
    # secondary: the chain with all 141 logits decoded in the first pass (round 1-2's headline): its own pipelines, same objects
    all_heads_step = None
    if secondary and not args.all_heads:
        ah = make_center_set(enc, dev, m["n_points"], m["k"], CONFIGS[args.config]["res"], m["n_obj"], seed0=100 * rank,
                             with_heads=True, use_graph=not args.no_graph, vote_workgroups=lambda P_, d_: vote_width(args, P_, d_))
        sts = lane_streams(dev, m["n_streams"])

        ah_steps = make_stepper(dev, [o["pipe"] for o in ah], sts, res_sec, steps, m["mlp_batch"], not args.no_vote_batch,
                                args.vote_batch_workgroups)
        for o in ah:
            o["pipe"].run()
        ah_steps(2 * len(ah))
        ah_calib = ah_steps.calibrate()
        settle()
        regs_ah = []
        for _ in range(9):
            torch.cuda.synchronize()
            ta0 = time.perf_counter()
            ah_steps(steps)
            torch.cuda.synchronize()
            regs_ah.append((time.perf_counter() - ta0) / steps * 1e3)
        t_ah = sorted(regs_ah)[len(regs_ah) // 2]
        lat_ah = events_per_chain(dev, [o["pipe"] for o in ah], 20)
        all_heads_step = {"vote_batch_workgroups": ah_steps.vote_batch_workgroups, "vote_batch_calibration": ah_calib, "ms_per_step": t_ah,
                          "ms_per_step_min_max": [min(regs_ah), max(regs_ah)], "regions": len(regs_ah),
                          "pairs_per_s": P / (t_ah * 1e-3), "median_ms_one_instance": lat_ah[len(lat_ah) // 2]}
        del ah

    # secondary: the vote stage alone, and then the whole step, on known-answer inputs -- every vote circle passes through the
    # object centre, so most samples land in the grid (what a trained network produces), unlike the near-uniform bins of the
    # random-weight MLP of the headline.
    t_vote_ka = t_tail_ka = n_surv_ka = trained = None
    if secondary:
        outs_ka = [d(syn.closed_form_outputs(o["ob"]["pc"], o["ob"]["center"], o["idx"], o["cfg"], quantise=True)) for o in objs]
        t_vote_ka = bracket([vote_fn(o, ws, ka) for o, ws, ka in zip(objs, wss, outs_ka)], 9)
        vote_roofline["known_answer_inputs"] = vote_regime(t_vote_ka, landed_samples(outs_ka))
    # secondary: the WHOLE step with a TRAINED network (round 3 fed the vote closed-form (mu, nu) and threw the random MLP's outputs
    # away): weights trained with the HIP forward + backward (tests/golden/trained_bottle.npz), features from the trained SPRIN
    # encoder, held-out posed objects; same four launches per step, same rotation over objects and streams as the headline
    if secondary and os.path.exists(TRAINED_WEIGHTS.format("bottle")):
        from cppf_amd import training
        from cppf_amd.inference import PosePipeline
        from cppf_amd.utils.util import fibonacci_sphere
        trained = {"weights": "trained", "weights_file": "tests/golden/trained_bottle.npz",
                   "note": "the headline's four launches per step (per-point projection, PPF + MLP + centre decode, vote, reduce + "
                           "arg-max) with networks trained by scripts/train_synthetic.py (HIP forward + backward, 10 000 steps on posed "
                           "synthetic bottles), per-point features from the trained SPRIN encoder, held-out objects; `axis_aligned`: "
                           "objects upright like the headline's (same grid class), `random_poses`: arbitrary rotations (larger "
                           "bounding boxes: more tiles); full_pose = the whole chain incl. back-vote, second pass, orientation vote, "
                           "sign and scale through PosePipeline on one object"}
        sph = np.array(fibonacci_sphere(480))
        for tag, rotate in (("axis_aligned", False), ("random_poses", True)):
            tobjs, penc_t, enc_t = make_trained_set(dev, m["n_points"], m["k"], m["n_obj"], 900100, rotate, use_graph=not args.no_graph,
                                                    vote_workgroups=lambda P_, d_: vote_width(args, P_, d_))
            tpipes = [o["pipe"] for o in tobjs]
            streams = lane_streams(dev, m["n_streams"])

            tr_steps = make_stepper(dev, tpipes, streams, res_sec, steps, m["mlp_batch"], not args.no_vote_batch,
                                    args.vote_batch_workgroups)
            for p_ in tpipes:
                p_.run()
            tr_steps(2 * len(tpipes))
            tr_calib = tr_steps.calibrate()
            settle()
            reg = []
            for _ in range(15):
                torch.cuda.synchronize()
                tt0 = time.perf_counter()
                tr_steps(steps)
                torch.cuda.synchronize()
                reg.append((time.perf_counter() - tt0) / steps * 1e3)
            reg.sort()
            t_tr = reg[len(reg) // 2]
            lat_tr = events_per_chain(dev, tpipes, 20)
            cell_err = []
            for o in tobjs:
                o["pipe"].run(check_weights=False)
                cell = np.array(np.unravel_index(int(o["pipe"].out_idx.item()), o["dims"]))
                cell_err.append(float(np.max(np.abs(cell - (o["ob"]["center"] - o["corners"][0]) / o["cfg"].res))))
            landed = float(np.mean([float(o["pipe"].grid.double().sum().item()) for o in tobjs]))
            entry = {"vote_batch_workgroups": tr_steps.vote_batch_workgroups, "vote_batch_calibration": tr_calib,
                     "ms_per_step": t_tr, "pairs_per_s": P / (t_tr * 1e-3), "median_ms_one_instance": lat_tr[len(lat_tr) // 2],
                     "regions": len(reg), "ms_per_step_min_max": [reg[0], reg[-1]],
                     "grid_dims": [list(map(int, o["dims"])) for o in tobjs[:3]],
                     "argmax_error_cells_max_over_objects": max(cell_err), "landed_samples_per_object": round(landed),
                     "share_of_samples_in_grid": landed / (P * 72.0)}
            # the full pose on the first object of the set
            o = tobjs[0]
            pp = PosePipeline(enc_t, o["cfg"], m["n_points"], P, o["dims"], dev, sph, NUM_ROTS)
            pp.load(o["ob"]["pc"], o["ob"]["normals"], o["feat"], o["idx"], o["u_tr"], o["u_rot"], o["corners"][0].copy())
            for _ in range(4):
                pose_t = pp.run()
            entry["full_pose_ms_incl_readback"], entry["full_pose_ms_min_max"] = repeated(pp.run, 10, 7)
            pose_t = pp.run()
            entry["full_pose_n_surv"] = pose_t["n_surv"]
            entry["full_pose_errors"] = training.pose_errors(pose_t, o["ob"])
            trained[tag] = entry
            del pp, tobjs, tpipes
        trained.update({k_: trained["axis_aligned"][k_] for k_ in ("ms_per_step", "pairs_per_s", "median_ms_one_instance")})

    # secondary: what each level of adoption buys a user of the reference's script: the per-instance body of
    # nocs/inference.py:177-339 at the reference's defaults (P = 100 000 pairs, clouds of whatever size voxel de-duplication left)
    #   level 1  the script's own call sequence and host round trips with the two imports switched (cppf_amd/dropin.py)
    #   level 2  cppf_amd.inference.estimate_pose: same stages fused, one stream, one read-back, eager launches
    #   level 3  BatchPoseRunner: shape-polymorphic captured pipelines, three instances in flight, pairs drawn on the device
    dropin = None
    if secondary and os.path.exists(TRAINED_WEIGHTS.format("mug")):
        from cppf_amd import training
        from cppf_amd.batch import BatchPoseRunner
        from cppf_amd.dropin import reference_style_instance
        from cppf_amd.inference import estimate_pose
        from cppf_amd.utils.util import fibonacci_sphere
        sph = np.array(fibonacci_sphere(480))
        cats = ["bottle", "mug", "laptop"]
        nets = {c: training.load_weights(TRAINED_WEIGHTS.format(c), syn.CATEGORIES[c], dev) for c in cats}
        sizes = (717, 1203, 1890, 960, 1544, 2011, 1333, 1777)
        robjs = [syn.make_posed_object(cats[j % 3], n_j, 910000 + j) for j, n_j in enumerate(sizes)]
        Pd = 100000

        def level1():
            rs = np.random.RandomState(0)
            return [reference_style_instance(nets[o["category"]][0], nets[o["category"]][1], o["pc"], o["normals"], o["cfg"], sph,
                                             n_pairs=Pd, rng=rs) for o in robjs]

        def level2():
            out = []
            for j, o in enumerate(robjs):
                out.append(training.infer(nets[o["category"]][0], nets[o["category"]][1], o, dev, n_pairs=Pd, seed=j, sphere=sph))
            return out
        runner = BatchPoseRunner({c: nets[c][1] for c in cats}, dev, point_encoders={c: nets[c][0] for c in cats})
        batch = [dict(pc=o["pc"], normals=o["normals"], cfg=o["cfg"], n_pairs=Pd) for o in robjs]

        def timed(fn, reps):
            r_ = fn()
            settle()
            med, mm = repeated(fn, reps, 5, per=len(robjs))
            return med, mm, r_
        t1, mm1, r1 = timed(level1, 1)
        t2, mm2, r2 = timed(level2, 2)
        for _ in range(6):      # a freshly captured graph's first replays are slow (the runtime instantiates it lazily), and the
            runner.run(batch)   # pipelines settle on their split / full-first form after the first instances: 4 batches measured
        t3, mm3, r3 = timed(lambda: runner.run(batch), 6)
        err = lambda poses: float(np.median([training.pose_errors(p_, o)["t_cells"] for p_, o in zip(poses, robjs)]))
        dropin = {"workload": f"{len(robjs)} held-out posed objects (bottle / mug / laptop, trained networks), N = {list(sizes)}, "
                              f"{Pd} pairs each (the reference's default), kNN + SPRIN + full pose per instance; ms per instance",
                  "level1_reference_call_sequence_ms": t1, "level2_estimate_pose_eager_ms": t2, "level3_batch_runner_captured_ms": t3,
                  "min_max_ms": {"level1": mm1, "level2": mm2, "level3": mm3}, "timing": "median of 5 repeated regions each",
                  "median_centre_error_cells": {"level1": err(r1), "level2": err(r2)},
                  "level3_records_finite": bool(torch.isfinite(r3[:, :12]).all())}
        del runner

    # secondary: one REAL depth frame (the reference's demo image, tests/golden/demo_0000_depth.png: Kinect noise and holes) through
    # nocs/inference.py:131-142,177-339 -- back-projection, voxel de-duplication, PCA normals, kNN + SPRIN, the whole pose -- per
    # instance, eager launches one at a time, pre-processing included (cppf_amd/frames.py; six rectangular "instances")
    real_frame = None
    depth_png = os.path.join(ROOT, "tests", "golden", "demo_0000_depth.png")
    if secondary and os.path.exists(depth_png) and os.path.exists(TRAINED_WEIGHTS.format("laptop")):
        from cppf_amd import training
        from cppf_amd.frames import frame_poses
        from cppf_amd.utils.util import read_depth_png
        depth = read_depth_png(depth_png)
        rects = [("mug", (262, 356), (124, 206), 90), ("bowl", (184, 246), (288, 366), 90), ("bowl", (194, 250), (370, 442), 90),
                 ("mug", (186, 250), (436, 504), 90), ("can", (112, 184), (376, 408), 60), ("laptop", (118, 322), (92, 302), 260)]
        inst = []
        for cat, (r0, r1), (c0, c1), win in rects:
            msk = np.zeros(depth.shape, bool)
            patch = depth[r0:r1, c0:c1]
            msk[r0:r1, c0:c1] = np.abs(patch.astype(np.int64) - np.median(patch[patch > 0])) <= win
            inst.append((cat, msk))
        src = {"mug": "mug", "laptop": "laptop", "bowl": "bottle", "can": "bottle"}       # (bottle weights stand in for bowl / can)
        nets_f = {c: training.load_weights(TRAINED_WEIGHTS.format(w_), syn.CATEGORIES[w_], dev) for c, w_ in src.items()}
        encs_f, pencs_f = {c: v[1] for c, v in nets_f.items()}, {c: v[0] for c, v in nets_f.items()}
        from cppf_amd.frames import FrameRunner
        for _ in range(2):
            poses_e = frame_poses(depth, inst, encs_f, pencs_f, device=dev)
        settle()
        t_rf_e, mm_rf_e = repeated(lambda: frame_poses(depth, inst, encs_f, pencs_f, device=dev), 1, 5, per=len(inst))
        frunner = FrameRunner(encs_f, pencs_f, dev)
        for _ in range(5):              # first sighting (members' own graphs), capture of the chains, their slow first replays
            poses_f = frunner.run(depth, inst)
        settle()
        t_rf, mm_rf = repeated(lambda: frunner.run(depth, inst), 4, 7, per=len(inst))
        def video(n_frames=6):                      # a video loop: frame k + 1 submitted before frame k is collected
            prev = None
            for _ in range(n_frames):
                cur = frunner.submit(depth, inst)
                if prev is not None:
                    prev.result()
                prev = cur
            prev.result()
        video()
        t_rf_p, mm_rf_p = repeated(lambda: video(6), 2, 7, per=6 * len(inst))
        poses_f = frunner.run(depth, inst)
        same = all((a is None) == (b is None) and (a is None or (a["argmax"] == b["argmax"] and np.array_equal(a["T"], b["T"])
                                                                   and np.array_equal(a["up"], b["up"]) and a["n_surv"] == b["n_surv"]))
                   for a, b in zip(poses_e, poses_f))
        real_frame = {"instances": len(inst), "points_per_instance": [int(p_["n_points"]) for p_ in poses_f], "pairs_per_instance": 100000,
                      "ms_per_instance_incl_preprocessing": t_rf, "ms_per_instance_min_max": mm_rf,
                      "ms_per_instance_pipelined": t_rf_p, "ms_per_instance_pipelined_min_max": mm_rf_p,     # (FrameRunner.submit)
                      "path": "FrameRunner: depth + one label image uploaded per frame, per-instance pre-processing count-driven on the "
                              "device (cppf_frame_cloud_dyn) at the head of captured chains, one read-back per frame",
                      "served_by": dict(frunner.last), "eager_loop_ms_per_instance": t_rf_e, "eager_loop_min_max": mm_rf_e,
                      "poses_equal_eager_loop": bool(same),
                      "n_surv": [int(p_["n_surv"]) for p_ in poses_f]}

    # secondary: centre vote + the whole pose tail on known-answer inputs, where (nearly) every pair survives the back-vote
    if secondary:
        from cppf_amd.inference import _enqueue_tail
        from cppf_amd.utils.util import fibonacci_sphere
        ws_ka = PoseWorkspace(dev, P, dims, 480)
        sph_ka = ws_ka.sphere(np.array(fibonacci_sphere(480)))
        idx32_ka = pipe.idx.to(torch.int32)
        heads_ka = d(syn.closed_form_heads(o0["ob"]["pc"], o0["ob"]["normals"], o0["idx"], cfg))

        def tail_ka():
            voting.vote_argmax(pipe.pc, outs_ka[0], None, pipe.idx, ws_ka.grid, pipe.corner, cfg.res, NUM_ROTS, True, ws_ka.out_idx,
                               ws_ka.out_val, accumulate=False)
            _enqueue_tail(ws_ka, pipe.pc, pipe.nrm, idx32_ka, outs_ka[0], heads_ka, pipe.corner, cfg, dims, NUM_ROTS, 1.5, 10000,
                          *sph_ka)
        with torch.no_grad():
            t_tail_ka = bracket([tail_ka], 5)
        n_surv_ka = int(ws_ka.count.item())
        del ws_ka, heads_ka
    if secondary:
        del outs_ka
    del wss

    # secondary metric: the same object through the FULL pose (centre chain + back-vote + orientation vote + axis
    # sign + scale + one read-back), one hipGraph replay per object
    t_pose, mm_pose, pose = None, None, {"n_surv": None}
    if secondary:
        from cppf_amd.inference import PosePipeline
        from cppf_amd.utils.util import fibonacci_sphere
        pp = PosePipeline(enc, cfg, m["n_points"], P, dims, dev, np.array(fibonacci_sphere(480)), NUM_ROTS)
        pp.load(o0["ob"]["pc"], o0["ob"]["normals"], o0["ob"]["feat"], o0["idx"], o0["u_tr"], o0["u_rot"], o0["corners"][0].copy())
        for _ in range(3):
            pose = pp.run()
        settle()
        t_pose, mm_pose = repeated(pp.run, 10, 7)
        pose = pp.run()
        del pp

    # secondaries: the other BASELINE.json configurations, each through the same code as a --config run of its own
    other, pending_checks, c4_pred = {}, {}, None
    if secondary and args.config == "c2":
        keep = (args.steps, args.objects)
        keep_r = (args.regions, args.min_seconds)
        args.regions, args.min_seconds = 0, 0.5
        for name in ("c3", "c5"):
            args.steps, args.objects = 12, 3 if name == "c5" else 6
            mm = run_center_config(name, enc, sd, dev, rank, world, args)
            entry = {"workload": workload_text(name, mm, args), "ms_per_step": mm["elapsed"] / args.steps * 1e3,
                     "pairs_per_s": args.steps * mm["P"] / mm["elapsed"],
                     "median_ms_one_instance": mm["lat"][len(mm["lat"]) // 2]}
            if not args.no_cpu_baseline:      # every object of the rotation against the oracle (the CPU worker, at the end)
                pending_checks[name] = (step_argmaxes(mm, args.steps), mm["n_obj"],
                                        {"kind": "argmax", "n_points": mm["n_points"], "k": mm["k"], "res": CONFIGS[name]["res"],
                                         "seeds": list(range(100 * rank, 100 * rank + min(mm["n_obj"], args.steps)))})
            other[name] = entry
            del mm
        args.steps, args.objects = keep
        # BASELINE.json configs[3] seen from one GPU: its share of the 64-object batch (8 objects) and the whole batch, objects
        # resident on the device (run_c4), medians of 7 batches; the same share staged from the host per batch (round 5's path)
        # beside it; and what the two predict for the 8-GPU run (bench.c4_prediction)
        args.steps = 8
        args.regions, args.min_seconds = 7, 0.0

        def c4_entry(n_obj, host_staged, what, reps_list=(2, 1, 8)):
            """one runner, regions of 2 batches back to back (what --steps 20 gives the 8-GPU run: the chains of a batch start while
            the previous one's drain), of ONE batch from an idle device to the last record, and of 8 (the pipelined rate)"""
            args.steps = 8 * reps_list[0]
            m4 = run_c4(dev, rank, world, args, n_objects=n_obj, host_staged=host_staged, reps_list=reps_list[1:])
            per = m4["reps"] * n_obj
            # (region start -> the rank's chains joined, HIP events)
            span = m4["own_ms"][len(m4["own_ms"]) // 2] / per if m4["own_ms"] else None
            entry = {"workload": what + f"; median of 7 regions of {reps_list[0]} batches", "batches_per_region": reps_list[0],
                     "ms_per_object": m4["elapsed"] / per * 1e3, "device_span_ms_per_object": span,
                     "ms_per_object_min_max": [m4["regions"][0] / per * 1e3, m4["regions"][-1] / per * 1e3],
                     "pairs_per_s": per * m4["P"] / m4["elapsed"]}
            for r_, regs in m4["by_reps"].items():
                key = "ms_per_object_one_batch_from_idle" if r_ == 1 else f"ms_per_object_{r_}_batches_per_region"
                entry[key] = regs[len(regs) // 2] / (r_ * n_obj) * 1e3
            return entry
        mixed = " mixed-category objects (N=4096 K=128), full pose each, BatchPoseRunner, pairs drawn on the device, "
        other["c4_one_gpu_share"] = c4_entry(8, False, "8" + mixed + "objects resident on the device (put()), records assembled on "
                                             "the device, no read-back in the batch")
        other["c4_one_gpu_share_host_staged"] = c4_entry(8, True, "8" + mixed + "clouds and features sent from pinned host memory "
                                                         "per batch (the PCIe-inclusive rate; round 5: 0.193)")
        other["c4_whole_batch_one_gpu"] = c4_entry(64, False, "64" + mixed + "objects resident on the device", reps_list=(2, 1))
        c4_pred = ctx.c4_prediction(other["c4_whole_batch_one_gpu"]["ms_per_object"], other["c4_one_gpu_share"]["ms_per_object"], 2)
        args.steps = keep[0]
        args.regions, args.min_seconds = keep_r

    # secondary (BASELINE config 4 with the point encoder in front): 8 instances through BatchPoseRunner -- cloud in from the
    # host, pairs and bin uniforms drawn on the device, kNN + SPRIN + full pose per instance, one read-back for the batch
    t_batch = mm_batch = None
    if secondary:
        from cppf_amd.batch import BatchPoseRunner
        from cppf_amd.models.model import PointEncoder
        torch.manual_seed(3)
        penc_b = PointEncoder(k=60, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32).eval().to(dev)
        runner = BatchPoseRunner({cfg.category: enc}, dev, point_encoders={cfg.category: penc_b})
        batch = []
        for j in range(8):
            obj_j = syn.make_object("bottle", m["n_points"], seed=100 + j)
            batch.append(dict(pc=obj_j["pc"], normals=obj_j["normals"], cfg=obj_j["cfg"], n_pairs=P))
        for _ in range(6):
            runner.run(batch)
        settle()
        t_batch, mm_batch = repeated(lambda: runner.run(batch), 2, 5, per=8)
        del runner

    # secondary: the step before the path -- kNN(60) + SPRIN point encoder producing `feat`
    # (nocs/inference.py:180-181), random-init weights of the reference's configuration (train.py:34)
    t_penc = mm_penc = None
    if secondary:
        from cppf_amd.models.model import PointEncoder
        torch.manual_seed(1)
        penc = PointEncoder(k=60, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32).eval().to(dev)
        settle()
        with torch.no_grad():
            t_penc, mm_penc = event_median(lambda: penc(pipe.pc[None], pipe.nrm[None]), inner=5)

    # secondary: one training-size forward + backward of the pair encoder (train.py:66,91:
    # 200 000 pairs, dL/dlogits given), HIP forward + HIP backward through the autograd.Function
    t_train = t_step = t_full = mm_train = mm_step = mm_full = None
    if secondary:
        Pt = 200000
        n_pts = m["n_points"]
        idx_t = d(syn.make_pairs(n_pts, (Pt + n_pts - 1) // n_pts, 7)[:Pt])
        Rt = torch.randn((Pt, cfg.out_dim), device=dev)
        pc, nrm = pipe.pc, pipe.nrm
        feat_t = pipe.feat.clone().requires_grad_(True)
        enc.train()
        settle()
        def step_fwd_bwd():
            enc.zero_grad()
            feat_t.grad = None
            enc.forward_with_idx(pc, nrm, feat_t, idx_t).backward(Rt)
        t_train, mm_train = event_median(step_fwd_bwd)
        # the same with the weights changing every step (train.py:89-92: zero_grad, backward, Adam step): the weight
        # image is re-packed on the device each step, nothing synchronises with the host
        import copy
        enc_t = copy.deepcopy(enc)
        opt = torch.optim.Adam(enc_t.parameters(), lr=1e-4)
        settle()

        def step_adam():
            opt.zero_grad()
            feat_t.grad = None
            enc_t.forward_with_idx(pc, nrm, feat_t, idx_t).backward(Rt)
            opt.step()
        t_step, mm_step = event_median(step_adam)
        # the whole of train.py:58-92 for one sample: cdist, point encoder, pair encoder, backward through both, Adam
        from cppf_amd.models.model import PointEncoder
        torch.manual_seed(2)
        penc_t = PointEncoder(k=60, spfcs=[32, 64, 32, 32], num_layers=1, out_dim=32).to(dev).train()
        opt2 = torch.optim.Adam([*penc_t.parameters(), *enc_t.parameters()], lr=1e-4)
        pcs_b, nrm_b = pc[None], nrm[None]
        settle()

        def step_full():
            opt2.zero_grad()
            with torch.no_grad():
                dist_b = torch.cdist(pcs_b, pcs_b)
            f_b = penc_t(pcs_b, nrm_b, dist_b)
            enc_t(pcs_b, nrm_b, f_b, idxs=idx_t)[0].backward(Rt)
            opt2.step()
        t_full, mm_full = event_median(step_full)
        enc.eval()

    extra = {
        "trained_regime": trained,
        "all_heads_first_pass": all_heads_step,
        "dropin_flow_reference_defaults": dropin,
        "real_frame": real_frame,
        "stage_ms": {"ppf_mlp_decode_all_heads": t_mlp_all, "ppf_mlp_decode_centre_heads": t_mlp_tr, "vote_reduce_argmax": t_vote,
                     "vote_reduce_argmax_known_answer_inputs": t_vote_ka,
                     "vote_plus_pose_tail_known_answer_inputs": t_tail_ka, "pose_tail_known_answer_n_surv": n_surv_ka,
                     "full_pose_incl_readback": t_pose, "full_pose_incl_readback_min_max": mm_pose, "full_pose_n_surv": pose["n_surv"],
                     "batch_of_8_instances_knn_sprin_full_pose_per_instance": t_batch,
                     "batch_of_8_instances_min_max": mm_batch,
                     "point_encoder_knn60_sprin": t_penc,
                     "pair_encoder_fwd_bwd_200k_pairs": t_train,
                     "pair_encoder_fwd_bwd_adam_step_200k_pairs": t_step,
                     "train_step_both_encoders_adam_200k_pairs": t_full,
                     "min_max": {"point_encoder_knn60_sprin": mm_penc, "pair_encoder_fwd_bwd_200k_pairs": mm_train,
                                 "pair_encoder_fwd_bwd_adam_step_200k_pairs": mm_step,
                                 "train_step_both_encoders_adam_200k_pairs": mm_full}},
        "other_configs": other or None,
        "c4_strong_scaling_predicted": c4_pred,
        "roofline_vote": R["finalize_vote"](vote_roofline),
    }
    return extra, pending_checks
