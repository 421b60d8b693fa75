#!/usr/bin/env python3
"""Headline benchmark: point-pairs/s through PPF -> pair MLP -> decode -> centre vote -> arg-max (BASELINE.json metric), one synthetic
object per step per GPU.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one object (default --config c2: N=4096 points, K=128 pairs/point -> P=524 288 pairs): inputs
(points, normals, 40-d point features, int64 pair indices, uniforms, packed weights) already resident in HBM; per step four kernels
(per-point layer-0 projection, fused PPF+MLP+decode of the two centre heads -- the 64 logits nocs/inference.py:185-188 consumes up
to the arg-max -- LDS-tiled vote, reduce+arg-max) replayed from a hipGraph.  Steps ROTATE over 9 distinct objects (~70 MB each: more
than the Infinity Cache holds).  With N GPUs every rank steps through its own objects (weak scaling) and ONE all_gather of the
result records closes the batch inside the timed region; rank 0 then checks EVERY rank's arg-max against the oracle.

  --config c1        BASELINE.json configs[0] (N=1024 K=64, the CPU path): the oracle chain swept over thread counts
  --config c3 | c5   the same chain at configs[2] (N=4096 K=256) / configs[4] (N=8192 K=256, res 2e-3) sizes
  --config c4        configs[3]: 64 mixed-category objects (C2 size), object j on rank j mod N, FULL pose per object through
                     BatchPoseRunner on device-resident objects, one gather of the finished records; strong scaling

This file: the contract line and the timed regions.  bench_secondary.py: the rooflines' kernel timings and every secondary figure;
bench_cpu.py: cpu_baseline and the oracle checks (a worker process); bench_util.py: pipeline sets, the stepper, emit / compact."""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_cpu                                       # noqa: E402
import cppf_amd.synthetic as syn                       # noqa: E402
from bench_dist import bind_rank_cpus, binding_dict, dist_info, self_launch      # noqa: E402
from bench_util import (CONFIGS, METRIC, TRAINED_WEIGHTS, compact, emit, events_per_chain, make_center_set,      # noqa: E402,F401
                        make_stepper, mlp_batch, settle, vote_width, workload_text)
from cppf_amd import sharding                          # noqa: E402
from cppf_amd._torch_util import lane_streams          # noqa: E402
from cppf_amd.config import NOCS_CATEGORIES            # noqa: E402
from cppf_amd.models.model import PPFEncoder           # noqa: E402


def timed_regions(region, args, group, dev):
    """The timed region repeated: at least 5 times and until --min-seconds of regions have run (one region of 20 steps is ~2 ms).
    Every region's time is the MAX over the ranks; the sorted list is returned and the caller reports its MEDIAN.  All ranks run
    the same number of regions: the decision to stop is taken on rank-reduced times."""
    times, total, rec = [], 0.0, None
    while len(times) < max(5, args.regions) or (args.regions == 0 and total < args.min_seconds and len(times) < 100000):
        t, rec = region()
        if group:
            tmax = torch.tensor([t], dtype=torch.float64, device=sharding.collective_device(dev))
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            t = float(tmax.item())
        times.append(t)
        total += t
    return sorted(times), rec


def run_center_config(name, enc, sd, dev, rank, world, args):
    """the timed region of a single-object config + its per-chain latencies; returns a dict of measurements"""
    c = CONFIGS[name]
    n_points, k = (args.n_points or c["n_points"]), (args.pairs_per_point or c["k"])
    n_streams = max(1, args.streams)
    group = torch.distributed.is_initialized()       # world > 1, or ONE rank with CPPF_FORCE_DIST=1 (the RCCL branches on one GPU)
    B = mlp_batch(args, n_points * k)
    n_obj = max(n_streams * B, -(-args.objects // (n_streams * B)) * n_streams * B)   # whole chains of B objects per stream
    objs = make_center_set(enc, dev, n_points, k, c["res"], n_obj, seed0=100 * rank, with_heads=args.all_heads,
                           use_graph=not args.no_graph, vote_workgroups=lambda P_, d_: vote_width(args, P_, d_))
    pipes = [o["pipe"] for o in objs]
    P = objs[0]["idx"].shape[0]
    streams = lane_streams(dev, n_streams)      # (streams on distinct hardware queues, cppf_amd._torch_util.lane_streams)
    steps = args.steps
    res_all = torch.zeros((steps, 16), dtype=torch.uint8, device=dev)   # {i64 arg-max, f32 peak} of every step
    rec_tmpl = torch.zeros((steps, sharding.RECORD), dtype=torch.float64, device=dev)
    rec_tmpl[:, 15] = (rank + world * torch.arange(steps, device=dev)).double()     # step i of rank r = object r + i * world
    res_i64, res_f32 = res_all.view(torch.int64), res_all.view(torch.float32)

    def close_batch():
        """Pack the K per-step results into records and run the single end-of-batch collective."""
        records = rec_tmpl.clone()
        records[:, 12] = res_i64[:, 0]          # arg-max index  (int64 -> f64 in the copy)
        records[:, 13] = res_f32[:, 2]          # peak value     (f32 -> f64 in the copy)
        if world > 1 or group:
            return sharding.gather_records(records, world * steps, rank, world, dev, validate=False)   # the one collective
        return records

    run_steps = make_stepper(dev, pipes, streams, res_all, steps, B, not args.no_vote_batch, args.vote_batch_workgroups)
    if B > 1:
        for p_ in pipes:                    # the single chains too (a remainder of steps mod B, the one-instance latency)
            p_.run()
    run_steps(max(args.warmup, 2 * n_obj))  # every chain is captured and replayed at least once
    calib = run_steps.calibrate()           # (--vote-batch-workgroups -1: the chains' vote width, timed on this workload)
    close_batch()                           # warm-up of the gather too (RCCL communicators are created on first use)
    settle()
    ev0, ev1, own_ms = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), []

    def region():
        """the contract's timed region: EXACTLY `steps` steps + the one gather, barrier + synchronize on both sides.  The two events
        bracket this rank's OWN steps on the device (before the gather, where a rank waits for the slowest): dist.rank_region_ms"""
        if group:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        run_steps(steps)
        ev1.record()
        rec = close_batch()
        if group:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        own_ms.append(ev0.elapsed_time(ev1))
        return t, rec

    regions, allrec = timed_regions(region, args, group, dev)
    elapsed = regions[len(regions) // 2]
    lat = events_per_chain(dev, pipes, max(20, steps))
    return dict(objs=objs, pipes=pipes, P=P, n_points=n_points, k=k, n_obj=n_obj, n_streams=n_streams, elapsed=elapsed,
                mlp_batch=run_steps.batch, vote_batch_workgroups=run_steps.vote_batch_workgroups, vote_batch_calibration=calib,
                regions=regions, allrec=allrec, lat=lat, what=c["what"], own_ms=sorted(own_ms))


def step_argmaxes(m, steps, rank=0, world=1):
    """the arg-max index of every step of the LAST timed region of `rank`'s objects (step i = object i mod n_obj), from the
    gathered records (row = rank + step * world)"""
    return [int(v) for v in m["allrec"][rank:world * steps:world, 12].cpu().tolist()]


def c4_objects(n_objects, n_points, k, seed0=500):
    """BASELINE.json configs[3]: mixed NOCS categories, C2-size clouds; pairs and bin uniforms are drawn on the device"""
    objs = []
    for j in range(n_objects):
        ob = syn.make_object(NOCS_CATEGORIES[j % len(NOCS_CATEGORIES)], n_points, seed0 + j)
        objs.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], cfg=ob["cfg"], n_pairs=n_points * k))
    return objs


def c4_encoders(dev=None):
    """one pair encoder per NOCS category (nocs/inference.py:79-90), random-init under torch.manual_seed(category index)"""
    encs = {}
    for i, c in enumerate(NOCS_CATEGORIES):
        torch.manual_seed(i)
        cfg = syn.CATEGORIES[c]
        encs[c] = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval()
        if dev is not None:
            encs[c] = encs[c].to(dev)
    return encs


def args_no_overlap(args):
    return bool(getattr(args, "no_overlap_batches", False))


def run_c4(dev, rank, world, args, n_objects=64, n_regions=0, host_staged=False, reps_list=None):
    """64 mixed-category objects sharded round-robin over the ranks, full pose per object, ONE gather inside the timed region.
    The objects are uploaded once (BatchPoseRunner.put: SURVEY.md 8d, "inputs already resident on device"); host_staged: every
    batch brings its clouds and features from the host (one pinned copy per object on its lane's stream, BatchPoseRunner._upload):
    the PCIe-inclusive rate -- never `value` of the default command.  reps_list: also time regions of that many batches each with the
    same runner -> m["by_reps"] = {batches per region: sorted region seconds} (bench_secondary.py: the share at matched region lengths)."""
    from cppf_amd.batch import BatchPoseRunner
    n_points, k = (args.n_points or 4096), (args.pairs_per_point or 128)
    runner = BatchPoseRunner(c4_encoders(dev), dev, n_lanes=max(1, args.streams), overlap_batches=not args_no_overlap(args),
                             vote_workgroups=None if args.vote_workgroups < 0 else args.vote_workgroups)
    objects = c4_objects(n_objects, n_points, k)
    if not host_staged:
        objects = runner.put(objects)
    for _ in range(max(6, args.warmup)):     # capture, the first (slow) replays of fresh graphs, form adaptation: ~4 batches
        runner.run(objects, rank, world)
    settle()
    reps = max(1, args.steps // 8)
    group = torch.distributed.is_initialized()
    ev0, own_ms = torch.cuda.Event(enable_timing=True), []
    # recorded by the runner when this rank's chains are enqueued, before the gather
    runner.own_done = torch.cuda.Event(enable_timing=True)

    def region():
        if group:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(reps):
            recs = runner.run(objects, rank, world)
        if group:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        own_ms.append(ev0.elapsed_time(runner.own_done))          # (region start -> the last batch's chains joined, on the device)
        return t, recs

    if n_regions == 1:
        regions, recs = [region()[0]], None
        recs = runner.run(objects, rank, world)
    else:
        regions, recs = timed_regions(region, args, group, dev)
    torch.cuda.synchronize()
    assert recs.shape[0] == n_objects and bool(torch.isfinite(recs[:, :12]).all())
    by_reps = {}
    for r_ in reps_list or ():
        reps = int(r_)                                     # (region() reads `reps`)
        by_reps[reps] = timed_regions(region, args, group, dev)[0]
    return dict(elapsed=regions[len(regions) // 2], regions=regions, reps=max(1, args.steps // 8), n_objects=n_objects, P=n_points * k,
                n_points=n_points, k=k, recs=recs.cpu().numpy(), own_ms=sorted(own_ms), host_staged=host_staged, by_reps=by_reps)


def run_c1(args):
    """BASELINE.json configs[0]: single 1024-point cloud, K=64, bottle, the CPU path.  value = the CPU's pairs/s."""
    torch.manual_seed(0)
    enc = PPFEncoder([84, 32, 32, 16], 141).eval()
    sd = {k_: v.detach().numpy().copy() for k_, v in enc.state_dict().items()}
    o1 = bench_cpu.cpu_object(1024, 64, seed=0)
    P = o1["idx"].shape[0]
    w1 = bench_cpu.run_cpu_worker([{"kind": "baseline", "n_points": 1024, "k": 64, "seed": 0}])[0]
    flat, cb = w1["argmax"], w1["cpu_baseline"]
    out = {"metric": METRIC, "value": cb["value"], "unit": "pairs/s", "n_gpus": 0, "steps": 3, "warmup": 1,
           "ms_per_step": P / cb["value"] * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": f"c1: single object N=1024 K=64 (P={P} pairs), bottle config, res {o1['cfg'].res:g}, grid "
                                  f"{'x'.join(str(int(v)) for v in o1['dims'])}, num_rots 72 adaptive: PPF + MLP + decode + centre vote + "
                                  "arg-max on the HOST cores (BASELINE.json configs[0]: the CPU path, no GPU); a step = one pass over "
                                  "the object at the best thread count", "pairs_per_step": P, "parallelism": "host threads"},
           "cpu_baseline": cb, "argmax_cpu": int(flat)}
    if torch.cuda.is_available():       # the same workload on the GPU, for the ratio
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        args.regions, args.min_seconds = 0, 1.0
        m = run_center_config("c1", enc.to(dev), sd, dev, 0, 1, args)
        out["gpu_same_workload"] = {"ms_per_step": m["elapsed"] / args.steps * 1e3, "pairs_per_s": args.steps * m["P"] / m["elapsed"],
                                    "median_ms_one_instance": m["lat"][len(m["lat"]) // 2],
                                    "argmax_matches_cpu": bool(int(m["allrec"][0, 12].item()) == int(flat))}
    emit(out, args)


def c4_prediction(t64_ms, t8_ms, batches_per_region, gather_ms=0.042):
    """BASELINE.json configs[3] on 8 GPUs from two one-GPU measurements at the SAME number of batches per timed region (the 8-GPU run
    and the one-GPU run hold --steps / 8 batches between their barriers: 2 at the driver's --steps 20): 64 objects on one GPU against
    one GPU's share of 8 objects plus the gather (profiles/r4_rccl_w1.txt: 42 us for the all_gather of the records at world 1;
    latency-bound at any world)"""
    return {"t64_ms_per_object": t64_ms, "t8_ms_per_object": t8_ms, "batches_per_region": batches_per_region, "gather_ms": gather_ms,
            "speedup_8_gpus": 64 * t64_ms / (8 * t8_ms + gather_ms),
            "formula": "64 x T64 / (8 x T8 + gather): one GPU's 64-object batch over one GPU's 8-object share + the one collective "
                       "per batch, both timed with the same number of batches per region"}


def main_c4(args, dev, rank, world, binding):
    m = run_c4(dev, rank, world, args, host_staged=args.host_staged)
    dinfo = dist_info(world, dev, m["own_ms"], binding)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    if rank != 0:
        return
    total_pairs = m["reps"] * m["n_objects"] * m["P"]
    how = ("clouds and features sent from pinned host memory per batch (the PCIe-inclusive rate), records assembled on the device"
           if m["host_staged"] else
           "objects uploaded once (inputs resident in HBM), staged into the chains' buffers by cppf_stage_batch, records assembled on "
           "the device" + ("" if args_no_overlap(args) else ", a batch's chains waiting for their own inputs only (overlap_batches)"))
    out = {"metric": METRIC, "value": total_pairs / m["elapsed"], "unit": "pairs/s", "n_gpus": world,
           "steps": m["reps"] * m["n_objects"], "warmup": args.warmup,
           "ms_per_step": m["elapsed"] / (m["reps"] * m["n_objects"]) * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"c4: batch of {m['n_objects']} objects of mixed NOCS categories (N={m['n_points']} K={m['k']}, "
                                  f"P={m['P']} pairs each; BASELINE.json configs[3]), object j on rank j mod {world}, FULL pose per "
                                  "object (centre chain + back-vote + second pass + orientation vote + sign + scale) through "
                                  f"BatchPoseRunner: {how}, pairs and bin uniforms drawn on the device, one all_gather of the "
                                  "160-byte records closes the batch; a step = one object",
                      "objects": m["n_objects"], "objects_per_gpu": m["n_objects"] / world, "parallelism": f"objects x{world}"},
           "objects_per_s": m["reps"] * m["n_objects"] / m["elapsed"],
           "regions": len(m["regions"]), "region_ms_min_max": [m["regions"][0] * 1e3, m["regions"][-1] * 1e3],
           "dist": dinfo, "cpu_binding": binding_dict(binding)}
    if not args.no_cpu_baseline:      # >= 1 object of every rank against the oracle's full pose (the worker reproduces the device's draws)
        sample = sorted({r + world * s for r in range(world) for s in range(max(1, args.c4_check // world))
                         if r + world * s < m["n_objects"]})
        out.update(bench_cpu.check_c4_records(m, sample, seed=0))
    emit(out, args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default="c2",
                    help="BASELINE.json configuration (default c2 = the headline; see the module docstring)")
    ap.add_argument("--regions", type=int, default=0, help="timed regions of --steps steps each (the median is reported); 0 = at "
                    "least 5 and until --min-seconds of regions have run")
    ap.add_argument("--min-seconds", type=float, default=6.0, help="with --regions 0: keep repeating the timed region until this "
                    "much region time has accumulated")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU worker: no cpu_baseline, no oracle check of the arg-maxes")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary stage timings")
    ap.add_argument("--vote-workgroups", type=int, default=-1, help="width of the vote launches of the timed pipelines: 0 = one "
                    "workgroup per CU, 64..256 = at most that many (cppf.h CPPF_VOTE_WORKGROUPS); -1 = chosen from the pairs per "
                    "tile when more than one instance is in flight (--streams > 1), one per CU otherwise")
    ap.add_argument("--mlp-batch", type=int, default=-1, help="objects whose pair lists share ONE launch of the pair kernel "
                    "(cppf_pair_mlp_decode_batch / CenterBatchPipeline); 1 = one launch per object; -1 = chosen from --streams, "
                    "--steps and the list length")
    ap.add_argument("--no-vote-batch", action="store_true", help="with --mlp-batch > 1: a vote + reduce launch per object instead "
                    "of ONE vote launch and ONE reduce launch for the objects of a chain (cppf_vote_argmax_batch)")
    ap.add_argument("--vote-batch-workgroups", type=int, default=-1, help="workgroups per object of the batched vote: 0 = 256 / "
                    "objects per chain (at least 64); 64..256; -1 = calibrated during the warm-up (64 / 96 / 128 / 192 timed on the "
                    "workload, fastest kept)")
    ap.add_argument("--streams", type=int, default=3, help="instances in flight per GPU: step k runs on HIP stream k mod S "
                    "(1 = strictly one instance at a time)")
    ap.add_argument("--objects", type=int, default=9, help="distinct objects the steps rotate over (rounded up to a multiple of "
                    "--streams); 9 x ~70 MB of buffers exceed the 256 MB Infinity Cache")
    ap.add_argument("--no-graph", action="store_true", help="launch the chain eagerly instead of replaying a hipGraph")
    ap.add_argument("--all-heads", action="store_true", help="first pass decodes all 141 logits of every pair (round 1-2's headline)")
    ap.add_argument("--host-staged", action="store_true", help="--config c4: every batch brings its clouds and features from the host "
                    "(the PCIe-inclusive rate) instead of objects resident on the device")
    ap.add_argument("--no-overlap-batches", action="store_true", help="--config c4: every batch's chains wait for the caller's stream "
                    "(i.e. for the previous batch's join) instead of for their own inputs only (BatchPoseRunner(overlap_batches=...))")
    ap.add_argument("--c4-check", type=int, default=8, help="--config c4: objects (spread over the ranks, at least one each) whose "
                    "records rank 0 checks against the oracle's full pose")
    ap.add_argument("--n-points", type=int, default=0, help="exploration only; overrides the config's N")
    ap.add_argument("--pairs-per-point", type=int, default=0, help="exploration only; overrides the config's K")
    ap.add_argument("--full-line", action="store_true", help="print the full record (~15 KB) instead of the compact line")
    ap.add_argument("--full-record", default=os.path.join(ROOT, "bench_full.json"),
                    help="where rank 0 writes the full record ('' = nowhere)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker is not None:
        return bench_cpu.cpu_worker(args.cpu_worker)
    if args.config == "c1":
        return run_c1(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # no launcher around us: be the launcher
        sys.exit(self_launch(args, sys.argv[1:]))
    rank, world, local = sharding.init_distributed()
    if world != args.gpus:
        sys.stderr.write(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}; run `python bench.py --gpus N` on its own "
                         "or under torch.distributed.run with --nproc-per-node N\n")
        sys.exit(2)
    dev = torch.device("cuda", local)          # (local = LOCAL_RANK, or LOCAL_RANK mod the GPUs present when ranks share devices)
    torch.cuda.set_device(dev)
    binding = bind_rank_cpus(int(os.environ.get("LOCAL_RANK", "0")), world, dev)      # (one rank too: next to its GPU)

    if args.config == "c4":
        main_c4(args, dev, rank, world, binding)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    torch.manual_seed(0)
    enc = PPFEncoder([84, 32, 32, 16], 141).eval()
    sd = {k_: v.detach().numpy().copy() for k_, v in enc.state_dict().items()}
    enc = enc.to(dev)
    m = run_center_config(args.config, enc, sd, dev, rank, world, args)
    P, steps, elapsed = m["P"], args.steps, m["elapsed"]
    secondary = rank == 0 and world == 1 and not args.no_secondary
    import bench_secondary
    ctx = types.SimpleNamespace(args=args, enc=enc, sd=sd, dev=dev, rank=rank, world=world, m=m, secondary=secondary,
                                run_center_config=run_center_config, run_c4=run_c4, workload_text=workload_text,
                                step_argmaxes=step_argmaxes, c4_prediction=c4_prediction)
    roof = bench_secondary.rooflines(ctx)            # every run: the dominant kernels between HIP events -> `roofline`, `roofline_vote`
    extra, pending_checks = bench_secondary.collect(ctx, roof) if secondary else ({}, {})
    dinfo = dist_info(world, dev, m["own_ms"], binding)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()                  # every rank is done with the device before rank 0 starts the CPU legs
    if rank == 0:
        lat = m["lat"]
        out = {
            "metric": METRIC, "value": world * steps * P / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(args.config, m, args), "pairs_per_step_per_gpu": P, "parallelism": f"objects x{world}"},
            "pairs_per_ms_per_gpu": steps * P / elapsed / 1e3,
            "metric_note": ("all 141 logits decoded in the first pass (rounds 1-2's definition of the step)" if args.all_heads else
                            "since round 3 the timed step decodes the 64 centre-bin logits the chain consumes up to the arg-max (like the "
                            "reference's first pass); rounds 1-2 decoded all 141: compare their numbers with `all_heads_first_pass`"),
            # the timed region (exactly `steps` steps + the gather, barrier + synchronize on both sides) was run `regions` times;
            # value / ms_per_step come from the MEDIAN region (max over ranks per region)
            "regions": len(m["regions"]), "region_ms_min_max": [m["regions"][0] * 1e3, m["regions"][-1] * 1e3],
            "dist": dinfo, "cpu_binding": binding_dict(binding),      # (this process's CPUs: the ones next to its GPU, bench_dist.py)
            # hipEvents around the whole chain on one object, one at a time, objects rotating
            "median_ms_one_instance": lat[len(lat) // 2], "one_instance_ms_min_max": [lat[0], lat[-1]], "one_instance_runs": len(lat),
            "vote_workgroups": m["objs"][0]["pipe"].vote_workgroups,      # width of the single chains' vote launches (0 = one per CU)
            "mlp_batch": m["mlp_batch"],                                  # objects per launch of the pair kernel in the timed regions
            "vote_batch": bool(m["mlp_batch"] > 1 and not args.no_vote_batch),
            "vote_batch_workgroups": m["vote_batch_workgroups"], "vote_batch_calibration_ms_per_step": m["vote_batch_calibration"],
        }
        out.update(roof["line"])
        out.update(extra)
        if not args.no_cpu_baseline:
            bench_cpu.attach_checks(out, m, args, rank, world, pending_checks, step_argmaxes)
        emit(out, args)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
