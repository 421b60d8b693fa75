#!/usr/bin/env python3
"""Headline benchmark: point-pairs/s through PPF -> pair MLP -> decode -> centre vote -> arg-max
(BASELINE.json metric), one synthetic object per step per GPU.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one object (default --config c2: N=4096 points, K=128 pairs/point -> P=524 288 pairs;
BASELINE.json configs[1] sizes, fused HIP path): inputs (points, normals, 40-d point features, int64 pair indices, uniforms,
packed weights) already resident in HBM; per step four kernels run (per-point layer-0 projection, fused PPF+MLP+decode of the
two centre heads -- the 64 logits nocs/inference.py:185-188 consumes up to the arg-max; the orientation / scale heads belong to the
second pass on the back-vote's survivors, :236-256, as in the reference and in PosePipeline -- LDS-tiled vote reading the int64
pair list directly, reduce+arg-max), replayed from a hipGraph.  `--all-heads` decodes all 141 logits of every pair in the first
pass instead (round 1-2's headline; reported as a secondary by the default run).  Steps ROTATE over
9 distinct objects (own seeds, own buffers, ~70 MB each: more than the 256 MB Infinity Cache holds), so every step streams its
pair list and uniforms from HBM.  With N GPUs every rank processes its own objects (weak scaling) and ONE all_gather of the K
result records closes the batch inside the timed region.

  --config c1        BASELINE.json configs[0] (N=1024 K=64, "reference CPU voting.py path (no GPU)"): a CPU-only line -- the oracle
                     chain swept over thread counts -- with the GPU's time for the same workload beside it when a GPU is there
  --config c3 | c5   the same chain at BASELINE.json configs[2] (N=4096 K=256) / configs[4] (N=8192 K=256, res 2e-3) sizes
  --config c4        BASELINE.json configs[3]: a batch of 64 mixed-category objects (C2 size) sharded round-robin over the ranks
                     through BatchPoseRunner (full pose per object), one gather at the end; strong scaling
The default run reports c3 / c5 / c4-share timings as secondaries next to the c2 headline.
"""
import argparse
import dataclasses
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cppf_amd.synthetic as syn                      # noqa: E402
from cppf_amd import sharding                         # noqa: E402
from cppf_amd.config import NOCS_CATEGORIES           # noqa: E402
from cppf_amd.inference import CenterPipeline, PoseWorkspace, grid_shape   # noqa: E402
from cppf_amd.models import voting                    # noqa: E402
from cppf_amd.models.model import PPFEncoder         # noqa: E402

NUM_ROTS = 72
FLOP_PER_PAIR = 23968            # 2 x 11 984 MAC of the pair MLP with all 141 outputs
FLOP_PER_PAIR_CENTRE = 21504     # the same with the 64 centre-bin outputs only (final layer 16 x 64 instead of 16 x 141): 2 x 10 752 MAC
FLOP_PER_PAIR_EXECUTED = 13728   # what the all-heads pair kernel issues after hoisting 2x40 layer-0 columns to a per-point table
FLOP_PER_PAIR_CENTRE_EXECUTED = 11168   # ... and the centre-heads kernel (20 of the 108 MFMAs per 16-pair tile fewer)
PEAK_F32_MFMA = 157.3            # TFLOP/s, MI355X_MICROARCH.md
PEAK_HBM = 8000.0                # GB/s
METRIC = "point-pairs/sec (PPF+MLP+vote+argmax), N=4096 K=128; 1/2/4/8 GPU"
PEAK_LDS_ATOMICS = 1.757         # T lane-atomics/s: measured ceiling of ds_add_rtn_u32 on random cells of a 26 k-cell LDS tile, all 256 CUs
                                 # (profiles/r1_atomics_microbench.txt; source profiles/microbench/atomics_bench.hip)
CONFIGS = {                      # BASELINE.json `configs`: single-object chains
    "c1": dict(n_points=1024, k=64, res=None, what="BASELINE.json configs[0] sizes"),
    "c2": dict(n_points=4096, k=128, res=None, what="BASELINE.json configs[1] sizes on the fused path of configs[2]"),
    "c3": dict(n_points=4096, k=256, res=None, what="BASELINE.json configs[2]"),
    "c5": dict(n_points=8192, k=256, res=2e-3, what="BASELINE.json configs[4] per-instance size, fine grid"),
}


def settle():
    """A generation-2 pass of Python's garbage collector over a process that has torch loaded takes 35-70 ms (measured:
    profiles/r2_pose_tail.txt) -- ten to twenty times a whole timed region here -- and when it runs depends on how many
    objects the set-up happened to allocate.  Collect now and move the survivors out of the collector's sight, so that the
    timed loops that follow measure the device path."""
    gc.collect()
    gc.freeze()


def pmc_traffic(kernel, which="pmc_traffic"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r<round>_pmc_traffic.json, newest round; every kernel at
    full width; `which` = "pmc_traffic_timed_width": the default command's timed regions, whose vote is launched narrower)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s.json" % which)), reverse=True):
        try:
            with open(path) as f:
                return json.load(f)[kernel]["hbm_bytes"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def host_threads():
    """hardware threads this process may use.  (In the CPU worker the launching process passes its own count: with OMP_PROC_BIND set
    the OpenMP runtime pins the main thread to ONE core when it loads, and the affinity mask read here would say 1.)"""
    if os.environ.get("CPPF_BENCH_HOST_THREADS"):
        return int(os.environ["CPPF_BENCH_HOST_THREADS"])
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def oracle_center(o, sd, threads=None, all_heads=False):
    """The oracle chain (CPU restatement, `threads` host threads via OpenMP) on one object, ONE pass, timed leg by leg:
    (arg-max, {"mlp": s, "decode": s, "vote_argmax": s})"""
    from oracle import oracle as O
    threads = threads or host_threads()
    O.set_threads(threads)
    cfg, idx = o["cfg"], o["idx"]
    idx32 = idx.astype(np.int32)
    probs = np.ones(o["ob"]["pc"].shape[0], np.float32)
    t0 = time.perf_counter()
    logits = O.pair_mlp(o["ob"]["pc"], o["ob"]["normals"], o["ob"]["feat"], idx, sd, cfg.ppffcs, cfg.out_dim, order=1)
    t1 = time.perf_counter()
    outputs, _ = O.decode_center(logits, o["u_tr"], cfg.tr_num_bins, cfg.vote_range)
    if all_heads:
        O.decode_rot(logits, o["u_rot"], cfg.tr_num_bins, cfg.rot_num_bins)
    t2 = time.perf_counter()
    grid = np.zeros(o["dims"], np.float32)
    O.ppf_voting(o["ob"]["pc"], outputs, probs, idx32, grid, o["corners"][0], cfg.res, NUM_ROTS, True, threads=threads)
    flat, _ = O.grid_argmax(grid)
    t3 = time.perf_counter()
    return flat, {"mlp": t1 - t0, "decode": t2 - t1, "vote_argmax": t3 - t2}


def physical_cores():
    """distinct (package, core) pairs among the CPUs this process may run on (0 when /proc/cpuinfo does not say)"""
    if os.environ.get("CPPF_BENCH_PHYSICAL_CORES"):
        return int(os.environ["CPPF_BENCH_PHYSICAL_CORES"])
    try:
        allowed = os.sched_getaffinity(0)
        cores, cpu, pkg = set(), None, 0
        with open("/proc/cpuinfo") as f:
            for ln in f:
                key, _, val = ln.partition(":")
                key = key.strip()
                if key == "processor":
                    cpu = int(val)
                elif key == "physical id":
                    pkg = int(val)
                elif key == "core id" and cpu in allowed:
                    cores.add((pkg, int(val)))
        return len(cores)
    except (OSError, ValueError):
        return 0


def thread_ladder():
    """8, 16, 32, ... up to every hardware thread this process may use (the ends included), plus the physical-core count"""
    n = host_threads()
    ladder = {t for t in (1, 8, 16, 32, 64, 128, 256, 512) if 8 <= t < n} | {n}
    pc = physical_cores()
    if 8 <= pc <= n:
        ladder.add(pc)
    return sorted(ladder)


CPU_PASSES = 5


def cpu_sweep(o, sd, all_heads=False, budget_s=25.0):
    """The CPU baseline is the CPU's BEST: the oracle chain at every thread count of the ladder (the vote leg keeps one private
    grid per thread and sums them, so more threads are not monotonically better: 256 threads were 3x slower than 8 on round 3's
    box), CPU_PASSES passes each (the budget may cut the last counts short, never below one pass); per count the best pass and the
    [min, median, max] of its passes.  Threads are bound (OMP_PROC_BIND=close OMP_PLACES=cores, set by the worker process this
    runs in: run_cpu_worker).  Returns (arg-max, best entry, all entries)."""
    P = o["idx"].shape[0]
    t_start, entries, flat = time.perf_counter(), [], -1
    oracle_center(o, sd, threads=min(8, host_threads()), all_heads=all_heads)       # page in the library, the tables, the pools
    for th in thread_ladder():
        passes = []
        for _ in range(CPU_PASSES):
            flat, legs = oracle_center(o, sd, threads=th, all_heads=all_heads)
            passes.append((sum(legs.values()), legs))
            if time.perf_counter() - t_start > budget_s:
                break
        passes.sort(key=lambda q: q[0])
        rates = sorted(P / q[0] for q in passes)
        entries.append({"threads": th, "pairs_per_s": P / passes[0][0], "passes": len(passes),
                        "spread_pairs_per_s": [rates[0], rates[len(rates) // 2], rates[-1]],
                        "legs_ms": {k_: v * 1e3 for k_, v in passes[0][1].items()}})
        if time.perf_counter() - t_start > budget_s:
            break
    return flat, max(entries, key=lambda e: e["pairs_per_s"]), entries


def torch_cpu_mlp(o, sd, n_sample=131072, budget_s=6.0):
    """'MLP via torch-CPU with the same weights': the composite of models/model.py:118-137 in torch ops on the
    host, on a bounded prefix of the pair list, at every thread count of the ladder -> best entry, all entries"""
    cfg = o["cfg"]
    enc = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval()
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    pc, nrm, feat = (torch.from_numpy(o["ob"][k]) for k in ("pc", "normals", "feat"))
    idx = torch.from_numpy(o["idx"][:n_sample])
    keep = torch.get_num_threads()
    entries, t_start = [], time.perf_counter()
    with torch.no_grad():
        for th in thread_ladder():
            torch.set_num_threads(th)
            enc._composite(pc, nrm, feat, idx)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                enc._composite(pc, nrm, feat, idx)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            entries.append({"threads": th, "pairs_per_s": idx.shape[0] / best})
            if time.perf_counter() - t_start > budget_s:
                break
    torch.set_num_threads(keep)
    return max(entries, key=lambda e: e["pairs_per_s"]), entries, idx.shape[0]


def cpu_object(n_points, k, seed, res=None, cat="bottle"):
    """a synthetic object with everything the oracle chain needs, no device involved"""
    ob = syn.make_object(cat, n_points, seed=seed)
    cfg = ob["cfg"] if res is None else dataclasses.replace(ob["cfg"], res=res)
    idx = syn.make_pairs(n_points, k, seed=seed)
    u_tr, u_rot = syn.make_uniforms(idx.shape[0], seed=seed)
    corners, dims = grid_shape(ob["pc"], cfg.res)
    return dict(ob=ob, cfg=cfg, idx=idx, u_tr=u_tr, u_rot=u_rot, corners=corners, dims=dims)


def cpu_baseline_block(o, sd, n_points, k, all_heads=False, budget_s=25.0):
    P = o["idx"].shape[0]
    flat, best, entries = cpu_sweep(o, sd, all_heads=all_heads, budget_s=budget_s)
    tbest, tentries, tn = torch_cpu_mlp(o, sd)
    return flat, dict(
        value=best["pairs_per_s"], unit="pairs/s", cores=best["threads"], kind="port",
        best_threads=best["threads"], host_threads_available=host_threads(), physical_cores=physical_cores(),
        legs=best["legs_ms"], spread=best["spread_pairs_per_s"], passes=best["passes"],
        omp_binding={k_: os.environ.get(k_) for k_ in ("OMP_PROC_BIND", "OMP_PLACES")},
        sweep=entries,
        sample=f"full workload (N={n_points}, K={k}, P={P}), best of {CPU_PASSES} passes at each thread count of {thread_ladder()} "
               "(spread = [min, median, max] pairs/s of the passes at the best count; threads bound close to cores): the repo's "
               "C oracle with OpenMP -- AVX2 fmaf-chain MLP + decode + vote (private grid per thread) + arg-max (the reference has "
               "no CPU vote path); value = the best thread count's pairs/s, legs in ms",
        mlp_torch_cpu={"value": tbest["pairs_per_s"], "unit": "pairs/s", "threads": tbest["threads"], "sweep": tentries,
                       "sample": f"PPF + gather + ResLayers + final as torch ops on the host (models/model.py:118-137), "
                                 f"{tn} pairs, same weights; MLP leg only; best thread count of the same ladder"})


def vote_width(args, n_pairs=524288, dims=(26, 76, 26)):
    """The vote launch width of a timed pipeline (cppf.h CPPF_VOTE_WORKGROUPS).  With several instances in flight fewer, longer-lived
    vote workgroups pay fewer 113 KB tiles per instance (zeroed, dumped, read back by the reduce kernel) and leave the rest of the chip
    to the neighbours: about 8 192 pairs per workgroup and tile (profiles/r4_vote_workgroups.txt), i.e. 128 at N=4096 K=128 on the
    bottle's two tiles, 192 on three, the full 256 from a million pairs on.  One instance at a time: one workgroup per CU (0)."""
    if args.vote_workgroups >= 0:
        return args.vote_workgroups
    if args.streams <= 1:
        return 0
    from cppf_amd.inference import grid_class
    T = max(1, grid_class(dims)[0])
    w = -(-int(n_pairs) // 8192) * T
    return 0 if w >= 256 else max(64, w)


def mlp_batch(args, n_pairs=524288):
    """objects per launch of the pair kernel in the timed regions: 8 with several instances in flight up to C2's size, 4 up to a
    million pairs (1, 2, 4 or 8 lists of equal length keep each list on its own XCDs: cppf_pair_mlp_decode_batch), 1 beyond -- a
    launch's fixed cost is under 2 % of it there and the longer chains overlap worse (C5: 0.449 against 0.441 ms per step)"""
    if args.mlp_batch >= 1:
        return min(args.mlp_batch, 8)
    if args.streams <= 1 or args.no_graph or n_pairs > (1 << 20):
        return 1
    # ... and never so long that a timed region of --steps steps holds fewer chains than streams (measured at C2: 20 steps per region
    # 6.04 G pairs/s in chains of 4 against 5.74 in chains of 8 -- two chains and a remainder of 4 --, 50 steps 6.19 in chains of 8)
    cap = max(1, args.steps // max(args.streams, 1))
    B = 8 if n_pairs <= (1 << 19) else 4
    while B > cap:
        B //= 2
    return max(B, 1)


VOTE_BATCH_WIDTHS = (64, 96, 128, 192)


def make_stepper(dev, pipes, streams, res_buf, steps, B, vote_batch=True, vote_batch_wgs=0):
    """-> run(n): n steps, step k = object k mod len(pipes).  B = 1: every step is its own chain on stream k mod S.  B > 1: B
    consecutive objects form ONE chain -- their pair lists in one launch of the pair kernel, then their votes in one vote + one reduce
    launch (CenterBatchPipeline; vote_batch=False: a vote + reduce launch per object) -- on a stream of its own; a remainder of
    n mod B steps runs as single chains, so that EXACTLY n objects are processed.  Every step's 16-byte result is kept (one device
    copy on its stream); the caller's stream waits for all of them.
    vote_batch_wgs: workgroups per object of a chain's vote launch; 0 = 256 / B; -1 = CALIBRATED (run.calibrate(), called by the
    warm-up): the stepper times VOTE_BATCH_WIDTHS on this workload with all streams in flight and keeps the fastest -- which width wins
    depends on how many samples land in the grid (few: the launch is mostly prologue / tile dump, 64 wins; a trained network: the
    launch is deposit arithmetic, 128 wins), and nothing but a run of the workload knows that."""
    from cppf_amd.inference import CenterBatchPipeline
    n_obj, S = len(pipes), len(streams)
    B = max(1, min(B, n_obj // S))      # at least one chain per stream (a captured chain does not run beside itself)
    batches = [CenterBatchPipeline(pipes[i:i + B], vote_batch=vote_batch, vote_workgroups=max(vote_batch_wgs, 0))
               for i in range(0, n_obj - n_obj % B, B)] if B > 1 else []
    rem_chains = {}     # a remainder of r = n mod B steps: ONE shorter chain of the objects whose turn it is (built on first use)

    def rem_chain(first, r):
        key = (first, r)
        if key not in rem_chains:
            rem_chains[key] = CenterBatchPipeline([pipes[(first + q) % n_obj] for q in range(r)], vote_batch=vote_batch,
                                                  vote_workgroups=batches[0].vote_workgroups, own_results=False)
        rem_chains[key].vote_workgroups = batches[0].vote_workgroups
        return rem_chains[key]

    def run(n):
        main = torch.cuda.current_stream(dev)
        for st in streams:
            st.wait_stream(main)
        j = 0
        if batches:
            for c in range(n // B):
                bp = batches[c % len(batches)]
                with torch.cuda.stream(streams[(c % len(batches)) % S]):     # a batch always on the same stream: it never runs beside itself
                    bp.run(check_weights=c < len(batches))
                    lo = (c * B) % steps
                    if lo + B <= steps:          # the chain's B result records in one copy (they sit side by side: bp.results)
                        res_buf[lo:lo + B].copy_(bp.results, non_blocking=True)
                    else:
                        for q, p in enumerate(bp.pipes):
                            res_buf[(c * B + q) % steps].copy_(p.result, non_blocking=True)
            j = (n // B) * B
            if n - j >= 2:            # the remainder as one shorter chain on the next stream in turn
                c = n // B
                with torch.cuda.stream(streams[(c % len(batches)) % S]):
                    rc_ = rem_chain(j % n_obj, n - j)
                    rc_.run(check_weights=False)
                    for q, p in enumerate(rc_.pipes):
                        res_buf[(j + q) % steps].copy_(p.result, non_blocking=True)
                j = n
        for k in range(j, n):
            with torch.cuda.stream(streams[k % S]):
                pipes[k % n_obj].run(check_weights=k < n_obj + j)
                res_buf[k % steps].copy_(pipes[k % n_obj].result, non_blocking=True)
        for st in streams:
            main.wait_stream(st)

    def calibrate(n_steps=None):
        """-> {width: ms per step}; leaves the fastest width set (no-op unless vote_batch_wgs == -1 and the votes are batched)"""
        if not (batches and vote_batch and vote_batch_wgs < 0):
            return None
        n_steps = n_steps or max(2 * len(batches) * B, 24)
        seen = {}
        for w in VOTE_BATCH_WIDTHS:
            for bp in batches:
                bp.vote_workgroups = w
            run(2 * len(batches) * B)                # capture + the slow first replays
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(n_steps)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / n_steps * 1e3)
            seen[w] = sorted(ts)[2]
        best = min(seen, key=seen.get)
        for bp in batches:
            bp.vote_workgroups = best
        run(2 * len(batches) * B)
        run.vote_batch_workgroups = best
        return seen
    run.batch = B
    run.calibrate = calibrate
    run.vote_batch_workgroups = (batches[0].vote_workgroups or 256 // B) if (batches and vote_batch) else None
    return run


def make_center_set(enc, dev, n_points, k, res, n_obj, seed0, with_heads=True, use_graph=True, cat="bottle", vote_workgroups=0):
    """n_obj distinct objects (seed0 + i), each with its own CenterPipeline (static buffers + captured graph), loaded"""
    out = []
    for i in range(n_obj):
        ob = syn.make_object(cat, n_points, seed=seed0 + i)
        cfg = ob["cfg"] if res is None else dataclasses.replace(ob["cfg"], res=res)
        idx = syn.make_pairs(n_points, k, seed=seed0 + i)
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], seed=seed0 + i)
        corners, dims = grid_shape(ob["pc"], cfg.res)
        pipe = CenterPipeline(enc, cfg, n_points, idx.shape[0], dims, dev, NUM_ROTS, adaptive=True, with_heads=with_heads,
                              use_graph=use_graph, vote_workgroups=vote_workgroups(idx.shape[0], dims) if callable(vote_workgroups) else vote_workgroups)
        pipe.load(ob["pc"], ob["normals"], ob["feat"], idx, u_tr, u_rot, corners[0].copy())
        out.append(dict(ob=ob, cfg=cfg, idx=idx, u_tr=u_tr, u_rot=u_rot, corners=corners, dims=dims, pipe=pipe))
    return out


TRAINED_WEIGHTS = os.path.join(ROOT, "tests", "golden", "trained_{}.npz")   # scripts/train_synthetic.py on one MI355X


def make_trained_set(dev, n_points, k, n_obj, seed0, rotate, use_graph=True, cat="bottle", vote_workgroups=0):
    """The headline chain in the regime a DEPLOYED model produces: the networks of tests/golden/trained_<cat>.npz (trained with
    the HIP forward + backward on posed synthetic objects, cppf_amd/training.py), per-point features from the trained SPRIN
    encoder, n_obj held-out posed objects (seeds no training step saw), each with its own CenterPipeline.  A trained network's
    (mu, nu) send every vote circle through the object centre: most samples land in the grid -- the expensive regime of the vote."""
    from cppf_amd import training
    cfg = syn.CATEGORIES[cat]
    penc, enc = training.load_weights(TRAINED_WEIGHTS.format(cat), cfg, dev)
    out = []
    for i in range(n_obj):
        ob = syn.make_posed_object(cat, n_points, seed0 + i, rotate=rotate)
        with torch.no_grad():
            feat = penc(torch.from_numpy(ob["pc"][None]).to(dev), torch.from_numpy(ob["normals"][None]).to(dev))[0]
        idx = syn.make_pairs(n_points, k, seed=seed0 + i)
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], seed=seed0 + i)
        corners, dims = grid_shape(ob["pc"], cfg.res)
        pipe = CenterPipeline(enc, cfg, n_points, idx.shape[0], dims, dev, NUM_ROTS, adaptive=True, with_heads=False, use_graph=use_graph,
                              vote_workgroups=vote_workgroups(idx.shape[0], dims) if callable(vote_workgroups) else vote_workgroups)
        pipe.load(ob["pc"], ob["normals"], feat, idx, u_tr, u_rot, corners[0].copy())
        out.append(dict(ob=ob, cfg=cfg, idx=idx, u_tr=u_tr, u_rot=u_rot, corners=corners, dims=dims, pipe=pipe, feat=feat))
    return out, penc, enc


def events_per_chain(dev, pipes, n):
    """n chains strictly one at a time, each bracketed by its own pair of HIP events on the launch stream (SURVEY.md 8d:
    'hipEvents around the whole chain on one object, median of >= 20 runs'); objects rotate.  Returns the sorted list (ms)."""
    widths = [p.vote_workgroups for p in pipes]
    for p in pipes:                      # one instance alone on the chip: the vote one workgroup per CU (re-captured, warmed)
        p.set_vote_workgroups(0)
    if any(widths):
        for p in pipes:
            p.run(check_weights=False)
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        pipes[i % len(pipes)].run(check_weights=False)
        e1.record()
        torch.cuda.synchronize(dev)
        ts.append(e0.elapsed_time(e1))
    for p, w in zip(pipes, widths):
        p.set_vote_workgroups(w)
    if any(widths):
        for p in pipes:
            p.run(check_weights=False)
        torch.cuda.synchronize(dev)
    return sorted(ts)


def repeated(fn, inner, n=5, per=1.0):
    """a secondary host-clocked timing, REPEATED: n regions of `inner` calls of fn (synchronize on both sides of each region) ->
    (median ms per unit, [min, max]); `per` = units per call.  One unrepeated region is a coin toss on a shared box: round 4
    committed an 8.78 ms full pose where six other runs said 0.25."""
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / inner / per * 1e3)
    ts.sort()
    return ts[len(ts) // 2], [ts[0], ts[-1]]


def event_median(step, inner=10, n=5, warm=1):
    """a secondary timed with HIP events: n brackets of `inner` calls of step() after `warm` untimed ones -> (median ms per call,
    [min, max])"""
    for _ in range(warm):
        step()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2], [ts[0], ts[-1]]


def bracket(fns, n):
    """the closures of `fns` (one per object, cycled) launched n times back to back between two HIP events on the launch stream,
    so that the device queue stays full and the quotient is the kernels' own duration (no host-side launch gaps inside the
    bracket); the smallest of three brackets, because one host hiccup inside a bracket idles the device"""
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fns[0]()
        e0.record()
        for i in range(n):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n
        best = t if best is None else min(best, t)
    return best


def run_center_config(name, enc, sd, dev, rank, world, args):
    """the timed region of a single-object config + its per-chain latencies; returns a dict of measurements"""
    c = CONFIGS[name]
    n_points, k = (args.n_points or c["n_points"]), (args.pairs_per_point or c["k"])
    n_streams = max(1, args.streams)
    group = torch.distributed.is_initialized()       # world > 1, or ONE rank with CPPF_FORCE_DIST=1 (the RCCL branches on one GPU)
    B = mlp_batch(args, n_points * k)
    n_obj = max(n_streams * B, -(-args.objects // (n_streams * B)) * n_streams * B)   # whole chains of B objects, a multiple of the streams
    objs = make_center_set(enc, dev, n_points, k, c["res"], n_obj, seed0=100 * rank, with_heads=args.all_heads,
                           use_graph=not args.no_graph, vote_workgroups=lambda P_, d_: vote_width(args, P_, d_))
    pipes = [o["pipe"] for o in objs]
    P = objs[0]["idx"].shape[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
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

    def region():
        """the contract's timed region: EXACTLY `steps` steps + the one gather, barrier + synchronize on both sides"""
        if group:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(steps)
        rec = close_batch()
        if group:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, rec

    regions, allrec = timed_regions(region, args, group, dev)
    elapsed = regions[len(regions) // 2]
    lat = events_per_chain(dev, pipes, max(20, steps))
    return dict(objs=objs, pipes=pipes, P=P, n_points=n_points, k=k, n_obj=n_obj, n_streams=n_streams, elapsed=elapsed,
                mlp_batch=run_steps.batch, vote_batch_workgroups=run_steps.vote_batch_workgroups, vote_batch_calibration=calib,
                regions=regions, allrec=allrec, lat=lat, what=c["what"])


def step_argmaxes(m, steps):
    """the arg-max index of every step of the LAST timed region of this rank's objects (step i = object i mod n_obj), from the
    gathered records (row = rank + step * world: with one rank, row = step)"""
    return [int(v) for v in m["allrec"][:steps, 12].cpu().tolist()]


def timed_regions(region, args, group, dev):
    """The timed region repeated: at least 5 times and until --min-seconds of regions have run (one region of 20 steps is ~3 ms:
    too short for one host hiccup not to matter and for anything outside the process to see the GPU busy).  Every region's time is
    the MAX over the ranks; the sorted list is returned and the caller reports its MEDIAN (`region_ms_min_max` beside it).
    All ranks run the same number of regions: the decision to stop is taken on rank-reduced times."""
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


def workload_text(name, m, args):
    d = m["objs"][0]["dims"]
    return (f"{name}: single object N={m['n_points']} K={m['k']} (P={m['P']} pairs), bottle config, res {m['objs'][0]['cfg'].res:g}, "
            f"grid {d[0]}x{d[1]}x{d[2]}, num_rots 72 adaptive, fused PPF+MLP(MFMA f32)+decode of " +
            ("all 141 logits" if args.all_heads else "the 64 centre-bin logits (what the chain consumes up to the arg-max; the other heads "
             "belong to the second pass on the survivors)") + " -> LDS-tiled vote -> argmax "
            f"({m['what']}); one object per GPU per step, steps rotate over {m['n_obj']} distinct objects (own buffers: inputs come "
            "from HBM, not the Infinity Cache), " +
            (f"{m['n_streams']} independent objects in flight on {m['n_streams']} HIP streams" +
             (f", each vote launched {m['objs'][0]['pipe'].vote_workgroups} workgroups wide; " if m['objs'][0]['pipe'].vote_workgroups else "; ") if m["n_streams"] > 1
             else "one object at a time; ") +
            (f"the pair lists of {m['mlp_batch']} consecutive objects share one launch of the pair kernel (cppf_pair_mlp_decode_batch), " +
             (f"their votes one vote launch and one reduce launch (cppf_vote_argmax_batch, {m['vote_batch_workgroups']} workgroups per object"
              + (": calibrated during the warm-up)" if args.vote_batch_workgroups < 0 else ")") if m.get("vote_batch_workgroups") else
              "each object then its own vote and reduce launch") + "; chains replayed from hipGraphs" if m.get("mlp_batch", 1) > 1 else
             ("four launches per step replayed from a hipGraph" if not args.no_graph else "eager launches")))


def c4_objects(n_objects, n_points, k, seed0=500):
    """BASELINE.json configs[3]: mixed NOCS categories, C2-size clouds; pairs and bin uniforms are drawn on the device"""
    objs = []
    for j in range(n_objects):
        ob = syn.make_object(NOCS_CATEGORIES[j % len(NOCS_CATEGORIES)], n_points, seed0 + j)
        objs.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], cfg=ob["cfg"], n_pairs=n_points * k))
    return objs


def run_c4(dev, rank, world, args, n_objects=64, n_regions=0):
    """64 mixed-category objects sharded round-robin over the ranks, full pose per object, ONE gather inside the timed region"""
    from cppf_amd.batch import BatchPoseRunner
    n_points, k = (args.n_points or 4096), (args.pairs_per_point or 128)
    encs = {}
    for i, c in enumerate(NOCS_CATEGORIES):
        torch.manual_seed(i)
        cfg = syn.make_object(c, 8, 0)["cfg"]
        encs[c] = PPFEncoder(cfg.ppffcs, cfg.out_dim).eval().to(dev)
    runner = BatchPoseRunner(encs, dev, n_lanes=max(1, args.streams),
                             vote_workgroups=None if args.vote_workgroups < 0 else args.vote_workgroups)
    objects = c4_objects(n_objects, n_points, k)
    for _ in range(max(6, args.warmup)):     # capture, the first (slow) replays of fresh graphs, form adaptation: ~4 batches
        runner.run(objects, rank, world)
    settle()
    reps = max(1, args.steps // 8)
    group = torch.distributed.is_initialized()

    def region():
        if group:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            recs = runner.run(objects, rank, world)
        if group:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, recs

    if n_regions == 1:
        regions, recs = [region()[0]], None
        recs = runner.run(objects, rank, world)
    else:
        regions, recs = timed_regions(region, args, group, dev)
    assert recs.shape[0] == n_objects and bool(torch.isfinite(recs[:, :12]).all())
    return dict(elapsed=regions[len(regions) // 2], regions=regions, reps=reps, n_objects=n_objects, P=n_points * k,
                n_points=n_points, k=k)


def run_cpu_worker(jobs, timeout=900, bind=True):
    """The CPU legs (cpu_baseline sweeps, the oracle's arg-max of every object) in a process of their own: thread binding
    (OMP_PROC_BIND=close OMP_PLACES=cores must be in the environment before the OpenMP runtimes load, and would pin THIS process's
    main thread -- the one that feeds the GPU -- to one core), no distributed environment.  jobs: list of dicts, see cpu_worker."""
    import subprocess
    env = dict(os.environ, CPPF_BENCH_HOST_THREADS=str(host_threads()), CPPF_BENCH_PHYSICAL_CORES=str(physical_cores()))
    if bind:
        env.update(OMP_PROC_BIND="close", OMP_PLACES="cores")
    else:
        env.pop("OMP_PROC_BIND", None)
        env.pop("OMP_PLACES", None)
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "CPPF_FORCE_DIST", "OMP_NUM_THREADS", "TORCHELASTIC_RUN_ID"):
        env.pop(k_, None)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", json.dumps(jobs)], env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("[")]
    if p.returncode != 0 or not lines:
        raise RuntimeError(f"bench.py --cpu-worker failed ({p.returncode}): {p.stderr[-2000:]}")
    return json.loads(lines[-1])


def cpu_worker(spec):
    """`bench.py --cpu-worker '<json>'` (internal): jobs {"kind": "baseline", n_points, k, seed, res, all_heads, budget_s} -> the
    cpu_baseline block + the object's arg-max; {"kind": "argmax", n_points, k, seeds, res, threads} -> the oracle's arg-max of
    every object (the same generator and seeds as the GPU side's make_center_set).  One JSON list on stdout."""
    torch.manual_seed(0)
    enc = PPFEncoder([84, 32, 32, 16], 141).eval()
    sd = {k_: v.detach().numpy().copy() for k_, v in enc.state_dict().items()}
    out = []
    for job in json.loads(spec):
        if job["kind"] == "baseline":
            o = cpu_object(job["n_points"], job["k"], job["seed"], job.get("res"))
            flat, cb = cpu_baseline_block(o, sd, job["n_points"], job["k"], all_heads=job.get("all_heads", False),
                                          budget_s=job.get("budget_s", 25.0))
            out.append({"argmax": int(flat), "cpu_baseline": cb})
        else:
            flats = []
            for seed in job["seeds"]:
                o = cpu_object(job["n_points"], job["k"], seed, job.get("res"))
                flats.append(int(oracle_center(o, sd, threads=job.get("threads") or min(32, host_threads()))[0]))
            out.append({"argmax": flats})
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher's environment: re-run this command as N ranks under
    torch.distributed.run on this node -- rank r on GPU r over RCCL when the node has N GPUs; with fewer GPUs the ranks share them
    (rank r on GPU r mod n) and rendezvous over gloo, so that the whole multi-rank code path runs on a one-GPU box (reported as
    dist.shared_gpu).  The ranks print through this process's stdout: rank 0's JSON line stays the last line."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def run_c1(args):
    """BASELINE.json configs[0]: single 1024-point cloud, K=64, bottle, the CPU path.  value = the CPU's best pairs/s."""
    torch.manual_seed(0)
    enc = PPFEncoder([84, 32, 32, 16], 141).eval()
    sd = {k_: v.detach().numpy().copy() for k_, v in enc.state_dict().items()}
    o1 = cpu_object(1024, 64, seed=0)
    P = o1["idx"].shape[0]
    w1 = run_cpu_worker([{"kind": "baseline", "n_points": 1024, "k": 64, "seed": 0}])[0]
    flat, cb = w1["argmax"], w1["cpu_baseline"]
    out = {"metric": METRIC, "value": cb["value"], "unit": "pairs/s", "n_gpus": 0, "steps": 3, "warmup": 1,
           "ms_per_step": P / cb["value"] * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": f"c1: single object N=1024 K=64 (P={P} pairs), bottle config, res {o1['cfg'].res:g}, grid "
                                  f"{'x'.join(str(int(v)) for v in o1['dims'])}, num_rots 72 adaptive: PPF + MLP + decode + centre vote + "
                                  "arg-max on the HOST cores (BASELINE.json configs[0]: the CPU path, no GPU); a step = one pass over "
                                  "the object at the best thread count", "pairs_per_step": P, "parallelism": "host threads"},
           "cpu_baseline": cb, "argmax_cpu": int(flat)}
    out_args = args
    if torch.cuda.is_available():       # the same workload on the GPU, for the ratio
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        args.regions, args.min_seconds = 0, 1.0
        m = run_center_config("c1", enc.to(dev), sd, dev, 0, 1, args)
        out["gpu_same_workload"] = {"ms_per_step": m["elapsed"] / args.steps * 1e3, "pairs_per_s": args.steps * m["P"] / m["elapsed"],
                                    "median_ms_one_instance": m["lat"][len(m["lat"]) // 2],
                                    "argmax_matches_cpu": bool(int(m["allrec"][0, 12].item()) == int(flat))}
    emit(out, out_args)


def emit(line, args=None):
    """rank 0's ONE JSON line, as the LAST line of stdout: RCCL prints a version banner through C stdio when its first communicator
    is created; with stdout redirected that text sits in libc's buffer until exit and would land BEHIND the JSON line -- so the
    C buffers are flushed first, then the line is written and flushed.  The printed line is the compact one (compact()) unless
    --full-line; the full record is written to --full-record (default bench_full.json beside bench.py) and named in the line."""
    import ctypes
    if args is not None:
        path = args.full_record
        if path:
            try:
                with open(path, "w") as f:
                    json.dump(line, f, indent=1)
                    f.write("\n")
            except OSError:
                path = None
        if not args.full_line:
            line = compact(line)
            line["full_record"] = os.path.basename(path) if path else None
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if torch.distributed.is_initialized():       # (anything the collective library says on its way out comes first, too)
        torch.distributed.destroy_process_group()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()


def dist_info(world, dev):
    """Which collective library carried the gather / barrier / max-over-ranks of this run, and what the group saw (None: no process
    group).  COLLECTIVE: every rank calls it.  ranks_seen = an all-reduced 1 per rank; device_per_rank = every rank's device index;
    shared_gpu = ranks outnumber the node's GPUs (gloo rendezvous, ranks time-share the devices: a code-path run, not a scaling
    measurement)."""
    if not torch.distributed.is_initialized():
        return None
    cd = sharding.collective_device(dev)
    one = torch.ones(1, dtype=torch.int64, device=cd)
    torch.distributed.all_reduce(one)
    mine = torch.tensor([dev.index], dtype=torch.int64, device=cd)
    every = torch.empty(max(world, 1), dtype=torch.int64, device=cd)
    torch.distributed.all_gather_into_tensor(every, mine)
    devs = [int(v) for v in every.cpu().tolist()]
    return {"backend": torch.distributed.get_backend(), "forced_single_rank": world == 1, "ranks_seen": int(one.item()),
            "device_per_rank": devs, "shared_gpu": len(set(devs)) < len(devs)}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary stage timings (counter-collection passes)")
    ap.add_argument("--vote-workgroups", type=int, default=-1, help="width of the vote launches of the timed pipelines: 0 = one "
                    "workgroup per CU, 64..256 = at most that many (cppf.h CPPF_VOTE_WORKGROUPS); -1 = 128 when more than one "
                    "instance is in flight (--streams > 1), one per CU otherwise")
    ap.add_argument("--mlp-batch", type=int, default=-1, help="objects whose pair lists share ONE launch of the pair kernel "
                    "(cppf_pair_mlp_decode_batch / CenterBatchPipeline: the ~9 us a launch spends before its first MFMA are paid once "
                    "per launch); 1 = one launch per object; -1 = 4 when more than one instance is in flight (--streams > 1) and an "
                    "object has at most a million pairs, else 1")
    ap.add_argument("--no-vote-batch", action="store_true", help="with --mlp-batch > 1: a vote + reduce launch per object (round 4's chain) "
                    "instead of ONE vote launch and ONE reduce launch for the objects of a chain (cppf_vote_argmax_batch)")
    ap.add_argument("--vote-batch-workgroups", type=int, default=-1, help="workgroups per object of the batched vote: 0 = 256 / objects per "
                    "chain (at least 32); 32..256; -1 = calibrated during the warm-up (64 / 96 / 128 / 192 timed on the workload, fastest kept)")
    ap.add_argument("--streams", type=int, default=3, help="instances in flight per GPU: step k runs on HIP stream k mod S "
                    "(1 = strictly one instance at a time)")
    ap.add_argument("--objects", type=int, default=9, help="distinct objects the steps rotate over (rounded up to a multiple of "
                    "--streams); 9 x ~70 MB of buffers exceed the 256 MB Infinity Cache")
    ap.add_argument("--no-graph", action="store_true", help="launch the chain eagerly instead of replaying a hipGraph")
    ap.add_argument("--all-heads", action="store_true", help="first pass decodes all 141 logits of every pair (round 1-2's headline) "
                    "instead of the two centre heads")
    ap.add_argument("--n-points", type=int, default=0, help="exploration only; overrides the config's N")
    ap.add_argument("--pairs-per-point", type=int, default=0, help="exploration only; overrides the config's K")
    ap.add_argument("--full-line", action="store_true", help="print the full record (~15 KB) instead of the compact line")
    ap.add_argument("--full-record", default=os.path.join(ROOT, "bench_full.json"), help="where rank 0 writes the full record ('' = nowhere)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker is not None:
        return cpu_worker(args.cpu_worker)
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

    if args.config == "c4":
        m = run_c4(dev, rank, world, args)
        dinfo = dist_info(world, dev)
        if rank == 0:
            total_pairs = m["reps"] * m["n_objects"] * m["P"]
            emit({
                "metric": METRIC, "value": total_pairs / m["elapsed"], "unit": "pairs/s", "n_gpus": world,
                "steps": m["reps"] * m["n_objects"], "warmup": args.warmup,
                "ms_per_step": m["elapsed"] / (m["reps"] * m["n_objects"]) * 1e3, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"c4: batch of {m['n_objects']} objects of mixed NOCS categories (N={m['n_points']} K={m['k']}, "
                                       f"P={m['P']} pairs each; BASELINE.json configs[3]), object j on rank j mod {world}, FULL pose per "
                                       "object (centre chain + back-vote + second pass + orientation vote + sign + scale) through "
                                       "BatchPoseRunner: clouds and features staged from pinned host memory, pairs and bin uniforms drawn "
                                       "on the device, one all_gather of the 160-byte records closes the batch; a step = one object",
                           "objects": m["n_objects"], "objects_per_gpu": m["n_objects"] / world, "parallelism": f"objects x{world}"},
                "objects_per_s": m["reps"] * m["n_objects"] / m["elapsed"],
                "regions": len(m["regions"]), "region_ms_min_max": [m["regions"][0] * 1e3, m["regions"][-1] * 1e3],
                "dist": dinfo}, args)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    torch.manual_seed(0)
    enc = PPFEncoder([84, 32, 32, 16], 141).eval()
    sd = {k_: v.detach().numpy().copy() for k_, v in enc.state_dict().items()}
    enc = enc.to(dev)
    m = run_center_config(args.config, enc, sd, dev, rank, world, args)
    objs, pipes, P, steps = m["objs"], m["pipes"], m["P"], args.steps
    o0, pipe = objs[0], pipes[0]
    cfg, dims = o0["cfg"], o0["dims"]
    elapsed = m["elapsed"]
    d = lambda a: torch.from_numpy(a).to(dev)
    secondary = rank == 0 and world == 1 and not args.no_secondary

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
                    it = dict(encoder=enc, pc=p_.pc, pc_normal=p_.nrm, feat=p_.feat, idxs=p_.idx, u_tr=p_.u_tr, vote_range=o["cfg"].vote_range)
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
        tr_vote_w = [pmc_traffic("v3_vote_batch_kernel", "pmc_traffic_timed_width"), pmc_traffic("v3_reduce_batch_kernel", "pmc_traffic_timed_width")]
        tr_vote_w = [None if v is None else v / m["mlp_batch"] for v in tr_vote_w]
    else:
        tr_vote_w = [pmc_traffic("v3_vote_kernel<true>", "pmc_traffic_timed_width"), pmc_traffic("v3_reduce_kernel", "pmc_traffic_timed_width")]
    alg_bytes = 24 * P + 4 * G_cells          # (mu, nu) 8 B + int64 pair 16 B read per pair, the grid written once
    vote_roofline = {"bound": "lds_atomics", "kernel": "v3_vote_kernel<true> (+ v3_reduce_kernel in the time)", "unit": "T lane-atomics/s",
                     "peak": PEAK_LDS_ATOMICS,
                     "benchmark_inputs": vote_regime(t_vote, landed_samples([o["pipe"].outputs for o in objs])),
                     "traffic": (tr_vote[0] + tr_vote[1]) if all(tr_vote) else None, "algorithmic_bytes": alg_bytes,
                     "traffic_ratio": ((tr_vote[0] + tr_vote[1]) / alg_bytes) if all(tr_vote) else None,
                     # the timed regions launch the vote `vote_workgroups` wide (half the partial tiles at 128): their own PMC passes
                     "traffic_timed_regions": ({"vote_workgroups_per_object": m.get("vote_batch_workgroups") or objs[0]["pipe"].vote_workgroups,
                                                "launch": "v3_vote_batch_kernel + v3_reduce_batch_kernel, per object" if m["mlp_batch"] > 1 and not args.no_vote_batch
                                                else "v3_vote_kernel<true> + v3_reduce_kernel", "bytes": tr_vote_w[0] + tr_vote_w[1],
                                                "ratio": (tr_vote_w[0] + tr_vote_w[1]) / alg_bytes} if all(tr_vote_w) else None),
                     "note": "the vote is bound by LDS read-modify-writes, not by HBM or MFMA (SURVEY.md 8d): achieved = samples that land in "
                             "the grid x 8 trilinear corners (one returning ds_add_u32 each) / time of vote + reduce kernels (HIP events "
                             "around the C-ABI call; the call is both kernels) / the measured ds_add_rtn_u32 ceiling of the chip "
                             "(profiles/r1_atomics_microbench.txt).  Conservative: the reduce kernel's share of the time does no "
                             "atomics (kernel-only durations: profiles/r*_vote_regimes_ktrace.txt).  traffic = HBM bytes of the two "
                             "kernels per call from the committed PMC passes (benchmark inputs) against the algorithmic 24 B/pair + "
                             "the grid: the surplus is the partial tiles -- one per vote workgroup, 256 at full width -- written by the vote and read "
                             "back by the reduce kernel.  Times and `traffic` are of the launch at full width (one workgroup per CU, one "
                             "instance alone on the chip); `traffic_timed_regions`: the narrower launch the timed regions use"}
    if args.config not in ("c2", "c1"):       # the committed PMC passes are of the default (c2) command
        vote_roofline["traffic"] = vote_roofline["traffic_ratio"] = vote_roofline["traffic_timed_regions"] = None

    # secondary: the chain with all 141 logits decoded in the first pass (round 1-2's headline): its own pipelines, same objects
    all_heads_step = None
    if secondary and not args.all_heads:
        ah = make_center_set(enc, dev, m["n_points"], m["k"], CONFIGS[args.config]["res"], m["n_obj"], seed0=100 * rank,
                             with_heads=True, use_graph=not args.no_graph, vote_workgroups=lambda P_, d_: vote_width(args, P_, d_))
        sts = [torch.cuda.Stream(device=dev) for _ in range(m["n_streams"])]

        ah_steps = make_stepper(dev, [o["pipe"] for o in ah], sts, res_sec, steps, m["mlp_batch"], not args.no_vote_batch, args.vote_batch_workgroups)
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
        all_heads_step = {"vote_batch_workgroups": ah_steps.vote_batch_workgroups, "vote_batch_calibration": ah_calib, "ms_per_step": t_ah, "ms_per_step_min_max": [min(regs_ah), max(regs_ah)], "regions": len(regs_ah),
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
            streams = [torch.cuda.Stream(device=dev) for _ in range(m["n_streams"])]

            tr_steps = make_stepper(dev, tpipes, streams, res_sec, steps, m["mlp_batch"], not args.no_vote_batch, args.vote_batch_workgroups)
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
                     "regions": len(reg), "ms_per_step_min_max": [reg[0], reg[-1]], "grid_dims": [list(map(int, o["dims"])) for o in tobjs[:3]],
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
        poses_f = frunner.run(depth, inst)
        same = all((a is None) == (b is None) and (a is None or (a["argmax"] == b["argmax"] and np.array_equal(a["T"], b["T"])
                                                                   and np.array_equal(a["up"], b["up"]) and a["n_surv"] == b["n_surv"]))
                   for a, b in zip(poses_e, poses_f))
        real_frame = {"instances": len(inst), "points_per_instance": [int(p_["n_points"]) for p_ in poses_f], "pairs_per_instance": 100000,
                      "ms_per_instance_incl_preprocessing": t_rf, "ms_per_instance_min_max": mm_rf,
                      "path": "FrameRunner: depth + one label image uploaded per frame, per-instance pre-processing count-driven on the "
                              "device (cppf_frame_cloud_dyn) at the head of captured chains, one read-back per frame",
                      "served_by": dict(frunner.last), "eager_loop_ms_per_instance": t_rf_e, "eager_loop_min_max": mm_rf_e, "poses_equal_eager_loop": bool(same),
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
    other, pending_checks = {}, {}
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
        args.steps = 8
        # one GPU's share of the 64-object batch; the smallest of three batches (a batch is ~2 ms of mostly host work: one
        # scheduler hiccup on the box triples it)
        args.regions, args.min_seconds = 7, 0.0
        m4 = run_c4(dev, rank, world, args, n_objects=8)
        other["c4_one_gpu_share"] = {"workload": "8 mixed-category objects (N=4096 K=128), full pose each, BatchPoseRunner, pairs "
                                                 "drawn on the device, one read-back per batch; median of 7 batches",
                                     "ms_per_object": m4["elapsed"] / (m4["reps"] * 8) * 1e3,
                                     "ms_per_object_min_max": [m4["regions"][0] / (m4["reps"] * 8) * 1e3, m4["regions"][-1] / (m4["reps"] * 8) * 1e3],
                                     "pairs_per_s": m4["reps"] * 8 * m4["P"] / m4["elapsed"]}
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

    dinfo = dist_info(world, dev)
    if rank == 0:
        flop_pair = FLOP_PER_PAIR if args.all_heads else FLOP_PER_PAIR_CENTRE
        flop_exec = FLOP_PER_PAIR_EXECUTED if args.all_heads else FLOP_PER_PAIR_CENTRE_EXECUTED
        argmax_gpu = int(m["allrec"][0, 12].item())
        lat = m["lat"]
        out = {
            "metric": METRIC,
            "value": world * steps * P / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_text(args.config, m, args), "pairs_per_step_per_gpu": P,
                       "parallelism": f"objects x{world}"},
            "pairs_per_ms_per_gpu": steps * P / elapsed / 1e3,
            "metric_note": ("all 141 logits decoded in the first pass (rounds 1-2's definition of the step)" if args.all_heads else
                            "since round 3 the timed step decodes the 64 centre-bin logits the chain consumes up to the arg-max (like the "
                            "reference's first pass); rounds 1-2 decoded all 141: compare their numbers with `all_heads_first_pass`"),
            # the timed region (exactly `steps` steps + the gather, barrier + synchronize on both sides) was run `regions` times;
            # value / ms_per_step come from the MEDIAN region (max over ranks per region)
            "regions": len(m["regions"]), "region_ms_min_max": [m["regions"][0] * 1e3, m["regions"][-1] * 1e3],
            "dist": dinfo,
            # hipEvents around the whole chain on one object, one at a time, objects rotating
            # (one instance alone on the chip: measured with the vote one workgroup per CU, whatever width the timed regions use)
            "median_ms_one_instance": lat[len(lat) // 2],
            "one_instance_ms_min_max": [lat[0], lat[-1]], "one_instance_runs": len(lat),
            # width of the vote launches in the timed regions (cppf.h CPPF_VOTE_WORKGROUPS; 0 = one per CU): with several instances
            # in flight fewer, longer-lived vote workgroups pay fewer 113 KB tiles (zeroed, dumped, reduced) per instance
            "vote_workgroups": m["objs"][0]["pipe"].vote_workgroups,
            # objects per launch of the pair kernel in the timed regions (1 = a launch per object)
            "mlp_batch": m["mlp_batch"],
            # ... and whether the votes of a chain's objects share ONE vote + ONE reduce launch (cppf_vote_argmax_batch)
            "vote_batch": bool(m["mlp_batch"] > 1 and not args.no_vote_batch),
            "vote_batch_workgroups": m["vote_batch_workgroups"], "vote_batch_calibration_ms_per_step": m["vote_batch_calibration"],
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
                                     "pair_encoder_fwd_bwd_adam_step_200k_pairs": mm_step, "train_step_both_encoders_adam_200k_pairs": mm_full}},
            "other_configs": other or None,
            # dominant kernel = the fused pair encoder (one launch between the two events): exact-fp32 MFMA.
            # frac is BOUNDED: the MFMA FLOP the kernel EXECUTES over the fp32-MFMA peak.  (Rounds 1-3 divided the reference's
            # algorithmic FLOP -- 80 of the 188 MFMAs per tile are hoisted to a per-point table and never executed per pair -- by
            # the same time and called that a fraction: it reached 1.01 at C5.  It is kept as `algorithmic_tflops`, a rate.)
            # With --mlp-batch B > 1 the timed regions launch the pair kernel once per B objects (pair_mlp_batch_kernel): the launch the
            # roofline is about is that one -- B lists, B x P pairs, between the two events -- and the single-list launch is kept beside it.
            "roofline": {"bound": "mfma", "kernel": ("pair_mlp_batch_kernel<%s>" if n_lists > 1 else "pair_mlp_kernel<false,true,%s>")
                                                    % ("true" if args.all_heads else "false"),
                         "achieved": flop_exec * P * n_lists / (t_mlp_launch * 1e-3) / 1e12, "peak": PEAK_F32_MFMA,
                         "unit": "TFLOP/s", "frac": flop_exec * P * n_lists / (t_mlp_launch * 1e-3) / 1e12 / PEAK_F32_MFMA,
                         "traffic": (pmc_traffic("pair_mlp_batch_kernel<%s>" % ("true" if args.all_heads else "false")) if n_lists > 1 else None)
                                    or (lambda tr_: None if tr_ is None else tr_ * n_lists)(
                                        pmc_traffic("pair_mlp_kernel<false, true, %s>" % ("true" if args.all_heads else "false"))),
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
                                 "command with --streams 1, whose averages agree"},
            "roofline_vote": dict(vote_roofline, **{k_: (vote_roofline.get("known_answer_inputs") or vote_roofline["benchmark_inputs"])[k_]
                                                    for k_ in ("achieved", "frac")},
                                  regime_of_achieved="known_answer_inputs" if "known_answer_inputs" in vote_roofline else "benchmark_inputs"),
        }
        if world == 1 and not args.no_cpu_baseline:
            # the CPU legs run in a worker process (thread binding, see run_cpu_worker): the baseline sweep on object 0, the oracle's
            # arg-max of EVERY object the timed region stepped through (the batched, XCD-pinned launches included), the other
            # configurations' objects, and BASELINE.json configs[0] (N=1024 K=64, "reference CPU voting.py path (no GPU)")
            n_chk = min(m["n_obj"], steps)
            jobs = [{"kind": "baseline", "n_points": m["n_points"], "k": m["k"], "seed": 100 * rank, "res": CONFIGS[args.config]["res"],
                     "all_heads": args.all_heads},
                    {"kind": "argmax", "n_points": m["n_points"], "k": m["k"], "res": CONFIGS[args.config]["res"],
                     "seeds": list(range(100 * rank + 1, 100 * rank + n_chk))}]
            names = list(pending_checks)
            jobs += [pending_checks[nm][2] for nm in names]
            if args.config == "c2" and not args.no_secondary:
                jobs.append({"kind": "baseline", "n_points": 1024, "k": 64, "seed": 0, "budget_s": 6.0})
            res_w = run_cpu_worker(jobs)
            out["cpu_baseline"] = res_w[0]["cpu_baseline"]
            # ... and the same sweep with the threads left to the scheduler (a shorter budget): the baseline is the better of the two
            free = run_cpu_worker([dict(jobs[0], budget_s=10.0)], bind=False)[0]["cpu_baseline"]
            brief = lambda cb_: {kk: cb_[kk] for kk in ("value", "cores", "spread", "passes", "omp_binding", "legs")}
            if free["value"] > out["cpu_baseline"]["value"]:
                out["cpu_baseline"], free = free, out["cpu_baseline"]
            out["cpu_baseline"]["other_binding"] = brief(free)
            want = [res_w[0]["argmax"]] + res_w[1]["argmax"]
            got = step_argmaxes(m, steps)
            ok_steps = sum(1 for i_, g_ in enumerate(got) if g_ == want[i_ % m["n_obj"]])
            ok_objs = sum(1 for j_ in range(n_chk) if all(g_ == want[j_] for g_ in got[j_::m["n_obj"]]))
            out["argmax_matches_oracle"] = bool(ok_steps == len(got))
            out["argmax_objects_matching_oracle"] = f"{ok_objs}/{n_chk}"
            out["argmax_steps_matching_oracle"] = f"{ok_steps}/{len(got)}"
            for q, nm in enumerate(names):
                got_o, n_obj_o, _ = pending_checks[nm]
                want_o = res_w[2 + q]["argmax"]
                ok_o = sum(1 for i_, g_ in enumerate(got_o) if g_ == want_o[i_ % n_obj_o])
                other[nm]["argmax_matches_oracle"] = bool(ok_o == len(got_o))
                other[nm]["argmax_steps_matching_oracle"] = f"{ok_o}/{len(got_o)}"
            if args.config == "c2" and not args.no_secondary:
                c1 = res_w[-1]["cpu_baseline"]
                out["cpu_baseline"]["c1"] = {kk: c1[kk] for kk in ("value", "unit", "best_threads", "legs", "spread", "sample")}
        emit(out, args)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def compact(out):
    """The line rank 0 prints (< 4 KB): the contract's fields, the two rooflines, the CPU baseline's summary and the secondaries a
    reader needs first.  The full record (every stage, sweep and note: ~15 KB) goes to bench_full.json -- a harness that keeps
    the tail of stdout loses the head of a long line, and with it everything but the key names (BENCH_r04)."""
    pick = lambda d, keys: None if d is None else {k_: d[k_] for k_ in keys if k_ in d}
    line = {k_: out[k_] for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data") if k_ in out}
    cfg = dict(out["config"])
    if len(cfg.get("workload", "")) > 300:
        cfg["workload"] = cfg["workload"][:240] + " ... (full text: bench_full.json)"
    line["config"] = cfg
    for k_ in ("pairs_per_ms_per_gpu", "regions", "region_ms_min_max", "median_ms_one_instance", "vote_workgroups", "mlp_batch", "vote_batch",
               "vote_batch_workgroups", "vote_batch_calibration_ms_per_step", "dist", "argmax_matches_oracle", "argmax_objects_matching_oracle", "argmax_steps_matching_oracle", "objects_per_s"):
        if k_ in out:
            line[k_] = out[k_]
    line["roofline"] = pick(out.get("roofline"), ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "lists_per_launch",
                                                  "executed_flop_per_pair"))
    rv = out.get("roofline_vote")
    if rv is not None:
        line["roofline_vote"] = pick(rv, ("bound", "kernel", "achieved", "peak", "unit", "frac", "regime_of_achieved", "traffic",
                                          "algorithmic_bytes", "traffic_ratio"))
        for reg in ("benchmark_inputs", "known_answer_inputs"):
            if rv.get(reg):
                line["roofline_vote"][reg] = pick(rv[reg], ("stage_ms", "landed_samples", "achieved", "frac"))
        if rv.get("traffic_timed_regions"):
            line["roofline_vote"]["traffic_timed_regions"] = pick(rv["traffic_timed_regions"], ("bytes", "ratio"))
    cb = out.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind", "spread", "passes", "omp_binding", "physical_cores",
                                         "host_threads_available"))
        line["cpu_baseline"]["sample"] = cb["sample"][:100] + " ..."
        line["cpu_baseline"]["sweep_Mpairs_per_s"] = {str(e["threads"]): round(e["pairs_per_s"] / 1e6, 2) for e in cb.get("sweep", [])}
        if cb.get("other_binding"):
            line["cpu_baseline"]["other_binding"] = pick(cb["other_binding"], ("value", "cores", "omp_binding"))
        if cb.get("c1"):
            line["cpu_baseline"]["c1"] = pick(cb["c1"], ("value", "unit", "best_threads"))
    tr = out.get("trained_regime")
    if tr:
        line["trained_regime"] = {"ms_per_step": tr.get("ms_per_step"), "pairs_per_s": tr.get("pairs_per_s")}
        for tag in ("axis_aligned", "random_poses"):
            if tr.get(tag):
                line["trained_regime"][tag] = pick(tr[tag], ("vote_batch_workgroups", "ms_per_step", "ms_per_step_min_max",
                                                             "full_pose_ms_incl_readback"))
    if out.get("all_heads_first_pass"):
        line["all_heads_first_pass"] = pick(out["all_heads_first_pass"], ("ms_per_step", "ms_per_step_min_max", "pairs_per_s"))
    if out.get("stage_ms"):
        line["stage_ms"] = pick(out["stage_ms"], ("ppf_mlp_decode_centre_heads", "vote_reduce_argmax", "vote_reduce_argmax_known_answer_inputs",
                                                  "full_pose_incl_readback"))
    if out.get("other_configs"):
        line["other_configs"] = {nm: pick(v, ("ms_per_step", "ms_per_object", "argmax_steps_matching_oracle")) for nm, v in out["other_configs"].items()}
    if out.get("dropin_flow_reference_defaults"):
        line["dropin_flow_reference_defaults"] = pick(out["dropin_flow_reference_defaults"],
                                                      ("level1_reference_call_sequence_ms", "level2_estimate_pose_eager_ms",
                                                       "level3_batch_runner_captured_ms"))
    if out.get("real_frame"):
        line["real_frame"] = pick(out["real_frame"], ("instances", "ms_per_instance_incl_preprocessing", "ms_per_instance_min_max",
                                                      "eager_loop_ms_per_instance", "poses_equal_eager_loop"))
    def rounded(x):          # 6 significant digits are plenty beside a spread; the contract's own numbers stay as measured
        if isinstance(x, float):
            return float(f"{x:.6g}")
        if isinstance(x, dict):
            return {k_: rounded(v) for k_, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [rounded(v) for v in x]
        return x
    return {k_: (v if k_ in ("value", "ms_per_step") else rounded(v)) for k_, v in line.items()}


if __name__ == "__main__":
    main()
