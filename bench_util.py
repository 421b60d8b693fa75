"""Shared pieces of the benchmark (bench.py, bench_secondary.py, bench_cpu.py): constants, timing helpers, the sets of loaded
pipelines the timed regions step through, the stepper that runs them as chains, and the printed line (emit / compact)."""
import dataclasses
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import cppf_amd.synthetic as syn                      # noqa: E402
from cppf_amd.inference import CenterPipeline, grid_shape   # noqa: E402

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
                with torch.cuda.stream(streams[(c % len(batches)) % S]):     # always the same stream: a batch never runs beside itself
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
        # Timings long enough to tell 2 % apart (64 against 96 at C2: 0.0877 against 0.0894 ms per step): 7 x 72 steps (~6 ms each) per
        # width.  Round 6 found the 5 x 24 steps of before mis-measuring the FIRST width by 2-3 % in one process of five on some boxes
        # (a 2 ms timing right behind the captures); the run then kept 96 workgroups and the headline read 5.8 instead of 6.0 G pairs/s.
        n_steps = n_steps or max(6 * len(batches) * B, 72)
        seen = {}
        for w in VOTE_BATCH_WIDTHS:
            for bp in batches:
                bp.vote_workgroups = w
            run(4 * len(batches) * B)                # capture + the slow first replays
            ts = []
            for _ in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(n_steps)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / n_steps * 1e3)
            seen[w] = sorted(ts)[3]
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
                              use_graph=use_graph,
                              vote_workgroups=vote_workgroups(idx.shape[0], dims) if callable(vote_workgroups) else vote_workgroups)
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



def workload_text(name, m, args):
    d, p0 = m["objs"][0]["dims"], m["objs"][0]["pipe"]
    heads = ("all 141 logits" if args.all_heads else
             "the 64 centre-bin logits (what the chain consumes up to the arg-max; the other heads belong to the second pass on the "
             "survivors)")
    if m["n_streams"] > 1:
        flight = f"{m['n_streams']} independent objects in flight on {m['n_streams']} HIP streams" + \
                 (f", each vote launched {p0.vote_workgroups} workgroups wide; " if p0.vote_workgroups else "; ")
    else:
        flight = "one object at a time; "
    if m.get("mlp_batch", 1) > 1:
        if m.get("vote_batch_workgroups"):
            votes = (f"their votes one vote launch and one reduce launch (cppf_vote_argmax_batch, {m['vote_batch_workgroups']} workgroups "
                     "per object" + (": calibrated during the warm-up)" if args.vote_batch_workgroups < 0 else ")"))
        else:
            votes = "each object then its own vote and reduce launch"
        launches = (f"the pair lists of {m['mlp_batch']} consecutive objects share one launch of the pair kernel "
                    f"(cppf_pair_mlp_decode_batch), {votes}; chains replayed from hipGraphs")
    else:
        launches = "four launches per step replayed from a hipGraph" if not args.no_graph else "eager launches"
    return (f"{name}: single object N={m['n_points']} K={m['k']} (P={m['P']} pairs), bottle config, res {m['objs'][0]['cfg'].res:g}, "
            f"grid {d[0]}x{d[1]}x{d[2]}, num_rots 72 adaptive, fused PPF+MLP(MFMA f32)+decode of {heads} -> LDS-tiled vote -> argmax "
            f"({m['what']}); one object per GPU per step, steps rotate over {m['n_obj']} distinct objects (own buffers: inputs come "
            f"from HBM, not the Infinity Cache), {flight}{launches}")


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



def compact(out):
    """The line rank 0 prints (< 4 KB): the contract's fields, the two rooflines, the CPU baseline's summary and the secondaries a
    reader needs first.  The full record (every stage, sweep and note: ~15 KB) goes to bench_full.json -- a harness that keeps
    the tail of stdout loses the head of a long line, and with it everything but the key names (BENCH_r04)."""
    pick = lambda d, keys: None if d is None else {k_: d[k_] for k_ in keys if k_ in d}
    line = {k_: out[k_] for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data") if k_ in out}
    cfg = dict(out["config"])
    if len(cfg.get("workload", "")) > 300:
        cfg["workload"] = cfg["workload"][:150] + " ... (full text: bench_full.json)"
    line["config"] = cfg
    for k_ in ("pairs_per_ms_per_gpu", "regions", "region_ms_min_max", "median_ms_one_instance", "vote_workgroups", "mlp_batch",
               "vote_batch", "vote_batch_workgroups", "vote_batch_calibration_ms_per_step", "dist", "cpu_binding", "argmax_matches_oracle",
               "argmax_objects_matching_oracle", "argmax_steps_matching_oracle", "argmax_objects_matching_oracle_per_rank",
               "records_matching_oracle", "objects_checked", "objects_per_s"):
        if k_ in out:
            line[k_] = out[k_]
    if out.get("c4_strong_scaling_predicted"):
        line["c4_strong_scaling_predicted"] = pick(out["c4_strong_scaling_predicted"],
                                                   ("t64_ms_per_object", "t8_ms_per_object", "batches_per_region", "speedup_8_gpus"))
    line["roofline"] = pick(out.get("roofline"), ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms",
                                                  "lists_per_launch",
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
        line["cpu_baseline"] = pick(cb, ("value", "best", "unit", "cores", "kind", "spread", "passes", "physical_cores",
                                         "host_threads_available"))
        line["cpu_baseline"]["bound"] = bool((cb.get("omp_binding") or {}).get("OMP_PROC_BIND"))      # threads bound close to cores?
        line["cpu_baseline"]["sample"] = cb["sample"][:60] + " ..."
        line["cpu_baseline"]["sweep_Mpairs_per_s"] = {str(e["threads"]): round(e["pairs_per_s"] / 1e6, 2) for e in cb.get("sweep", [])}
        if cb.get("other_binding"):
            line["cpu_baseline"]["other_binding"] = pick(cb["other_binding"], ("value", "cores"))
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
        line["stage_ms"] = pick(out["stage_ms"], ("ppf_mlp_decode_centre_heads", "vote_reduce_argmax",
                                                  "vote_reduce_argmax_known_answer_inputs",
                                                  "full_pose_incl_readback"))
    if out.get("other_configs"):
        line["other_configs"] = {nm: pick(v, ("ms_per_step", "ms_per_object", "argmax_steps_matching_oracle"))
                                 for nm, v in out["other_configs"].items()}
    if out.get("dropin_flow_reference_defaults"):
        line["dropin_flow_reference_defaults"] = pick(out["dropin_flow_reference_defaults"],
                                                      ("level1_reference_call_sequence_ms", "level2_estimate_pose_eager_ms",
                                                       "level3_batch_runner_captured_ms"))
    if out.get("real_frame"):
        line["real_frame"] = pick(out["real_frame"], ("instances", "ms_per_instance_incl_preprocessing", "ms_per_instance_pipelined",
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

