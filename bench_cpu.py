"""The CPU legs of the benchmark: the oracle chain (the repo's C restatement of the reference path, OpenMP) timed on the host cores
-- `cpu_baseline` -- and its arg-max of every object the GPU stepped through -- the parity check of the printed line.  They run in
a worker process of their own (`bench.py --cpu-worker`): thread binding must be in the environment before the OpenMP runtime loads
and would pin the process that feeds the GPU.  Only this file (and tests/, __graft_entry__.smoke()) touches oracle/."""
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np
import torch

from bench_util import NUM_ROTS, ROOT, grid_shape, syn
from cppf_amd.models.model import PPFEncoder
import dataclasses

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


def original_cpus():
    """the CPUs this process could run on before bench_dist.bind_rank_cpus narrowed it to its GPU's share (None: never bound)"""
    text = os.environ.get("CPPF_BENCH_ORIG_CPUS")
    return {int(c_) for c_ in text.split(",")} if text else None


def physical_cores(allowed=None):
    """distinct (package, core) pairs among the CPUs this process may run on (0 when /proc/cpuinfo does not say)"""
    if allowed is None and os.environ.get("CPPF_BENCH_PHYSICAL_CORES"):
        return int(os.environ["CPPF_BENCH_PHYSICAL_CORES"])
    try:
        allowed = allowed if allowed is not None else os.sched_getaffinity(0)
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
    """The oracle chain at every thread count of the ladder (the vote leg keeps one private grid per thread and sums them, so more
    threads are not monotonically better: 256 threads were 3x slower than 8 on round 3's box).  Per count: ONE DISCARDED warm pass
    (a new count pays thread-pool creation and the first touch of its private grids: round 5's best-of-5 at 128 threads was 6x its
    own median), then CPU_PASSES timed passes (the budget may cut the last counts short, never below one pass); the count's figure is
    the MEDIAN pass, with the best and the [min, median, max] beside it.  Threads are bound (OMP_PROC_BIND=close OMP_PLACES=cores,
    set by the worker process this runs in: run_cpu_worker).  Returns (arg-max, entry of the best median, all entries)."""
    P = o["idx"].shape[0]
    t_start, entries, flat = time.perf_counter(), [], -1
    oracle_center(o, sd, threads=min(8, host_threads()), all_heads=all_heads)       # page in the library, the tables, the pools
    for th in thread_ladder():
        oracle_center(o, sd, threads=th, all_heads=all_heads)                       # warm pass at this count, not timed
        passes = []
        for _ in range(CPU_PASSES):
            flat, legs = oracle_center(o, sd, threads=th, all_heads=all_heads)
            passes.append((sum(legs.values()), legs))
            if time.perf_counter() - t_start > budget_s:
                break
        passes.sort(key=lambda q: q[0])
        rates = sorted(P / q[0] for q in passes)
        med = passes[(len(passes) - 1) // 2]                 # the median pass (the slower middle one of an even count)
        entries.append({"threads": th, "pairs_per_s": P / med[0], "best_pairs_per_s": P / passes[0][0], "passes": len(passes),
                        "spread_pairs_per_s": [rates[0], rates[len(rates) // 2], rates[-1]],
                        "legs_ms": {k_: v * 1e3 for k_, v in med[1].items()}})
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
        value=best["pairs_per_s"], best=best["best_pairs_per_s"], unit="pairs/s", cores=best["threads"], kind="port",
        best_threads=best["threads"], host_threads_available=host_threads(), physical_cores=physical_cores(),
        legs=best["legs_ms"], spread=best["spread_pairs_per_s"], passes=best["passes"],
        omp_binding={k_: os.environ.get(k_) for k_ in ("OMP_PROC_BIND", "OMP_PLACES")},
        sweep=entries,
        sample=f"full workload (N={n_points}, K={k}, P={P}): value = the MEDIAN of {CPU_PASSES} passes (after one discarded warm "
               f"pass) at the best thread count of {thread_ladder()}, `best` = that count's fastest pass, spread = [min, median, "
               "max] pairs/s of its passes; threads bound close to cores; the repo's C oracle with OpenMP -- AVX2 fmaf-chain MLP + "
               "decode + vote (private grid per thread) + arg-max (the reference has no CPU vote path); legs in ms",
        mlp_torch_cpu={"value": tbest["pairs_per_s"], "unit": "pairs/s", "threads": tbest["threads"], "sweep": tentries,
                       "sample": f"PPF + gather + ResLayers + final as torch ops on the host (models/model.py:118-137), "
                                 f"{tn} pairs, same weights; MLP leg only; best thread count of the same ladder"})


def run_cpu_worker(jobs, timeout=900, bind=True):
    """The CPU legs (cpu_baseline sweeps, the oracle's arg-max of every object) in a process of their own: thread binding
    (OMP_PROC_BIND=close OMP_PLACES=cores must be in the environment before the OpenMP runtimes load, and would pin THIS process's
    main thread -- the one that feeds the GPU -- to one core), no distributed environment.  jobs: list of dicts, see cpu_worker."""
    # a rank of a multi-GPU run is bound to the CPUs next to its GPU (bench_dist.bind_rank_cpus); the CPU legs are the HOST's numbers:
    # the worker gets the whole machine back
    orig = original_cpus()
    env = dict(os.environ, CPPF_BENCH_HOST_THREADS=str(len(orig) if orig else host_threads()),
               CPPF_BENCH_PHYSICAL_CORES=str(physical_cores(orig) if orig else physical_cores()))
    if bind:
        env.update(OMP_PROC_BIND="close", OMP_PLACES="cores")
    else:
        env.pop("OMP_PROC_BIND", None)
        env.pop("OMP_PLACES", None)
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "CPPF_FORCE_DIST", "OMP_NUM_THREADS", "TORCHELASTIC_RUN_ID",
               "CPPF_BENCH_ORIG_CPUS"):
        env.pop(k_, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", json.dumps(jobs)], env=env, capture_output=True,
                       text=True, timeout=timeout, preexec_fn=(lambda: os.sched_setaffinity(0, orig)) if orig else None)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("[")]
    if p.returncode != 0 or not lines:
        raise RuntimeError(f"bench.py --cpu-worker failed ({p.returncode}): {p.stderr[-2000:]}")
    return json.loads(lines[-1])


def cpu_worker(spec):
    """`bench.py --cpu-worker '<json>'` (internal): jobs {"kind": "baseline", n_points, k, seed, res, all_heads, budget_s} -> the
    cpu_baseline block + the object's arg-max; {"kind": "argmax", n_points, k, seeds, res, threads} -> the oracle's arg-max of
    every object (the same generator and seeds as the GPU side's make_center_set); {"kind": "c4pose", ...} -> c4_oracle_poses.  One
    JSON list on stdout."""
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
        elif job["kind"] == "c4pose":
            out.append({"poses": c4_oracle_poses(job)})
        else:
            flats = []
            for seed in job["seeds"]:
                o = cpu_object(job["n_points"], job["k"], seed, job.get("res"))
                flats.append(int(oracle_center(o, sd, threads=job.get("threads") or min(32, host_threads()))[0]))
            out.append({"argmax": flats})
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()




def c4_oracle_poses(job):
    """{"kind": "c4pose", n_points, k, seed0, seed, objects: [ids]}: the oracle's FULL pose (oracle.estimate_pose: MLP, decode, vote,
    arg-max, back-vote, second pass, orientation vote, sign, scale) of objects of bench.c4_objects -- same generator, same
    per-category weights (bench.c4_encoders), and the pairs / bin uniforms the device drew (cppf_amd.synthetic.philox_pairs, the host
    twin of cppf_stage_batch's sampler, keyed like BatchPoseRunner: seed * 1000003 + object id)"""
    from oracle import oracle as O
    from bench import c4_encoders
    from cppf_amd.config import NOCS_CATEGORIES
    from cppf_amd.utils.util import fibonacci_sphere
    O.set_threads(min(32, host_threads()))
    sph = np.array(fibonacci_sphere(480))
    sds = {c: {k_: v.detach().numpy().copy() for k_, v in e.state_dict().items()} for c, e in c4_encoders().items()}
    out = []
    for j in job["objects"]:
        cat = NOCS_CATEGORIES[j % len(NOCS_CATEGORIES)]
        ob = syn.make_object(cat, job["n_points"], job["seed0"] + j)
        cfg = ob["cfg"]
        idx, u_tr, u_rot = syn.philox_pairs(job["seed"] * 1000003 + j, job["n_points"] * job["k"], job["n_points"])
        ocfg = dict(res=cfg.res, tr_num_bins=cfg.tr_num_bins, rot_num_bins=cfg.rot_num_bins, vote_range=cfg.vote_range,
                    scale_mean=cfg.scale_mean, regress_right=cfg.regress_right, ppffcs=cfg.ppffcs, out_dim=cfg.out_dim)
        o = O.estimate_pose(ob["pc"], ob["normals"], ob["feat"], idx, sds[cat], ocfg, u_tr, u_rot, sph)
        out.append({"object": j, "argmax": int(o["argmax"]), "T": [float(v) for v in o["T"]], "up": [float(v) for v in o["up"]],
                    "scale": [float(v) for v in o["scale"]], "n_surv": int(o["mask"].sum())})
    return out


def check_c4_records(m, sample, seed=0, seed0=500):
    """the gathered records of the sampled objects of a --config c4 run against the oracle's poses: arg-max and survivor count
    bit for bit, T and up to 1e-9, scale to 1e-5 (north_star: arg-max identical, pose within 1e-4) -> fields of the printed line"""
    poses = run_cpu_worker([{"kind": "c4pose", "n_points": m["n_points"], "k": m["k"], "seed0": seed0, "seed": seed,
                             "objects": list(sample)}])[0]["poses"]
    recs, ok_arg, ok_pose = m["recs"], 0, 0
    for p in poses:
        r = recs[p["object"]]
        same = int(r[12]) == p["argmax"] and int(r[15]) == p["object"]
        ok_arg += same
        ok_pose += bool(same and int(r[14]) == p["n_surv"] and np.allclose(r[0:3], p["T"], atol=1e-9) and
                        np.allclose(r[3:6], p["up"], atol=1e-9) and np.allclose(r[9:12], p["scale"], rtol=1e-5))
    return {"argmax_matches_oracle": bool(ok_arg == len(poses)), "argmax_objects_matching_oracle": f"{ok_arg}/{len(poses)}",
            "records_matching_oracle": f"{ok_pose}/{len(poses)}", "objects_checked": [p["object"] for p in poses]}


def attach_checks(out, m, args, rank, world, pending_checks, step_argmaxes):
    """rank 0, after the timed regions (and a barrier): cpu_baseline on object 0, and the oracle's arg-max of EVERY object EVERY rank
    stepped through (rank r's objects are seeds 100 r + i; the batched, XCD-pinned launches included) against the gathered records;
    the other configurations' objects (pending_checks) and BASELINE.json configs[0] ride along in the same worker."""
    from bench_util import CONFIGS
    steps, n_obj = args.steps, m["n_obj"]
    n_chk = min(n_obj, steps)
    res = CONFIGS[args.config]["res"]
    jobs = [{"kind": "baseline", "n_points": m["n_points"], "k": m["k"], "seed": 0, "res": res, "all_heads": args.all_heads},
            {"kind": "argmax", "n_points": m["n_points"], "k": m["k"], "res": res,
             "seeds": [100 * r + i for r in range(world) for i in range(n_chk)]}]
    names = list(pending_checks)
    jobs += [pending_checks[nm][2] for nm in names]
    with_c1 = args.config == "c2" and not args.no_secondary and world == 1
    if with_c1:
        jobs.append({"kind": "baseline", "n_points": 1024, "k": 64, "seed": 0, "budget_s": 6.0})
    res_w = run_cpu_worker(jobs)
    out["cpu_baseline"] = res_w[0]["cpu_baseline"]
    # ... and the same sweep with the threads left to the scheduler (a shorter budget): the baseline is the better of the two
    free = run_cpu_worker([dict(jobs[0], budget_s=10.0)], bind=False)[0]["cpu_baseline"]
    brief = lambda cb_: {kk: cb_[kk] for kk in ("value", "best", "cores", "spread", "passes", "omp_binding", "legs")}
    if free["value"] > out["cpu_baseline"]["value"]:
        out["cpu_baseline"], free = free, out["cpu_baseline"]
    out["cpu_baseline"]["other_binding"] = brief(free)
    want_all = res_w[1]["argmax"]
    ok_steps = ok_objs = n_steps = 0
    per_rank = []
    for r in range(world):
        want = want_all[r * n_chk:(r + 1) * n_chk]
        got = step_argmaxes(m, steps, r, world)
        oks = sum(1 for i_, g_ in enumerate(got) if g_ == want[i_ % n_obj])
        oko = sum(1 for j_ in range(n_chk) if all(g_ == want[j_] for g_ in got[j_::n_obj]))
        ok_steps, ok_objs, n_steps = ok_steps + oks, ok_objs + oko, n_steps + len(got)
        per_rank.append(f"{oko}/{n_chk}")
    out["argmax_matches_oracle"] = bool(ok_steps == n_steps)
    out["argmax_objects_matching_oracle"] = f"{ok_objs}/{world * n_chk}"
    out["argmax_steps_matching_oracle"] = f"{ok_steps}/{n_steps}"
    if world > 1:
        out["argmax_objects_matching_oracle_per_rank"] = per_rank
    other = out.get("other_configs") or {}
    for q, nm in enumerate(names):
        got_o, n_obj_o, _ = pending_checks[nm]
        want_o = res_w[2 + q]["argmax"]
        ok_o = sum(1 for i_, g_ in enumerate(got_o) if g_ == want_o[i_ % n_obj_o])
        other[nm]["argmax_matches_oracle"] = bool(ok_o == len(got_o))
        other[nm]["argmax_steps_matching_oracle"] = f"{ok_o}/{len(got_o)}"
    if with_c1:
        c1 = res_w[-1]["cpu_baseline"]
        out["cpu_baseline"]["c1"] = {kk: c1[kk] for kk in ("value", "unit", "best_threads", "legs", "spread", "sample")}
