"""The product's world > 1 path executed for real: two ranks (processes) share GPU 0, rendezvous over gloo
(CPPF_DIST_BACKEND=gloo -- RCCL needs one GPU per rank), each runs BatchPoseRunner.run(objects, rank, world) on the SAME
mixed-category batch and must end up with the records of the single-rank run, bit for bit, in object order.  The reference's
analogue is its per-instance loop (nocs/inference.py:120); the collective is sharding.gather_records' one all_gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(n_objects=8, n_points=768, k=24):
    import cppf_amd.synthetic as syn
    from cppf_amd.config import NOCS_CATEGORIES
    objects = []
    for j in range(n_objects):
        ob = syn.make_object(NOCS_CATEGORIES[j % 6], n_points + 64 * (j % 3), 300 + j)      # ragged N: shape-polymorphic pipelines
        idx = syn.make_pairs(ob["pc"].shape[0], k, 300 + j)[:n_points * k]
        u_tr, u_rot = syn.make_uniforms(idx.shape[0], 300 + j)
        objects.append(dict(pc=ob["pc"], normals=ob["normals"], feat=ob["feat"], point_idxs=idx, u_tr=u_tr, u_rot=u_rot,
                            cfg=ob["cfg"]))
    return objects


def _encoders(dev):
    from cppf_amd.config import NOCS_CATEGORIES
    from cppf_amd.models.model import PPFEncoder
    encs = {}
    for i, c in enumerate(NOCS_CATEGORIES):
        torch.manual_seed(i)
        e = PPFEncoder([84, 32, 32, 16], 141)
        with torch.no_grad():
            e.final.weight.mul_(4)
            e.final.bias.mul_(4)
        encs[c] = e.eval().to(dev)
    return encs


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      CPPF_DIST_BACKEND="gloo")
    from cppf_amd import sharding
    from cppf_amd.batch import BatchPoseRunner
    r, w, local = sharding.init_distributed()
    assert (r, w, local) == (rank, world, 0)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    runner = BatchPoseRunner(_encoders(dev), dev)
    objects = _batch()
    recs = runner.run(objects, rank, world)
    recs2 = runner.run(objects, rank, world)             # replays (and adapted forms) give the same records
    assert torch.equal(recs, recs2)
    torch.save(recs.cpu(), os.path.join(out_dir, f"rank{rank}.pt"))
    # device-drawn pairs: (seed, object index) fixes the draw whatever rank runs the object
    objs_dev = [dict(pc=o["pc"], normals=o["normals"], feat=o["feat"], cfg=o["cfg"], n_pairs=o["point_idxs"].shape[0])
                for o in objects]
    torch.save(runner.run(objs_dev, rank, world, seed=11).cpu(), os.path.join(out_dir, f"dev_rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_single_rank_batch(oracle, golden, dev, tmp_path):
    from cppf_amd.batch import BatchPoseRunner
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    objects = _batch()
    runner = BatchPoseRunner(_encoders(dev), dev)
    single = runner.run(objects).cpu()
    assert single.shape == (8, 20) and single[:, 15].tolist() == list(range(8))
    objs_dev = [dict(pc=o["pc"], normals=o["normals"], feat=o["feat"], cfg=o["cfg"], n_pairs=o["point_idxs"].shape[0])
                for o in objects]
    single_dev = runner.run(objs_dev, seed=11).cpu()
    for rank in range(world):
        got = torch.load(os.path.join(tmp_path, f"rank{rank}.pt"))
        assert torch.equal(got, single), rank                     # every rank, object order, bit for bit
        assert torch.equal(torch.load(os.path.join(tmp_path, f"dev_rank{rank}.pt")), single_dev), rank
    # ... and the single-rank records are the oracle's poses (so the two-rank ones are too)
    sph = golden("sphere.npz")["pts"]
    sd = {k: v.detach().cpu().numpy() for k, v in _encoders(torch.device("cpu"))["bottle"].state_dict().items()}
    from cppf_amd.config import NOCS_CATEGORIES
    sds = {c: {k: v.detach().cpu().numpy() for k, v in e.state_dict().items()} for c, e in _encoders(torch.device("cpu")).items()}
    recs = single.numpy()
    for j, obj in enumerate(objects):
        cfg = obj["cfg"]
        ocfg = dict(res=cfg.res, tr_num_bins=32, rot_num_bins=36, vote_range=cfg.vote_range, scale_mean=cfg.scale_mean,
                    regress_right=cfg.regress_right, ppffcs=[84, 32, 32, 16], out_dim=141)
        o = oracle.estimate_pose(obj["pc"], obj["normals"], obj["feat"], obj["point_idxs"], sds[cfg.category], ocfg, obj["u_tr"],
                                 obj["u_rot"], sph)
        assert int(recs[j, 12]) == o["argmax"] and int(recs[j, 14]) == int(o["mask"].sum())
        np.testing.assert_allclose(recs[j, 0:3], o["T"], atol=1e-12)
        np.testing.assert_allclose(recs[j, 3:6], o["up"], atol=1e-12)
        np.testing.assert_allclose(recs[j, 9:12], o["scale"], rtol=1e-6)


def test_bench_runs_with_two_ranks_on_one_gpu(tmp_path):
    """the driver's scaling run launches `bench.py --gpus N` with one rank per GPU over RCCL; here the same code path (object per rank
    per step, barrier-bracketed timed region, max over ranks, ONE gather of the records, rank 0 prints the JSON line) with two ranks
    sharing GPU 0 over gloo -- for the single-object chain and for the 64-object batch of BASELINE.json configs[3]"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in (["--steps", "6", "--objects", "3"], ["--config", "c4", "--steps", "8", "--n-points", "1024", "--pairs-per-point", "32"]):
        port = _free_port()
        procs = []
        for rank in range(2):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       CPPF_DIST_BACKEND="gloo")
            procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--warmup", "1",
                                           "--no-secondary", "--no-cpu-baseline"] + extra, env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=600) for p in procs]
        assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
        lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # rank 0 alone prints
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["unit"] == "pairs/s" and d["higher_is_better"] is True
        assert d["metric"].startswith("point-pairs/sec") and "workload" in d["config"]
        assert d["scaling"] == ("strong" if "c4" in extra else "weak")


@pytest.mark.parametrize("extra", [["--steps", "6", "--objects", "3"],
                                   ["--config", "c4", "--steps", "8", "--n-points", "1024", "--pairs-per-point", "32"]])
def test_bench_launches_its_own_ranks(extra):
    """`python bench.py --gpus 2` with a CLEAN environment (no launcher: the shape of the driver's 1-GPU command with another N):
    bench.py re-runs itself as two ranks under torch.distributed.run; this box has one GPU, so the ranks share it over gloo and say
    so -- one JSON line, as the last line of stdout, n_gpus 2, ranks_seen 2"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT") and not k.startswith("CPPF_")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--warmup", "1", "--no-secondary",
                        "--no-cpu-baseline", "--min-seconds", "0.2"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    lines = [ln for ln in out_lines if ln.startswith("{")]
    assert len(lines) == 1 and out_lines[-1] == lines[0]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["unit"] == "pairs/s"
    n_dev = torch.cuda.device_count()
    assert d["dist"]["ranks_seen"] == 2 and len(d["dist"]["device_per_rank"]) == 2
    assert len(d["dist"]["rank_region_ms"]) == 2 and all(v > 0 for v in d["dist"]["rank_region_ms"]) and d["dist"]["imbalance"] >= 1.0
    if n_dev < 2:
        assert {k: d["dist"][k] for k in ("backend", "forced_single_rank", "ranks_seen", "device_per_rank", "shared_gpu")} == \
            {"backend": "gloo", "forced_single_rank": False, "ranks_seen": 2, "device_per_rank": [0, 0], "shared_gpu": True}
    else:
        assert d["dist"]["backend"] == "nccl" and d["dist"]["device_per_rank"] == [0, 1] and d["dist"]["shared_gpu"] is False
    # a launcher's environment with the wrong size is an error message and exit code 2, not an assertion trace
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-secondary", "--no-cpu-baseline"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode == 2 and "WORLD_SIZE=1" in bad.stderr and "Traceback" not in bad.stderr


@pytest.mark.parametrize("extra,n_checked", [(["--steps", "6", "--objects", "3", "--min-seconds", "0.2"], 48),
                                             (["--config", "c4", "--steps", "8", "--regions", "5"], 8)])
def test_eight_rank_line_checks_every_rank_against_the_oracle(extra, n_checked):
    """`python bench.py --gpus 8` the way the driver's scaling run issues it, on however many GPUs this box has (one: the eight ranks
    share it over gloo): the printed line must prove itself -- ranks_seen 8, every rank's objects against the oracle (the headline: the
    arg-max of every object every rank stepped through; c4: >= 1 full pose record per rank, pairs re-drawn on the host by the
    sampler's numpy twin), per-rank region times and their imbalance, the CPU binding, and cpu_baseline still on the line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT") and not k.startswith("CPPF_")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--warmup", "1", "--no-secondary"] + extra,
                       env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert out_lines[-1].startswith("{")
    d = json.loads(out_lines[-1])
    assert d["n_gpus"] == 8 and d["value"] > 0 and d["dist"]["ranks_seen"] == 8 and len(d["dist"]["device_per_rank"]) == 8
    assert len(d["dist"]["rank_region_ms"]) == 8 and all(v > 0 for v in d["dist"]["rank_region_ms"]) and d["dist"]["imbalance"] >= 1.0
    assert len(d["dist"]["cpu_binding"]) == 8
    assert d["argmax_matches_oracle"] is True and d["argmax_objects_matching_oracle"] == f"{n_checked}/{n_checked}"
    if "c4" in extra:
        assert d["records_matching_oracle"] == "8/8" and sorted(o % 8 for o in d["objects_checked"]) == list(range(8))
        assert d["scaling"] == "strong"
    else:
        assert d["argmax_objects_matching_oracle_per_rank"] == ["6/6"] * 8
        assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["spread"][1] <= d["cpu_baseline"]["best"]
