#!/bin/bash
# Full evidence pass of a round on the GPU box (gpurun -- bash profiles/collect_all.sh): the GPU suite, the bench line of every
# BASELINE.json config, then profiles/collect.sh (rocprofv3 kernel traces and PMC passes); outputs under gpurun_out/{final,prof}/,
# copied into profiles/ as r<round>_* by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/final/gputest.txt
python bench.py --steps 20 --warmup 5 --full-record gpurun_out/final/bench_default_full.json > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
python bench.py --steps 20 --warmup 5 --all-heads --no-cpu-baseline --no-secondary --full-line --full-record "" > gpurun_out/final/bench_all_heads.json 2>/dev/null
for c in c3 c5; do python bench.py --steps 20 --warmup 5 --config $c --no-cpu-baseline --no-secondary --full-line --full-record "" > gpurun_out/final/bench_$c.json 2>/dev/null; done
python bench.py --config c4 --full-line --full-record "" > gpurun_out/final/bench_c4.json 2>/dev/null     # (with the oracle check of 8 records)
# two ranks on this box's one GPU (gloo rendezvous, shared device: a code-path run, not a scaling measurement) -- the line must prove
# itself: every rank's objects against the oracle, per-rank times, CPU binding, cpu_baseline (bench.py, bench_dist.py)
python bench.py --gpus 2 --no-secondary --min-seconds 1 --full-record "" > gpurun_out/final/bench_gpus2_shared.json 2>/dev/null
python bench.py --gpus 2 --config c4 --full-record "" > gpurun_out/final/bench_c4_gpus2_shared.json 2>/dev/null
python bench.py --config c1 --full-line --full-record "" > gpurun_out/final/bench_c1.json 2>/dev/null
bash profiles/collect.sh > gpurun_out/final/collect.log 2>&1
cat gpurun_out/final/gputest.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.4g"%d["value"], "ms %.5f"%d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("launch_ms"))
    except Exception as e: print(f, "failed", e)
PY
