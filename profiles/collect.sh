#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default benchmark command  -> gpurun_out/prof/kernel_trace_stats.txt
#      (and with --streams 1: each launch alone on the chip      -> gpurun_out/prof/kernel_trace_stats_one_stream.txt)
#   2. PMC passes, one counter group per pass, no trace domains -> gpurun_out/prof/pmc_counters.txt
#   3. the HBM-traffic summary bench.py reads                   -> gpurun_out/prof/pmc_traffic.json  (kernels at full width)
#      and the same for the default command's narrower vote     -> gpurun_out/prof/pmc_traffic_timed_width.json
# Copy the three files into profiles/ (renamed r<round>_*) to have them judged.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --min-seconds 0.3"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- $BENCH > $OUT/bench_under_trace.log 2>&1
python $R/profiles/kstats.py $(find /tmp/kt -name '*.db' | head -1) > $OUT/kernel_trace_stats.txt 2>&1
# the same chain with ONE object in flight: every launch alone on the chip, so a kernel's average here is its own duration (in the
# default command three objects are in flight and a launch that shares the chip with a neighbour's head or tail lasts longer)
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -- $BENCH --streams 1 --no-secondary > $OUT/bench_under_trace_one_stream.log 2>&1
python $R/profiles/kstats.py $(find /tmp/kt1 -name '*.db' | head -1) > $OUT/kernel_trace_stats_one_stream.txt 2>&1
# ... and the launches the timed regions really make (4 objects per chain: ONE pair-kernel launch, ONE vote + ONE reduce launch), one chain
# at a time: the averages of pair_mlp_batch_kernel / v3_vote_batch_kernel here are what bench.py's roofline.launch_ms is about
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt4 -- $BENCH --streams 1 --mlp-batch 4 --vote-batch-workgroups 64 --no-secondary > $OUT/bench_under_trace_batch_one_stream.log 2>&1
python $R/profiles/kstats.py $(find /tmp/kt4 -name '*.db' | head -1) > $OUT/kernel_trace_stats_batch_one_stream.txt 2>&1
: > $OUT/pmc_counters.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  # (--vote-workgroups 0: every kernel at its full width, one workgroup per CU -- the launches bench.py's stage timings and
  #  rooflines are about; the timed regions' narrower vote is counted separately below)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc$i -- python $R/bench.py --steps 12 --mlp-batch 4 --warmup 2 --no-cpu-baseline --no-secondary --regions 5 --vote-workgroups 0 > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name '*counter_collection.csv' | head -1)
  python $R/profiles/pmcstats.py $f >> $OUT/pmc_counters.txt
done
python $R/profiles/make_traffic_json.py $OUT/pmc_counters.txt > $OUT/pmc_traffic.json
# the same two traffic counters for the default command: its timed regions vote 4 objects per launch, 64 workgroups each (three chains in flight)
: > $OUT/pmc_counters_timed_width.txt
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc$i -- python $R/bench.py --steps 12 --mlp-batch 4 --warmup 2 --no-cpu-baseline --no-secondary --regions 5 --vote-batch-workgroups 64 > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name '*counter_collection.csv' | head -1)
  python $R/profiles/pmcstats.py $f >> $OUT/pmc_counters_timed_width.txt
done
python $R/profiles/make_traffic_json.py $OUT/pmc_counters_timed_width.txt > $OUT/pmc_traffic_timed_width.json
# the batch drivers (round 6): per-kernel device microseconds PER INSTANCE (kstats.py <db> <rows> <instances>) of
#   FrameRunner on the reference's demo depth frame (bench.py real_frame)             -> kernel_trace_stats_frame.txt
#   the reference-default batch (100 000 pairs, kNN + SPRIN + pose; bench.py level3)  -> kernel_trace_stats_level3.txt (host-staged), _level3_resident.txt
#   one GPU's share of the C4 batch, 8 mixed-category objects resident on the device  -> kernel_trace_stats_c4_share.txt
batch_trace() {   # name, then env assignments + script
  local name=$1; shift
  timeout 600 env "$@" > /dev/null 2>&1   # (warm the file cache: the first import of a fresh box is slow)
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -- env "$@" > $OUT/$name.log 2>&1
  local n=$(grep -o "instances_processed [0-9]*" $OUT/$name.log | tail -1 | cut -d" " -f2)
  grep "ms per" $OUT/$name.log | tail -2 > $OUT/kernel_trace_stats_$name.txt
  python $R/profiles/kstats.py $(find /tmp/kt_$name -name '*.db' | head -1) 30 ${n:-1} >> $OUT/kernel_trace_stats_$name.txt 2>&1
}
batch_trace frame python $R/scripts/profile_frame.py
batch_trace level3 MODE=level3 python $R/scripts/profile_level3.py
batch_trace level3_resident MODE=level3 RESIDENT=1 python $R/scripts/profile_level3.py
batch_trace c4_share MODE=c4 RESIDENT=1 python $R/scripts/profile_level3.py
# the vote stage alone per configuration and regime (known-answer = what a trained network emits)
bash $R/profiles/vote_ktrace.sh c2 > $OUT/vote_regimes_ktrace.txt 2>&1
bash $R/profiles/vote_ktrace.sh c5 >> $OUT/vote_regimes_ktrace.txt 2>&1
bash $R/profiles/vote_ktrace.sh c2posed >> $OUT/vote_regimes_ktrace.txt 2>&1
tail -3 $OUT/bench_under_trace.log | cut -c1-300
