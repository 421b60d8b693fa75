#!/bin/bash
# rocprofv3 PMC passes over the vote kernel alone, one regime at a time (kernel-development aid; run through gpurun from the repo
# root):  bash profiles/collect_vote.sh c2 known-answer  -> gpurun_out/prof/vote_pmc_<cfg>_<regime>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c2}; REG=${2:-known-answer}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
F=$OUT/vote_pmc_${CFG}_${REG}.txt
: > $F
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/vpmc_${CFG}_${REG}_$i -- python $R/profiles/microbench/vote_regimes.py $CFG $REG > /tmp/vpmc$i.log 2>&1
  f=$(find /tmp/vpmc_${CFG}_${REG}_$i -name '*counter_collection.csv' | head -1)
  python $R/profiles/pmcstats.py $f v >> $F
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/vkt_${CFG}_${REG} -- python $R/profiles/microbench/vote_regimes.py $CFG $REG > /tmp/vkt.log 2>&1
python $R/profiles/kstats.py $(find /tmp/vkt_${CFG}_${REG} -name '*.db' | head -1) 2>&1 | grep -E "v3_|vote_kernel|reduce_tiles|Name" >> $F
cat $F
