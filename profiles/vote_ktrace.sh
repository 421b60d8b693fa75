cd /tmp && export TMPDIR=/tmp
for reg in known-answer uniform-bin; do
rocprofv3 --kernel-trace --stats -d /tmp/vkt_$1_$reg -- python $GRAFT_REPO_ROOT/profiles/microbench/vote_regimes.py $1 $reg > /tmp/vkt.log 2>&1
echo "== $1 $reg"
python $GRAFT_REPO_ROOT/profiles/kstats.py $(find /tmp/vkt_$1_$reg -name '*.db' | head -1) 2>&1 | grep -E "v3_|vote_kernel|reduce_|Name"
done
