#!/usr/bin/env python3
"""pmcstats.py output -> the per-kernel HBM traffic summary bench.py reads (profiles/r<round>_pmc_traffic.json)."""
import collections
import json
import re
import sys

acc = collections.defaultdict(dict)
kernel = None
for line in open(sys.argv[1]):
    m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+avg=\s*([0-9.eE+-]+)", line)
    if m and kernel:
        acc[kernel][m.group(1)] = float(m.group(2))
    elif line.strip() and not line.startswith(" "):
        kernel = line.strip()
names = {"point_proj_kernel": "point_proj_kernel", "pair_mlp_kernel<false, true, true, false>": "pair_mlp_kernel<false, true, true>",
         "pair_mlp_kernel<false, true, false, false>": "pair_mlp_kernel<false, true, false>",
         "pair_mlp_batch_kernel<false, false>": "pair_mlp_batch_kernel<false>", "pair_mlp_batch_kernel<true, false>": "pair_mlp_batch_kernel<true>",
         "point_proj_batch_kernel": "point_proj_batch_kernel",
         "reduce_argmax_kernel": "reduce_argmax_kernel", "v3_vote_kernel<true, false>": "v3_vote_kernel<true>",
         "v3_vote_kernel<false, false>": "v3_vote_kernel<false>", "v3_bin_kernel<false>": "v3_bin_kernel", "v3_reduce_kernel": "v3_reduce_kernel",
         "v3_vote_batch_kernel": "v3_vote_batch_kernel", "v3_reduce_batch_kernel": "v3_reduce_batch_kernel",
         "sprin_conv_kernel": "sprin_conv_kernel", "knn_kernel<false>": "knn_kernel<false>"}
out = {"_source": "rocprofv3 --pmc <one counter group per pass> -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "
                  "(MI355X; profiles/collect.sh); per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB as reported; gfx950: "
                  "FETCH_SIZE = TCC_EA0_RDREQ x 64 B under-counts wide coalesced reads 2x (MI355X_MICROARCH.md), so "
                  "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, an upper estimate for narrow accesses"}
for full, d in acc.items():
    for key, short in names.items():
        if key in full and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            e = {"FETCH_SIZE_KiB": round(d["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(d["WRITE_SIZE"], 1)}
            for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "SQ_INSTS_VALU", "SQ_INSTS_MFMA",
                      "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"):
                if c in d:
                    e[c] = round(d[c])
            e["hbm_bytes"] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
            out[short] = e
print(json.dumps(out, indent=1))
