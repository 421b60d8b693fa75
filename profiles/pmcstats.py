#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from a *_counter_collection.csv."""
import csv
import collections
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%3d avg=%16.1f" % (c, len(v), sum(v) / len(v)))
