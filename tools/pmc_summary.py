#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSV rows per (kernel, counter)."""
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r.get('Kernel_Name', '')[:60]
        acc[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for (k, c), v in sorted(acc.items()):
    if 'cnf::' in k:
        print("%-62s %-28s calls %4d  total %.4g  per-call %.4g" % (k, c, n[(k, c)], v, v / n[(k, c)]))
