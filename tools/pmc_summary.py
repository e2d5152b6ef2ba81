#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per launch of one kernel: tools/pmc_summary.py <counter_collection.csv> <kernel substring> [min grid]"""
import collections
import csv
import json
import sys

path, kern = sys.argv[1], sys.argv[2]
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
acc = collections.defaultdict(list)
dur = []
with open(path) as f:
    for row in csv.DictReader(f):
        if kern not in row['Kernel_Name'] or int(row['Grid_Size']) < grid:
            continue
        acc[row['Counter_Name']].append(float(row['Counter_Value']))
        dur.append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
out = {k: sum(v) / len(v) for k, v in acc.items()}
out['launches'] = max(len(v) for v in acc.values()) if acc else 0
out['avg_ns'] = sum(dur) / max(len(dur), 1)
print(json.dumps(out, indent=1))
