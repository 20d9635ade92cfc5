"""Median per kernel of the counters in rocprofv3 counter_collection.csv files.  usage: pmc_summary.py <csv>..."""
import csv
import re
import statistics
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            k = re.sub(r"^void |\(.*$", "", k)
            suffix = ""
            m = re.match(r"k_rows_finish<.*?, (\d+), \d+>", k)       # lean / mid / full build of the correction kernel
            if m:
                suffix = {"0": "", "1": "_mid", "2": "_rest"}[m.group(1)]
            k = re.sub(r"<.*$", "", k).split("::")[-1] + suffix
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names))
for k in sorted(acc):
    print(k + "," + ",".join(f"{statistics.median(acc[k][c]):.0f}" if c in acc[k] else "" for c in names))
