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
            rest = k.startswith("k_rows_finish<") and k.rstrip().endswith("false>")    # the full build, launched after the lean one
            k = re.sub(r"<.*$", "", k).split("::")[-1] + ("_rest" if rest else "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names))
for k in sorted(acc):
    print(k + "," + ",".join(f"{statistics.median(acc[k][c]):.0f}" if c in acc[k] else "" for c in names))
