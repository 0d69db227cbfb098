"""Summarises the rocprofv3 outputs of tools/run_cell_pmc.sh: per kernel the average duration (kernel trace) and the SQ counters per launch."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for f in glob.glob(os.path.join(root, "t", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Name"].split("(")[0]
        if "cell" in name:
            out.setdefault(name[:90], {})["avg_us"] = round(float(row["AverageNs"]) / 1e3, 1)
            out[name[:90]]["calls"] = int(row["Calls"])
for sub in ("a", "b"):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0][:90]
            if "cell" not in name:
                continue
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[name][row["Counter_Name"]] += 1
    for name in acc:
        for c in acc[name]:
            out.setdefault(name, {})[c] = round(acc[name][c] / max(cnt[name][c], 1))
print(json.dumps(out, indent=1))
