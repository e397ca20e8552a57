"""Sums rocprofv3 --pmc counter CSVs per kernel.  usage: pmc_summarise.py DIR  (DIR holds *_counter_collection.csv of one or more passes)
Prints per-kernel launches and the per-launch average of every counter found (FETCH_SIZE / WRITE_SIZE are in KiB)."""
import csv, glob, os, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]].add(row["Dispatch_Id"])
out = {}
for k in sorted(acc):
    out[k] = {c: {"launches": len(cnt[k][c]), "per_launch": acc[k][c] / max(1, len(cnt[k][c]))} for c in acc[k]}
print(json.dumps(out, indent=1))
