"""Mean per-launch value of every counter in rocprofv3 counter_collection CSV files, for kernels whose
name contains --match."""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("--match", required=True)
ap.add_argument("files", nargs="+")
args = ap.parse_args()
for path in args.files:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if args.match not in k:
                continue
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(path.split("/")[-2], k)
        for c, v in d.items():
            print("   %-32s mean/launch %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
