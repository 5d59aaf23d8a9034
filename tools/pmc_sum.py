"""Sums rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_sum.py <dir> -> table (kernel x counter)."""
import csv, glob, os, sys, collections, re
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void jxlhip::", "").replace("jxlhip::", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add(row["Dispatch_Id"])
names = sorted({c for k in tot for c in tot[k]})
print("kernel,calls," + ",".join(names))
for k in sorted(tot):
    print(k + "," + str(len(calls[k])) + "," + ",".join("%.0f" % tot[k].get(c, 0) for c in names))
