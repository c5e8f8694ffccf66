"""Per-kernel averages of a rocprofv3 counter_collection.csv (apply kernels only)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"]
    if "apply" not in name:
        continue
    acc[name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for cn, v in d.items():
        print("   %-28s %14.0f  (n=%d)" % (cn, sum(v) / len(v), len(v)))
