"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "pmc_summary.txt"), "w") as fo:
    for name, ctrs in sorted(res.items()):
        if not any(k in name for k in ("spmm", "layer_dense", "copy_rows", "gather_rows")):
            continue
        line = name + ": " + ", ".join(f"{c} mean={sum(v)/len(v):.6g} n={len(v)}" for c, v in sorted(ctrs.items()))
        print(line); fo.write(line + "\n")
