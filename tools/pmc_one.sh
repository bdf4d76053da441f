#!/bin/bash
# usage: tools/pmc_one.sh <outdir> "<counters>" <python script + args...>   (GPU box)
out=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export NGCF_NO_BUILD=1   # the profiler initialises the GPU in every process of the tree: never spawn a compiler under it
timeout -k 10 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out" -- python "$@" > "$out.log" 2>&1
python - "$out" <<'PY'
import csv,glob,collections,sys
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0]
        if "spmm" in n or "dense" in n: res[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(res.items()):
    print(k, {c: "%.4g"%(sum(x)/len(x)) for c,x in sorted(v.items())}, "n=%d"%len(next(iter(v.values()))))
PY
