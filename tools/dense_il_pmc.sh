#!/bin/bash
# GPU box: effective clock and wave-cycle split of the dense kernels taken apart (tools/dense_il_parts_lab.py).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export NGCF_EXTRA_HIPCC_FLAGS=-DNGCF_LAB
python -c "from seoul_tourism_recommendation_ngcf_amd import _lib; _lib.load(); print('lab library built')"
export NGCF_NO_BUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out/r03/dense_il_pmc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o il -- python3 $GRAFT_REPO_ROOT/tools/dense_il_parts_lab.py > $out.log 2>&1
cd $GRAFT_REPO_ROOT
python - $out <<'PY'
import csv, glob, collections, sys, re
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "layer_dense_resident" not in n:
            continue
        m = re.search(r"il_kernel<(\d+), (true|false), (\d+)>", n)
        key = ("il lab=%2s" % m.group(3)) if m else "resident"
        res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur:
            res[key]["ms"].append(dur[r["Dispatch_Id"]])
with open(out + "_summary.txt", "w") as fo:
    for k, v in sorted(res.items()):
        mean = {c: sum(x) / len(x) for c, x in v.items()}
        ms = mean.get("ms", float("nan"))
        line = (f"{k:10s} ms {ms:.4f}  clock {mean['GRBM_GUI_ACTIVE'] / 8 / ms / 1e6:.3f} GHz  wave_cycles {mean['SQ_WAVE_CYCLES']:.4g} wait_any {mean['SQ_WAIT_ANY'] / mean['SQ_WAVE_CYCLES']:.3f} "
                f"wait_inst {mean['SQ_WAIT_INST_ANY'] / mean['SQ_WAVE_CYCLES']:.3f} active {mean['SQ_ACTIVE_INST_ANY'] / mean['SQ_WAVE_CYCLES']:.3f} "
                f"mfma_busy {mean['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} sq_busy {mean['SQ_BUSY_CYCLES']:.4g}")
        print(line); fo.write(line + "\n")
PY
