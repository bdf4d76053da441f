#!/bin/bash
# GPU box: wave-cycle split of the wide dense kernels at the Seoul row count (tools/dense_wide_lab.py, LAB_ROWS=5940)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export NGCF_NO_BUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out/r03/dense_wide_pmc
cd /tmp && export TMPDIR=/tmp
LAB_ROWS=5940 timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o w -- python3 $GRAFT_REPO_ROOT/tools/dense_wide_lab.py > $out.log 2>&1
cd $GRAFT_REPO_ROOT
python - $out <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "layer_dense" not in n:
            continue
        key = n.split("(")[0][-40:] + " grid " + r["Grid_Size"]
        res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur:
            res[key]["us"].append(dur[r["Dispatch_Id"]])
with open(out + "_summary.txt", "w") as fo:
    for k, v in sorted(res.items()):
        m = {c: sum(x) / len(x) for c, x in v.items()}
        wc = m["SQ_WAVE_CYCLES"]
        line = (f"{k:60s} us {m.get('us', 0):7.1f} wave_cycles {wc:.4g} wait_any {m['SQ_WAIT_ANY'] / wc:.3f} wait_inst {m['SQ_WAIT_INST_ANY'] / wc:.3f} "
                f"(lds {m['SQ_WAIT_INST_LDS'] / wc:.3f}) active {m['SQ_ACTIVE_INST_ANY'] / wc:.3f} mfma_busy {m['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} "
                f"lds_conflict {m['SQ_LDS_BANK_CONFLICT']:.4g} of lds_active {m['SQ_LDS_IDX_ACTIVE']:.4g}")
        print(line); fo.write(line + "\n")
PY
