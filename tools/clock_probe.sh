#!/bin/bash
# Average shader clock while a GEMM kernel runs: GRBM_GUI_ACTIVE cycles / kernel duration.
# Usage: tools/clock_probe.sh <nt|tn> [library.so]
K=${1:-nt}
[ -n "$2" ] && export MPNN_AMD_LIB=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/clk_$K -o p --output-format csv -- python3 $R/tools/gemm_one.py $K > /dev/null 2> $R/gpurun_out/clk_$K.err || exit 1
cd $R
python3 - $K <<'PY'
import csv, sys
k = sys.argv[1]
cyc = {}
for r in csv.DictReader(open(f"gpurun_out/clk_{k}/p_counter_collection.csv")):
    if "gemm" in r["Kernel_Name"]:
        e = cyc.setdefault(r["Dispatch_Id"], [r["Kernel_Name"][:40], 0.0])
        e[1] = max(e[1], float(r["Counter_Value"]))
dur = {}
for r in csv.DictReader(open(f"gpurun_out/clk_{k}/p_kernel_trace.csv")):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for d, (name, c) in cyc.items():
    if d in dur:
        print(f"{name}  cycles={c:.0f}  ns={dur[d]}  clock={c / dur[d]:.3f} GHz")
PY
