export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/clk; mkdir -p gpurun_out/clk
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/clk -- python3 tools/one_conv.py fwd > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
cc = {}
for f in glob.glob("gpurun_out/clk/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv3x3_patch" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cc.setdefault(r["Dispatch_Id"], 0.0); cc[r["Dispatch_Id"]] += float(r["Counter_Value"])
dur = {}
for f in glob.glob("gpurun_out/clk/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "conv3x3_patch" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for k in cc:
    if k in dur:
        print("dispatch", k, "GRBM_GUI_ACTIVE", cc[k], "time ms", dur[k] * 1e3, "clock GHz (÷8)", cc[k] / 8 / dur[k] / 1e9)
PY
rm -rf gpurun_out/clk
