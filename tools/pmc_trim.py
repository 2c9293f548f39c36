"""Trim a rocprofv3 --pmc counter_collection.csv to this library's kernels and the columns the notes under profiles/ use.
    python tools/pmc_trim.py <pass dir> <out.csv>"""
import csv, glob, os, re, sys
src = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Counter_Name", "Counter_Value"])
    for r in csv.DictReader(open(src)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
        if not any(t in k for t in ("conv_", "wgrad", "in_apply", "in_bwd", "col_partial")):
            continue
        k = re.sub(r"\((anonymous namespace::)?\w*Args.*$", "", k)
        w.writerow([r["Dispatch_Id"], k, r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Counter_Name"], r["Counter_Value"]])
