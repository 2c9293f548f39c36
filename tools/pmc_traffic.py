"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/<tag>_traffic.json, keyed by kernel symbol.

    python tools/pmc_traffic.py <tag> <label>=<fetch_dir>,<write_dir> [...]

Each pass directory holds a *_counter_collection.csv of one workload (tools/conv_once.py, tools/in_once.py,
tools/wgrad_once.py: a handful of launches of ONE shape).  Per kernel symbol of this library the average
counter value per launch is stored (KB, as rocprofv3 reports them; bench.py applies the gfx950 x2 correction
of FETCH_SIZE from MI355X_MICROARCH.md) together with the rows themselves (trimmed to this library's kernels)
as profiles/<tag>_<label>_pmc_{FETCH,WRITE}_SIZE.csv."""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def symbol(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                       # cut the argument list: the first '(' outside template brackets
        if ch == "<": depth += 1
        if ch == ">": depth -= 1
        if ch == "(" and depth == 0: break
        out.append(ch)
    return "".join(out).strip()


def read_pass(d, counter, keep):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    rows = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == counter]
    ours = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv_", "in_apply", "in_bwd", "col_partial", "wgrad", "adam"))]
    per = collections.defaultdict(list)
    for r in ours:
        per[symbol(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    with open(keep, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Symbol", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name",
                    "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        for r in ours:
            w.writerow([r["Dispatch_Id"], symbol(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"],
                        r["VGPR_Count"], r["Counter_Name"], r["Counter_Value"], r["Start_Timestamp"], r["End_Timestamp"]])
    return per


def main():
    tag = sys.argv[1]
    out = {}
    for spec in sys.argv[2:]:
        label, dirs = spec.split("=")
        fd, wd = dirs.split(",")
        fetch = read_pass(fd, "FETCH_SIZE", os.path.join(ROOT, "profiles", f"{tag}_{label}_pmc_FETCH_SIZE.csv"))
        write = read_pass(wd, "WRITE_SIZE", os.path.join(ROOT, "profiles", f"{tag}_{label}_pmc_WRITE_SIZE.csv"))
        for sym in fetch:
            if sym not in write:
                continue
            f, w = fetch[sym], write[sym]
            out.setdefault(label, {})[sym] = {"fetch_size_kb": sum(f) / len(f), "write_size_kb": sum(w) / len(w), "launches": len(f),
                                              "files": [f"profiles/{tag}_{label}_pmc_FETCH_SIZE.csv",
                                                        f"profiles/{tag}_{label}_pmc_WRITE_SIZE.csv"]}
    # TRAFFIC_JSON=<file>: merge into an existing file (another precision's symbols under the same labels)
    path = os.environ.get("TRAFFIC_JSON") or os.path.join(ROOT, "profiles", f"{tag}_traffic.json")
    if os.environ.get("TRAFFIC_JSON") and os.path.exists(path):
        old = json.load(open(path))
        for label, syms in out.items():
            old.setdefault(label, {}).update(syms)
        out = old
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
