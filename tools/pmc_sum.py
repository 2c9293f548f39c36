import csv, sys, glob, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "conv_gather" in k or "wgrad3" in k:
                print(k, {c: round(sum(x[-3:]) / len(x[-3:])) for c, x in v.items()})
