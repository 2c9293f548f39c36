"""Turn a rocprofv3 --stats kernel_stats.csv into the markdown table kept under profiles/."""
import csv, sys
src, title = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
print(f"# {title}\n")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{sum(int(r['Calls']) for r in rows)} kernel launches, {tot / 1e6:.1f} ms of kernel time in total.\n")
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows[:30]:
    print(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
