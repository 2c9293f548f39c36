"""Per launch shape (kernel, grid, workgroup) totals of a rocprofv3 kernel trace: python3 tools/trace_shapes.py <kernel_trace.csv> <steps>."""
import csv, collections, sys
rows = csv.DictReader(open(sys.argv[1]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    short = name.split('(')[0][:58]
    key = (short, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))
    agg[key][0] += 1
    agg[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print("kernel ms/step", round(sum(v[1] for v in agg.values()) / steps / 1e3, 3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k[0]:58s} grid {k[1]:6d}x{k[2]:<4d} calls/step {v[0]/steps:5.1f} avg {v[1]/v[0]:7.1f} us  ms/step {v[1]/steps/1e3:6.3f}")
