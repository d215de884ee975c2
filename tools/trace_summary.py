"""Group a rocprofv3 kernel trace (…_kernel_trace.csv) by (kernel, grid size): calls, avg / min us, ms per step.
usage: python tools/trace_summary.py <kernel_trace.csv> <steps_in_trace>"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("gg::(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    agg[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e3/steps:.2f} ms/step")
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{name[:58]:58s} blocks {grid:7d} calls/step {len(v)/steps:6.1f} avg {sum(v)/len(v):8.1f} min {min(v):8.1f} us  {sum(v)/1e3/steps:7.3f} ms/step")
