"""Average rocprofv3 --pmc counters per (kernel, grid).  usage: python tools/pmc_summary.py <counter_collection.csv>... [--filter substr]"""
import csv, sys, collections, re
files = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = None
if "--filter" in sys.argv: flt = sys.argv[sys.argv.index("--filter") + 1]; files = [f for f in files if f != flt]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in files:
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("gg::(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        if flt and flt not in name: continue
        key = (name, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d = (f, r["Dispatch_Id"])
        if d not in seen:
            seen.add(d); dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, cs in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
    print(f"== {key[0][:60]} blocks={key[1]} calls={len(dur[key])} avg_us(pmc run)={sum(dur[key])/len(dur[key]):.1f}")
    for c, v in sorted(cs.items()):
        print(f"     {c:32s} {sum(v)/len(v):16.0f}")
