"""profiles/rNN_pmc_traffic.json from the two passes of tools/pmc_traffic.sh.
usage: python tools/pmc_traffic_summary.py gpurun_out/pmc_fetch/*/*_counter_collection.csv gpurun_out/pmc_write/*/*_counter_collection.csv [out.json]"""
import csv, sys, json, re, collections
acc = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for f in sys.argv[1:3]:
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        name = r["Kernel_Name"].replace("gg::(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    if not v["FETCH_SIZE"] or not v["WRITE_SIZE"] or not (k.startswith("tlin") or k.startswith("wst") or k.startswith("gp_") or k.startswith("attn") or k.startswith("wgrad") or "ln_bwd" in k or k.startswith("sqx")):
        continue
    fb = 2.0 * 1024 * sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])      # gfx950: FETCH_SIZE counts half of a wide coalesced read
    wb = 1024.0 * sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    out[k] = {"launches": len(v["FETCH_SIZE"]), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
doc = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_traffic.sh) on `bench.py --steps 1 "
                 "--warmup 1`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE halves wide coalesced reads, "
                 "MI355X_MICROARCH.md HBM section); launch-weighted means per kernel instantiation",
       "steps_profiled": 2,
       "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["launches"] * kv[1]["hbm_bytes_per_launch"]))}
json.dump(doc, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/r03_pmc_traffic.json", "w"), indent=1)
doc["hbm_bytes_per_step"] = sum(v["launches"] * v["hbm_bytes_per_launch"] for v in doc["kernels"].values()) / doc["steps_profiled"]
json.dump(doc, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/r03_pmc_traffic.json", "w"), indent=1)
print(f"HBM bytes per step (heavy kernels): {doc['hbm_bytes_per_step'] / 1e9:.1f} GB")
for k, v in doc["kernels"].items():
    print(f"{k[:56]:56s} launches {v['launches']:4d}  {v['hbm_bytes_per_launch']/1e6:8.1f} MB/launch")
