"""Per-kernel table from the two passes of tools/probe_pmc.sh (see tools/pmc_mfma_summary.py for the normalisations)."""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sys.argv[1:]:
    seen = set()
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("gg::(anonymous namespace)::", "").replace("void ", "").replace("gg::", ""))
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d = (f, r["Dispatch_Id"])
        if d not in seen:
            seen.add(d)
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, cs in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if "SQ_WAVE_CYCLES" not in m or sum(dur[name]) < 50:
        continue
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wc = max(m["SQ_WAVE_CYCLES"], 1.0)
    us = sum(dur[name]) / len(dur[name])
    print(f"{name[:70]:70s} n={len(dur[name])//2:3d} {us:8.1f} us  clk {cyc/us/1e3:4.2f} GHz  mfma_util {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(cyc*1024,1):5.3f} "
          f"wait_cnt {m.get('SQ_WAIT_ANY',0)/wc:5.3f} wait_issue {m.get('SQ_WAIT_INST_ANY',0)/wc:5.3f} wait_lds {m.get('SQ_WAIT_INST_LDS',0)/wc:5.3f} "
          f"active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:5.3f} valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:5.3f} lds {m.get('SQ_ACTIVE_INST_LDS',0)/wc:5.3f} "
          f"bank_conf {m.get('SQ_LDS_BANK_CONFLICT',0)/max(4*m.get('SQ_ACTIVE_INST_LDS',0),1):5.3f} "
          f"insts: mfma {m.get('SQ_INSTS_MFMA',0):.3g} valu {m.get('SQ_INSTS_VALU',0):.3g} lds {m.get('SQ_INSTS_LDS',0):.3g} waves {m.get('SQ_WAVES',0):.0f}")
