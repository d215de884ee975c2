"""Per-kernel MFMA utilisation and issue / stall shares from the rocprofv3 --pmc passes of tools/pmc_mfma.sh.

usage: python tools/pmc_mfma_summary.py gpurun_out/pmcM1/*/*_counter_collection.csv gpurun_out/pmcM2/... gpurun_out/pmcM3/... > profiles/rNN_pmc_mfma.json

Normalisations (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is summed over the 8 XCDs -> shader cycles of the dispatch = value / 8;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix core is busy, summed over the 1024 SIMDs -> utilisation =
value / (cycles * 1024); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave -> shares of the waves' time."""
import collections
import csv
import json
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sys.argv[1:]:
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("gg::(anonymous namespace)::", "").replace("void ", "").replace("gg::", "")
        name = re.sub(r"\(.*", "", name)
        key = (name, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d = (f, r["Dispatch_Id"])
        if d not in seen:
            seen.add(d)
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for key, cs in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if "GRBM_GUI_ACTIVE" not in m or "SQ_WAVE_CYCLES" not in m:
        continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    wc = max(m["SQ_WAVE_CYCLES"], 1.0)
    row = {"kernel": key[0], "workgroups": key[1], "launches_in_trace": len(dur[key]) // 3,
           "avg_us_profiled": round(sum(dur[key]) / len(dur[key]), 1),
           "total_ms_in_trace": round(sum(dur[key]) / 3e3, 2),
           "mfma_util": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0), 4),
           "mfma_instructions": int(m.get("SQ_INSTS_MFMA", 0)),
           "wave_time_waiting_on_memory_counters": round(m.get("SQ_WAIT_ANY", 0.0) / wc, 3),
           "wave_time_waiting_to_issue": round(m.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
           "wave_time_valu": round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 3),
           "wave_time_lds": round(m.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 3),
           "lds_bank_conflict_cycles_per_lds_cycle": round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(4.0 * m.get("SQ_ACTIVE_INST_LDS", 0.0), 1.0), 3),
           "l2_hit_rate": round(m.get("TCC_HIT_sum", 0.0) / max(m.get("TCC_HIT_sum", 0.0) + m.get("TCC_MISS_sum", 0.0), 1.0), 3),
           "effective_clock_ghz": round(cyc / (sum(dur[key]) / len(dur[key])) / 1e3, 2)}
    out.append(row)
json.dump({"command": "bash tools/pmc_mfma.sh (bench.py --steps 2 --warmup 1, three rocprofv3 --pmc passes)",
           "kernels": [r for r in out if r["total_ms_in_trace"] >= 0.05]}, sys.stdout, indent=1)
