"""Per-stream busy time and the largest idle gaps of the main stream inside the timed bf16 window of a bench.py kernel
trace (rocprofv3 --kernel-trace).  usage: python tools/stream_timeline.py <kernel_trace.csv> <steps> [ms_per_step]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]); ms = float(sys.argv[3]) if len(sys.argv) > 3 else 38.6
def nm(r):
    n = r["Kernel_Name"].replace("gg::(anonymous namespace)::", "").replace("void ", "").replace("gg::", "")
    return re.sub(r"\(.*", "", n)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bf = rows          # run bench.py with --no-parity-mode: the whole trace is the bf16 run
eng = [r for r in bf if any(k in r["Kernel_Name"] for k in ("tlin_", "attn_", "wgrad_", "gemm_small"))] or bf
t_end = max(int(r["End_Timestamp"]) for r in eng)            # the last engine kernel: host-side epilogue work is not a step
t0 = t_end - int(steps * ms * 1e6)
win = [r for r in bf if t0 <= int(r["Start_Timestamp"]) <= t_end]
by = collections.defaultdict(list)
for r in win: by[r["Stream_Id"]].append(r)
print("window %.2f ms, %d dispatches" % ((t_end - t0) / 1e6, len(win)))
for s, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v) / 1e6
    print(f"stream {s}: {len(v)} kernels, busy {busy/steps:.2f} ms/step")
main = max(by.items(), key=lambda kv: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kv[1]))[1]
# gaps on the main stream, grouped by (previous kernel -> next kernel)
gaps = collections.defaultdict(list)
for a, b in zip(main, main[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gaps[(nm(a)[:34], nm(b)[:34])].append(g)
tot = sum(sum(v) for v in gaps.values())
print("main-stream idle total %.2f ms/step" % (tot / 1e3 / steps))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{k[0]:34s} -> {k[1]:34s} n/step {len(v)/steps:5.1f} avg gap {sum(v)/len(v):7.1f} us  {sum(v)/1e3/steps:6.3f} ms/step")
agg = collections.defaultdict(list)
for r in main: agg[nm(r)[:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("main-stream kernels:")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:30]:
    print(f"  {k:50s} n/step {len(v)/steps:6.1f} avg {sum(v)/len(v):7.1f} us {sum(v)/1e3/steps:6.3f} ms/step")
