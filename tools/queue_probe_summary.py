"""Stream -> hardware queue map of a rocprofv3 kernel trace: python tools/queue_probe_summary.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    m = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        m[r["Stream_Id"]][r["Queue_Id"]] += 1
    print(f)
    for s in sorted(m, key=int):
        print(f"  stream {s}: " + ", ".join(f"queue {q} x{c}" for q, c in m[s].items()))
