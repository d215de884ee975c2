import json, sys
d = json.load(open(sys.argv[1]))
print(d["ms_per_step"], "ms/step", d["value"], d["unit"])
for c in d["roofline"].get("kernel_classes", d["roofline"].get("all_gemm_classes", [])):
    print(c)
