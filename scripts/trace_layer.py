"""Print the kernel sequence of one prompt-pass layer from a rocprofv3 kernel trace (csv), with start offsets and durations.
    python scripts/trace_layer.py gpurun_out/x/s_kernel_trace.csv [marker-substring] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "gemm_prompt_kernel"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
start = idx[-(len(idx) // 3)] if len(idx) >= 3 else idx[0]
seg = rows[max(start - 3, 0):start + count]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %8.1f us  %s" % (s / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
