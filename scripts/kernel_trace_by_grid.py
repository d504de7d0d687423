"""Per-launch-shape times from a rocprofv3 kernel trace: launches of one kernel with different grids (the four
projections of a decoder layer share one GEMM kernel) are listed separately.
    python scripts/kernel_trace_by_grid.py <..._kernel_trace.csv> [name filter] [skip first N launches per shape]"""
import collections
import csv
import sys

rows = collections.defaultdict(list)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if flt not in n:
        continue
    grid = (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    rows[(n[:60], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, grid), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v = v[skip:] or v
    v.sort()
    print(f"{len(v):6d} x  median {v[len(v) // 2]:8.2f} us  mean {sum(v) / len(v):8.2f}  grid {'x'.join(grid[:3])} / {grid[3]}  {n}")
