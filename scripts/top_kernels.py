"""Print the top kernels of a rocprofv3 sqlite result (``rocprofv3 --kernel-trace --stats -d DIR -o NAME`` writes
DIR/NAME_results.db): name, calls, total ms, average us, percent.
    python scripts/top_kernels.py gpurun_out/prof/NAME_results.db [N]"""
import sqlite3
import sys

if __name__ == "__main__":
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (n,)):
        print(f"{name[:90]:90s} {calls:7d} {total / 1e3:10.2f} ms {avg:10.1f} us {pct:6.2f} %")
