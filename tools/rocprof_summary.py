"""Turn a rocprofv3 (rocpd SQLite) result into the per-kernel stats CSV kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for name, calls, tot, avg, pct in rows:
        short = name if len(name) < 120 else name[:117] + "..."
        w.writerow([short, calls, f"{tot:.3f}", f"{avg:.3f}", f"{pct:.3f}"])
print(open(out).read()[:1500])
