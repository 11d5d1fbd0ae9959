"""Condense a rocprofv3 --kernel-trace csv: per-kernel count / total / avg, and the busy vs idle time of the traced window.
   python tools/trace_summary.py <kernel_trace.csv> [skip_first_n_rows_fraction]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]  # the timed half (after build + warm-up)
acc = collections.defaultdict(lambda: [0, 0])
busy = 0
last_end = None
gaps = 0
for r in rows:
    m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?)", r["Kernel_Name"])
    k = m.group(1) if m else r["Kernel_Name"][:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    acc[k][0] += 1; acc[k][1] += e - s
    busy += e - s
    if last_end is not None and s > last_end:
        gaps += s - last_end
    last_end = max(last_end or 0, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("window %.3f ms, kernels busy %.3f ms (%.0f %%), gaps between kernels %.3f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, gaps / 1e6))
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:22]:
    print("%-50s n=%5d total %8.3f ms avg %8.1f us" % (k, n, t / 1e6, t / n / 1e3))
