"""Timeline of the last dispatches of a rocprofv3 --kernel-trace CSV: start offset, duration and the idle gap before
each kernel (microseconds), to tell launch gaps from kernel time on launch-bound workloads."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
busy = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    busy += (e - s) / 1e3
    print(f"+{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  {r['Kernel_Name'].replace('void ', '')[:80]}")
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy:.1f} us")
