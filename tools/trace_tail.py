"""Prints the last dispatches of a rocprofv3 --kernel-trace CSV in launch order (name, duration in ms)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-n:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"{d:9.3f} ms  grid={r.get('Grid_Size_X','?'):>10}  {r['Kernel_Name'][:90]}")
