"""Prints the top rows of a rocprofv3 kernel_stats.csv: calls, average milliseconds, kernel name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{int(r['Calls']):5d} {float(r['AverageNs']) / 1e6:9.4f} ms  {r['Name'].replace('void ', '')[:120]}")
