"""Print the rows of a rocprofv3 kernel_stats.csv whose kernel name matches any of the given substrings."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r['Name'] for k in sys.argv[2:]):
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:8.1f}")
