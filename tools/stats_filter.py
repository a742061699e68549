"""Print the rows of a rocprofv3 kernel_stats.csv whose kernel name contains one of the given substrings: python tools/stats_filter.py file.csv nhwc nchw igemm"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', round(tot / 1e6, 2))
for r in rows:
    n = r['Name']
    if any(k in n for k in sys.argv[2:]):
        print('%8.2f ms %5s %8.1f us  %s' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], float(r['AverageNs']) / 1e3, n[:120]))
