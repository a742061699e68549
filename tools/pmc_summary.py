"""Mean per-dispatch value of every counter in rocprofv3 --pmc CSV outputs, per kernel-name substring.
usage: python tools/pmc_summary.py <substring> <dir-or-csv> [...]"""
import collections, csv, glob, os, sys
key = sys.argv[1]
files = []
for a in sys.argv[2:]:
    files += glob.glob(os.path.join(a, '**', '*counter_collection.csv'), recursive=True) if os.path.isdir(a) else [a]
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if key in row['Kernel_Name']:
                tot[row['Counter_Name']] += float(row['Counter_Value']); cnt[row['Counter_Name']] += 1
for k in sorted(tot):
    print(f'{k},{tot[k] / cnt[k]:.1f},{cnt[k]}')
