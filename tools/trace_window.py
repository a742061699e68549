"""Aggregate a rocprofv3 kernel_trace.csv over the final `window_ms` of the trace (the timed steps
of bench.py), so that warm-up / MIOpen find kernels are excluded."""
import csv, sys, collections
path, window_ms, steps = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
t_end = max(r[1] for r in rows)
cut = t_end - window_ms * 1e6
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if s >= cut:
        agg[n][0] += e - s; agg[n][1] += 1
tot = sum(v[0] for v in agg.values())
print(f'# window {window_ms:.1f} ms, {steps} steps, kernel busy {tot/1e6:.2f} ms ({100*tot/(window_ms*1e6):.1f}% of window), {sum(v[1] for v in agg.values())} launches')
print('pct,ms_per_step,calls_per_step,avg_us,kernel')
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f'{100*d/tot:.2f},{d/1e6/steps:.3f},{c/steps:.1f},{d/c/1e3:.1f},"{n[:140]}"')
