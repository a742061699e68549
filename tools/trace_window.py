"""Aggregate a rocprofv3 kernel_trace.csv over the final `window_ms` of the trace (the timed steps
of bench.py), so that warm-up / MIOpen find kernels are excluded.

Per kernel name: summed duration AND the time it was the ONLY kernel on the device (`alone_ms`: what removing it can
save at most -- kernels of the parallel hipGraph branch that run under a convolution cost nothing on the wall clock)."""
import csv, sys, collections
path, window_ms, steps = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], (r.get('Queue_Id', ''), r.get('Stream_Id', ''))))
t_end = max(r[1] for r in rows)
cut = t_end - window_ms * 1e6
rows = [r for r in rows if r[0] >= cut]
agg = collections.defaultdict(lambda: [0, 0, 0])
ev = []
by_q = collections.defaultdict(lambda: [0, 0])
for i, (s, e, n, q) in enumerate(rows):
    by_q[q][0] += e - s; by_q[q][1] += 1
    agg[n][0] += e - s; agg[n][1] += 1
    ev.append((s, 1, i)); ev.append((e, 0, i))
ev.sort()
active, last, union, alone_tot, idle = set(), None, 0, 0, 0
for t, kind, i in ev:
    if last is not None and t > last:
        if len(active) == 1:
            agg[rows[next(iter(active))][2]][2] += t - last; alone_tot += t - last
        if active: union += t - last
    if kind: active.add(i)
    else: active.discard(i)
    last = t
tot = sum(v[0] for v in agg.values())
span = max(r[1] for r in rows) - min(r[0] for r in rows)
print(f'# window {window_ms:.1f} ms, {steps} steps, kernel time summed {tot/1e6:.2f} ms ({100*tot/(window_ms*1e6):.1f}% of window), {sum(v[1] for v in agg.values())} launches;'
      f' device busy (union) {union/1e6:.2f} ms, exactly one kernel running {alone_tot/1e6:.2f} ms, idle inside the span {(span-union)/1e6:.2f} ms')
print('# per (queue, stream): ' + '; '.join(f'{q}: {d/1e6/steps:.2f} ms/step in {c/steps:.0f} launches' for q, (d, c) in sorted(by_q.items(), key=lambda kv: -kv[1][0])))
print('pct,ms_per_step,alone_ms_per_step,calls_per_step,avg_us,kernel')
for n, (d, c, a) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f'{100*d/tot:.2f},{d/1e6/steps:.3f},{a/1e6/steps:.3f},{c/steps:.1f},{d/c/1e3:.1f},"{n[:140]}"')
# kernels of every queue except the busiest one (the parallel hipGraph branches)
main = max(by_q.items(), key=lambda kv: kv[1][0])[0]
side = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_, q_ in rows:
    if q_ != main:
        side[n_][0] += e_ - s_; side[n_][1] += 1
print('# kernels outside the busiest queue: ms_per_step,calls_per_step,kernel')
for n_, (d, c) in sorted(side.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'#  {d/1e6/steps:.3f},{c/steps:.1f},"{n_[:110]}"')
