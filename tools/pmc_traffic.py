"""HBM traffic per launch of the custom HIP kernels from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: TCC has 4 slots, they cost 3 + 2).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]

Corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports HALF the bytes of a wide coalesced
streaming read -- doubled here. WRITE_SIZE is taken as reported. Families are matched on the
kernel name; the result is bytes per launch averaged over the profiled launches."""
import collections
import csv
import json
import sys

FAMILIES = {'conv2d_wgrad': 'conv2d_wgrad', 'conv2d_igemm': ', 2, 2, true, ', 'conv3d_igemm': 'conv3d_igemm', 'conv3d_wgrad': 'conv3d_wgrad', 'bias_act': 'bias_act', 'upfirdn2d': 'upfirdn2d', 'filtered_lrelu_strip': 'filtered_lrelu_strip', 'filtered_lrelu_band': 'filtered_lrelu_band', 'filtered_lrelu_wave': 'filtered_lrelu_wave', 'filtered_lrelu_mfma': 'filtered_lrelu_mfma', 'filtered_lrelu': 'filtered_lrelu',
            'tapconv_epilogue': 'tapconv_', 'modconv_epilogue': 'epilogue_', 'modconv2d_layout': 'LayoutArgs'}     # first match wins


def per_family(path, counter):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get('Counter_Name') != counter:
                continue
            name = row['Kernel_Name']
            for fam, key in FAMILIES.items():
                if key in name:
                    tot[fam] += float(row['Counter_Value'])
                    cnt[fam] += 1
                    break
    return tot, cnt


def main():
    fetch, nf = per_family(sys.argv[1], 'FETCH_SIZE')
    write, nw = per_family(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for fam in sorted(set(fetch) | set(write)):
        rd = 2.0 * 1024.0 * fetch[fam] / max(nf[fam], 1)
        wr = 1024.0 * write[fam] / max(nw[fam], 1)
        out[fam] = int(rd + wr)
        out[fam + '_detail'] = dict(read_bytes=int(rd), write_bytes=int(wr), launches_profiled=nf[fam],
                                    note='FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported, KiB -> bytes')
    # the 16-bit fused filtered_lrelu launches of a step are shared between the row-band, wave-per-tile and four-wave kernels: one launch-weighted figure
    fused = [f for f in ('filtered_lrelu_strip', 'filtered_lrelu_band', 'filtered_lrelu_wave', 'filtered_lrelu_mfma') if nf[f] or nw[f]]
    if fused:
        n = sum(max(nf[f], nw[f]) for f in fused)
        rd = 2.0 * 1024.0 * sum(fetch[f] for f in fused) / n
        wr = 1024.0 * sum(write[f] for f in fused) / n
        out['filtered_lrelu_fused16'] = int(rd + wr)
        out['filtered_lrelu_fused16_detail'] = dict(read_bytes=int(rd), write_bytes=int(wr), launches_profiled=n, kernels=fused,
                                                    note='launch-weighted over the 16-bit fused kernels of the workload')
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 3:
        with open(sys.argv[3], 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
