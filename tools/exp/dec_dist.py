"""Per-launch durations of the decode kernels from a rocprofv3 kernel trace: python tools/exp/dec_dist.py <trace dir>"""
import csv, sys, glob, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith('fg_dec_index_kernel')]
for k, i0 in enumerate(starts):
    i1 = starts[k + 1] if k + 1 < len(starts) else len(rows)
    seg = [r for r in rows[i0:i1] if 'fg_dec' in r[2] or 'signal_kernel' in r[2]]
    t0 = seg[0][0]
    d = {}
    for a, b, n in seg:
        d.setdefault(n.split('<')[0], []).append(((a - t0) / 1e3, (b - t0) / 1e3))
    def se(n): return d.get(n, [(0, 0)])[0]
    end = max(b for a, b, n in seg if 'signal' in n or 'wrestore' in n)
    print('launch %2d: total %6.1f | index %5.1f resolve ..%5.1f | parse %5.1f..%5.1f (%5.1f) | crc %5.1f..%5.1f hdr %5.1f scan ..%5.1f | restore %5.1f..%5.1f (%5.1f)' % (
        k, (end - t0) / 1e3, se('fg_dec_index_kernel')[1], se('fg_dec_index_resolve_kernel')[1], se('fg_dec_wparse_kernel')[0], se('fg_dec_wparse_kernel')[1],
        se('fg_dec_wparse_kernel')[1] - se('fg_dec_wparse_kernel')[0], se('fg_dec_crc_kernel')[0], se('fg_dec_crc_kernel')[1], se('fg_dec_headers_kernel')[0],
        se('fg_dec_scan_kernel')[1], se('fg_dec_wrestore_kernel')[0], se('fg_dec_wrestore_kernel')[1], se('fg_dec_wrestore_kernel')[1] - se('fg_dec_wrestore_kernel')[0]))
