"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/rocprof_run.sh into profiles/r01_traffic.json.

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of coalesced streaming reads, so it is doubled (checked here: the encode kernel must read 4 B per channel-sample
and the doubled counter lands within a few percent of that); WRITE_SIZE matches byte counts of the decode kernels' known
230 MB outputs one to one and is taken as is.
usage: python tools/rocprof_traffic.py gpurun_out/prof_<tag> <blocks> <level>
"""
import csv, glob, json, os, sys

root, blocks, level = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
KERNEL = 'fg_encode_fast_kernel'


def mean_counter(sub, name):
    vals = []
    for f in glob.glob(os.path.join(root, sub, '*counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            if KERNEL in row.get('Kernel_Name', '') and row.get('Counter_Name') == name:
                vals.append(float(row['Counter_Value']))
    return sum(vals) / len(vals) if vals else None


fetch, write = mean_counter('pmc3', 'FETCH_SIZE'), mean_counter('pmc4', 'WRITE_SIZE')
if fetch is None or write is None:
    raise SystemExit('counters not found under ' + root)
out = {'kernel': KERNEL, 'blocks': blocks, 'level': level,
       'FETCH_SIZE_KiB_raw': round(fetch, 1), 'WRITE_SIZE_KiB_raw': round(write, 1), 'fetch_correction': 2.0,
       'traffic_bytes_per_launch': int(fetch * 1024 * 2.0 + write * 1024),
       'source': os.path.basename(root.rstrip('/')) + ' (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)'}
os.makedirs('profiles', exist_ok=True)
json.dump(out, open('profiles/r01_traffic.json', 'w'), indent=1)
print(out)
