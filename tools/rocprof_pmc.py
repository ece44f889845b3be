"""Turn the PMC passes of tools/rocprof_run.sh into profiles/<round>_pmc.json: HBM traffic and VALU instruction counts per
encode launch (all fg_pipe_* kernels + sizes/scan) and per decode launch (all fg_dec_* kernels).

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of coalesced streaming reads, so it is doubled (round 1 checked this against a kernel with a known
230 MB input); WRITE_SIZE is taken as is.  Counter values are per dispatch; a "launch" is one dispatch of every kernel of
the group (the short-block packing kernel included).
usage: python tools/rocprof_pmc.py gpurun_out/prof_<tag> <workload> <blocks> <level> [round prefix, default r03]
"""
import csv, glob, json, os, sys
from collections import defaultdict

root, workload, blocks, level = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
RND = sys.argv[5] if len(sys.argv) > 5 else 'r05'


def per_kernel(sub, name):
    """Counter total per LAUNCH and kernel: since round 4 a launch may dispatch a kernel more than once (the encoder's two groups,
    pipe_shape.inc), so the dispatches of a pass are summed and divided by the number of launches in it -- the dispatch count of a
    kernel that runs exactly once per launch (the assembly kernel for encode launches, the header pass for decode launches)."""
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, '*counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') == name:
                k = row.get('Kernel_Name', '').split('(anonymous namespace)::')[-1].split('(')[0]
                acc[k].append(float(row['Counter_Value']))
    # (round 5: the direct packing path has no assembly kernel; every step of the bench is one encode and one decode launch, and the
    # decoder's header pass runs once a launch)
    # (later in round 5: the launches of a direction = the smallest dispatch count among its kernels that run in every launch -- the
    # passes are made with --no-passes, so every launch of a run is alike; a run may hold one decode launch more than encode launches,
    # the bench's check.  A dispatch far above its kernel's median is replaced by the median: under the profiler's serialisation of
    # kernels the FIRST decode launch's restore kernel waits for its join word until the bound, 0.2 s of polling -- the call then
    # falls back to events, as designed, and the other launches are what a launch costs)
    is_enc = lambda k: k.startswith('fg_pipe_') or k.startswith('fg_scan_') or k.startswith('fg_encode')
    is_dec = lambda k: k.startswith('fg_dec')
    def launches_of(pred):
        n = [len(v) for k, v in acc.items() if pred(k) and len(v) >= 3]
        return min(n) if n else 0
    once_enc, once_dec = launches_of(is_enc), launches_of(is_dec)
    out = {}
    for k, v in acc.items():
        launches = once_enc if is_enc(k) else (once_dec if is_dec(k) else 0)
        if len(v) >= 3:
            med = sorted(v)[len(v) // 2]
            v = [med if (med > 0 and x > 5 * med) else x for x in v]
        out[k] = sum(v) / launches if (launches and len(v) >= 3) else sum(v) / len(v)
    return out


fetch, write, valu = per_kernel('pmc3', 'FETCH_SIZE'), per_kernel('pmc4', 'WRITE_SIZE'), per_kernel('pmc1', 'SQ_INSTS_VALU')
enc = lambda k: k.startswith('fg_pipe_') or k.startswith('fg_scan_sizes') or k.startswith('fg_encode')
dec = lambda k: k.startswith('fg_dec')
rows = {}
for k in sorted(set(fetch) | set(write) | set(valu)):
    if enc(k) or dec(k):
        rows[k] = {'FETCH_SIZE_KiB_raw': round(fetch.get(k, 0.0), 1), 'WRITE_SIZE_KiB_raw': round(write.get(k, 0.0), 1),
                   'traffic_bytes': int(fetch.get(k, 0.0) * 2048 + write.get(k, 0.0) * 1024), 'valu_insts': int(valu.get(k, 0))}
# the build the passes ran on: the collect run leaves the ids of the library on the GPU box in build_id.txt (argument 6): one line
# "build kernel host" (round 6; a file with one word is a round-5 build id)
ids = open(sys.argv[6]).read().split() if len(sys.argv) > 6 and os.path.exists(sys.argv[6]) else []
if not ids:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pyflac_amd import _lib
    ids = [_lib.lib().flacgpu_build_id().decode(), _lib.lib().flacgpu_kernel_id().decode(), _lib.lib().flacgpu_host_id().decode()]
bid = ids[0]
out = {'workload': workload, 'blocks': blocks, 'level': level, 'fetch_correction': 2.0, 'build_id': bid,
       'kernel_id': ids[1] if len(ids) > 1 else None, 'host_id': ids[2] if len(ids) > 2 else None,
       'encode_traffic_bytes_per_launch': sum(r['traffic_bytes'] for k, r in rows.items() if enc(k)),
       'decode_traffic_bytes_per_launch': sum(r['traffic_bytes'] for k, r in rows.items() if dec(k)),
       'encode_valu_insts_per_launch': sum(r['valu_insts'] for k, r in rows.items() if enc(k)),
       'decode_valu_insts_per_launch': sum(r['valu_insts'] for k, r in rows.items() if dec(k)),
       'kernels': rows,
       'source': os.path.basename(root.rstrip('/')) + ' (rocprofv3 --pmc, separate passes: SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE)'}
os.makedirs('profiles', exist_ok=True)
name = 'profiles/%s_pmc.json' % RND if workload == 'stream16' else 'profiles/%s_pmc_%s.json' % (RND, workload)
json.dump(out, open(name, 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'kernels'}))
