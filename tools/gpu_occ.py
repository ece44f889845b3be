"""Occupancy experiment: pad the fast encode kernel's dynamic LDS (FLACGPU_LDS_PAD) and watch the kernel time."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for pad in (0, 4000, 9000, 17000, 30000, 57000):
    e = dict(os.environ); e['FLACGPU_LDS_PAD'] = str(pad)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--seconds', '600', '--steps', '4', '--warmup', '1',
                        '--no-cpu-baseline'], env=e, capture_output=True, text=True)
    j = json.loads(r.stdout.strip().splitlines()[-1])
    lds = 23600 + pad
    print('pad %6d  ~%2d waves/CU  encode_kernel_ms %.3f' % (pad, 163840 // lds, j['encode_kernel_ms']))
