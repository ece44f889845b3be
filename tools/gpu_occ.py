"""Occupancy experiment: pad the fast encode kernel's dynamic LDS (FLACGPU_LDS_PAD) and watch the kernel time."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for pad in [int(x) for x in (sys.argv[1:] or ['0', '300', '600', '1200', '1300', '2600'])]:
    e = dict(os.environ); e['FLACGPU_LDS_PAD'] = str(pad)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--seconds', '600', '--steps', '4', '--warmup', '1',
                        '--no-cpu-baseline'], env=e, capture_output=True, text=True)
    j = json.loads(r.stdout.strip().splitlines()[-1])
    print('pad %6d  encode_kernel_ms %.3f' % (pad, j['encode_kernel_ms']))
