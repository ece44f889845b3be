"""Bring-up helper: run one small encode with the fast kernel stopped after each stage (FLACGPU_STOP)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from pyflac_amd import batch, synth
ctx = batch.Context(0)
pcm = synth.config2_stereo16(0.5, 0)
t = torch.from_numpy(pcm.astype(np.int32)).cuda()
s = batch.settings(5, 2, 16, 48000, 4096)
try:
    out, offs, st = ctx.encode(s, t, debug=False)
    torch.cuda.synchronize()
    print('ok', st.nblocks, st.total_bytes)
except Exception as e:
    print('exc', e)
''' % ROOT
for env in ({'FLACGPU_NO_FAST': '1'},) + tuple({'FLACGPU_STOP': str(k)} for k in range(1, 9)) + ({},):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, '-c', code], env=e, capture_output=True, text=True, timeout=120)
    print(env, '->', (r.stdout.strip().splitlines() or ['?'])[-1], '| rc', r.returncode, '|', (r.stderr.strip().splitlines() or [''])[-1][:150])
