"""Encode launch time of the headline workload over repeated calls (no checks): python tools/exp/direct_time.py [seconds] [reps] [direct 0/1]"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from pyflac_amd import batch, synth, _lib
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
direct = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pcm = synth.config2_stereo16(secs, 3)
import os
if os.environ.get('QUIET'):
    pcm = pcm >> int(os.environ['QUIET'])
t = torch.from_numpy(pcm.astype(np.int32)).cuda()
a = batch.Context(0)
_lib.lib().flacgpu_set_direct(a._h, direct)
s = batch.settings(5, 2, 16, 48000, 4096)
o = f = None
ms = []
for r in range(reps):
    o, f, st = a.encode(s, t, out=o, offsets=f)
    ms.append(st.total_gpu_ms)
ms = sorted(ms[3:])
print('direct %d path %d  encode gpu ms: min %.3f median %.3f  bytes/frame %.0f' % (direct, st.direct_path, ms[0], ms[len(ms) // 2], st.total_bytes / st.nblocks))
