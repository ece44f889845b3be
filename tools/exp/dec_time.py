"""Decode launch time of the headline workload from the bytes alone over repeated calls (no checks):
python tools/exp/dec_time.py [seconds] [reps]"""
import sys
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyflac_amd import batch, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pcm = synth.config2_stereo16(secs, 3)
t = torch.from_numpy(pcm.astype(np.int32)).cuda()
a = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096)
o, f, st = a.encode(s, t)
data = o[:st.total_bytes].clone()
out = None
ms = []
for r in range(reps):
    dec, status, ds = a.decode_stream(data, 2, 16, t.shape[0], nframes=st.nblocks, out=out)
    out = dec
    ms.append(ds.total_gpu_ms)
ok = bool(torch.equal(dec.reshape(-1, 2), t))
ms = sorted(ms[3:])
print('decode gpu ms: min %.3f median %.3f  round trip %s  status max %d' % (ms[0], ms[len(ms) // 2], ok, int(status[:, 0].max())))
