"""Encode / decode launch times of the batch workload (configs[4]'s share of one GPU) over repeated calls, no checks:
python tools/exp/batch_time.py [streams] [seconds] [reps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyflac_amd import batch, synth
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 128
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
one = synth.config2_stereo16(secs, 3).astype(np.int32)
pcm = np.concatenate([np.roll(one, 977 * i, axis=0) for i in range(streams)])
lengths = [one.shape[0]] * streams
t = torch.from_numpy(pcm).cuda()
a = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096)
o = f = None
em, dm = [], []
out = None
for r in range(reps):
    o, f, st = a.encode(s, t, stream_lengths=lengths, out=o, offsets=f)
    em.append(st.total_gpu_ms)
    dec, status, ds = a.decode(o[:st.total_bytes], f, 2, 16, t.shape[0], out=out)
    out = dec
    dm.append(ds.total_gpu_ms)
ok = bool(torch.equal(dec.reshape(-1, 2), t))
em, dm = sorted(em[2:]), sorted(dm[2:])
print('batch %d x %.0f s: encode gpu ms median %.3f, decode gpu ms median %.3f (with the index of the encoder), round trip %s' % (streams, secs, em[len(em) // 2], dm[len(dm) // 2], ok))
