import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyflac_amd import batch, synth
pcm16 = synth.config2_stereo16(600.0, 0, 48000)
pcm = torch.from_numpy(pcm16.astype(np.int32)).cuda()
ctx = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096, True)
out, offs, est = ctx.encode(s, pcm)
for _ in range(5):
    try:
        dec, status, dst = ctx.decode_stream(out[:est.total_bytes], 2, 16, pcm.shape[0], nframes=est.nblocks)
    except Exception as e:
        print(e)
