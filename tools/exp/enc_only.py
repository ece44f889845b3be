"""Encode launches of the headline stream only (device stamps): for A/B runs of builds whose output a decoder would refuse."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflac_amd import batch, synth
ctx = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096, True)
t = torch.from_numpy(synth.config2_stereo16(600.0, 0).astype(np.int32)).cuda()
o = f = None
for _ in range(5):
    o, f, st = ctx.encode(s, t, out=o, offsets=f)
g = 0.0
for _ in range(300):
    o, f, st = ctx.encode(s, t, out=o, offsets=f); g += st.total_gpu_ms
print('%s encode %.4f ms' % (os.environ.get('FLACGPU_LIBRARY', 'default'), g / 300))
