"""One-block encode calls under rocprofv3 --kernel-trace: kernel durations and the gaps between them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflac_amd import batch, synth
dev = torch.device('cuda', 0)
ctx = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096, True)
t = torch.from_numpy(synth.config2_stereo16(1.0, 0, 48000)[:4096].astype(np.int16)).to(dev)
o = f = None
for _ in range(60):
    o, f, st = ctx.encode(s, t, out=o, offsets=f)
torch.cuda.synchronize()
