"""cProfile of the wrappers' step loop: python tools/exp/step_profile.py"""
import os, sys, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyflac_amd import batch, synth
pcm = synth.config2_stereo16(600.0, 3)
t = torch.from_numpy(pcm.astype(np.int32)).cuda()
a = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096)
o, f, st = a.encode(s, t)
n = t.shape[0]
dec = None
def loop(k):
    global o, f, dec
    for _ in range(k):
        o, f, st = a.encode(s, t, out=o, offsets=f)
        dec, status, ds = a.decode_stream(o[:st.total_bytes], 2, 16, n, nframes=st.nblocks, out=dec)
loop(5)
pr = cProfile.Profile()
pr.enable(); loop(300); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
