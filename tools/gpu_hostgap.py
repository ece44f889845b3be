"""Wall time of a batch encode / decode call against the GPU time the library measured for it (host overhead per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyflac_amd import batch, synth
ctx = batch.Context(0)
pcm = torch.from_numpy(synth.config2_stereo16(600.0, 0).astype(np.int32)).cuda()
s = batch.settings(5, 2, 16, 48000, 4096)
out = offs = dec = None
for _ in range(3):
    out, offs, est = ctx.encode(s, pcm, out=out, offsets=offs)
    dec, status, dst = ctx.decode(out, offs, 2, 16, pcm.shape[0], out=dec)
torch.cuda.synchronize()
N = 50
we = wd = ge = gd = 0.0
for _ in range(N):
    t0 = time.perf_counter()
    out, offs, est = ctx.encode(s, pcm, out=out, offsets=offs)
    t1 = time.perf_counter()
    dec, status, dst = ctx.decode(out, offs, 2, 16, pcm.shape[0], out=dec)
    t2 = time.perf_counter()
    we += t1 - t0; wd += t2 - t1; ge += est.total_gpu_ms; gd += dst.total_gpu_ms
print('encode: wall %.1f us, gpu (first event to last) %.1f us, host share %.1f us' % (we / N * 1e6, ge / N * 1e3, (we / N) * 1e6 - ge / N * 1e3))
print('decode: wall %.1f us, gpu %.1f us, host share %.1f us' % (wd / N * 1e6, gd / N * 1e3, (wd / N) * 1e6 - gd / N * 1e3))
