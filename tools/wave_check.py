"""Decode the headline stream with the current decoder selection (FLACGPU_DEC_WAVE / FLACGPU_DEC_FUSED) and compare with the input."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyflac_amd import batch, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pcm16 = synth.config2_stereo16(secs, 0, 48000)
pcm = torch.from_numpy(pcm16.astype(np.int32)).cuda()
ctx = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096, True)
out, offs, est = ctx.encode(s, pcm)
stream = out[:est.total_bytes]
for r in range(reps):
    t0 = time.perf_counter()
    dec, status, dst = ctx.decode_stream(stream, 2, 16, pcm.shape[0], nframes=est.nblocks)
    dt = time.perf_counter() - t0
    if r == 0:
        ok = bool(torch.equal(dec, pcm))
        bad = int((status[:, 0] != 0).sum())
        print('equal', ok, 'bad frames', bad, 'nframes', dst.nframes)
        if not ok:
            d = (dec != pcm).any(dim=1).nonzero()
            print('first diff sample', int(d[0]), 'count', d.numel())
    print('wall %.3f ms  gpu %.3f ms' % (dt * 1e3, dst.total_gpu_ms))
