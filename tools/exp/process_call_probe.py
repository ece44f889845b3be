"""A StreamEncoder.process() call of one block (pyflac/encoder.py:86-119): wall time per call against the GPU time of the launch
inside it (device stamps of the encode call), for 16-bit stereo at level 5 -- the split the small-call work of round 5 is judged on.
usage: python tools/exp/process_call_probe.py [frames per call]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pyflac_amd
from pyflac_amd import synth, _lib, batch
import torch
L = _lib.lib()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pcm = synth.config2_stereo16(20.0, 0, 48000)
for md5 in (1, 0):
    enc = pyflac_amd.StreamEncoder(48000, lambda b, n, s, f: None, compression_level=5, blocksize=4096)
    if not md5:
        L.FLAC__stream_encoder_set_do_md5(enc._encoder, 0)
    enc.process(pcm[:frames])
    lat = []
    for a in range(frames, len(pcm) - frames + 1, frames):
        t0 = time.perf_counter()
        enc.process(pcm[a:a + frames])
        lat.append(time.perf_counter() - t0)
    enc.finish()
    lat.sort()
    print('process(%d frames), MD5 %s: median %.3f ms  p10 %.3f  p90 %.3f  (%d calls) -> %.1f M samples/s' %
          (frames, 'on' if md5 else 'off', lat[len(lat) // 2] * 1e3, lat[len(lat) // 10] * 1e3, lat[len(lat) * 9 // 10] * 1e3, len(lat),
           frames * 2 / lat[len(lat) // 2] / 1e6))
# the GPU part of such a call: one block through the batch entry point, device time from the first to the last kernel
ctx = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096, True)
nb = max(1, frames // 4096)
t = torch.from_numpy(pcm[:4096 * nb].astype(np.int32)).cuda()
o = f = None
g = []
for _ in range(300):
    o, f, st = ctx.encode(s, t, out=o, offsets=f)
    g.append(st.total_gpu_ms)
g.sort()
t0 = time.perf_counter()
for _ in range(300):
    o, f, st = ctx.encode(s, t, out=o, offsets=f)
w = (time.perf_counter() - t0) / 300 * 1e3
print('flacgpu_encode_streams on %d block(s), PCM resident: GPU first-to-last kernel %.3f ms (median), wall %.3f ms per call' % (nb, g[len(g) // 2], w))
