"""Where the host time of a batch call goes: the whole Python method, the ctypes call alone, the GPU time the library reports."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from pyflac_amd import batch, synth, _lib
ctx = batch.Context(0)
L = _lib.lib()
pcm = torch.from_numpy(synth.config2_stereo16(600.0, 0).astype(np.int32)).cuda()
s = batch.settings(5, 2, 16, 48000, 4096)
out, offs, est = ctx.encode(s, pcm)
descs = (_lib.StreamDesc * 1)()
descs[0].pcm_offset = 0; descs[0].nsamples = pcm.shape[0]; descs[0].first_frame = 0
st = _lib.EncodeStats()
N = 200
for name in ('method', 'ctypes'):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); g = 0.0
    for _ in range(N):
        if name == 'method':
            out, offs, est = ctx.encode(s, pcm, out=out, offsets=offs); g += est.total_gpu_ms
        else:
            L.flacgpu_encode_streams(ctx._h, C.byref(s), pcm.data_ptr(), 0, descs, 1, out.data_ptr(), out.numel(), offs.data_ptr(), C.byref(st)); g += st.total_gpu_ms
    dt = (time.perf_counter() - t0) / N
    print('encode via %-7s wall %.1f us, gpu %.1f us, host share %.1f us' % (name, dt * 1e6, g / N * 1e3, dt * 1e6 - g / N * 1e3))
