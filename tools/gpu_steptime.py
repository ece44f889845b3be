import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pyflac_amd import batch, synth
ctx = batch.Context(0)
pcm16 = synth.config2_stereo16(600.0, 0, 48000)
pcm = torch.from_numpy(pcm16.astype(np.int32)).cuda()
s = batch.settings(5, 2, 16, 48000, 4096, True)
out = offs = dec = None
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, offs, est = ctx.encode(s, pcm, out=out, offsets=offs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    h = offs.cpu().numpy()
    t2 = time.perf_counter()
    dec, status, dst = ctx.decode(out, h, 2, 16, pcm.shape[0], out=dec)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print('encode call %.3f ms (kernel %.3f, gpu total %.3f) | offs copy %.3f | decode call %.3f ms (kernel %.3f, gpu total %.3f)' % (
        (t1 - t0) * 1e3, est.encode_kernel_ms, est.total_gpu_ms, (t2 - t1) * 1e3, (t3 - t2) * 1e3, dst.decode_kernel_ms, dst.total_gpu_ms))
