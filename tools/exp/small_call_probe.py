"""Where does a one-block encode call spend its time?  batch.Context.encode on 1, 2, 4, 16, 64 blocks: wall time per call, device
time between the first and the last kernel (stamps), and the stage times (events)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflac_amd import batch, synth, _lib
L = _lib.lib()
dev = torch.device('cuda', 0)
ctx = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096, True)
pcm16 = synth.config2_stereo16(10.0, 0, 48000)
for nb in (1, 2, 4, 16, 64, 256):
    t = torch.from_numpy(pcm16[:4096 * nb].astype(np.int16)).to(dev)
    o = f = None
    for _ in range(20):
        o, f, st = ctx.encode(s, t, out=o, offsets=f)
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter(); g = 0.0
    for _ in range(K):
        o, f, st = ctx.encode(s, t, out=o, offsets=f); g += st.total_gpu_ms
    wall = (time.perf_counter() - t0) / K * 1e3
    L.flacgpu_set_stage_timing(ctx._h, 2)
    stg = np.zeros(4); ek = 0.0
    for _ in range(50):
        o, f, st = ctx.encode(s, t, out=o, offsets=f); stg += np.array(list(st.stage_ms)[:4]); ek += st.encode_kernel_ms
    L.flacgpu_set_stage_timing(ctx._h, 0)
    print('blocks %4d: wall %.3f ms per call, device first-to-last kernel %.3f ms; stages (events) analysis %.3f pack %.3f scan %.3f assemble %.3f' %
          (nb, wall, g / K, *(stg / 50)), flush=True)
