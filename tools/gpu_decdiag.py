"""Decode diagnostics: encode synthetic PCM on the GPU, decode it, report where the PCM differs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyflac_amd import batch, synth

ctx = batch.Context(0)
for secs, ch, bs in ((1.0, 2, 4096), (0.35, 2, 4096), (0.2, 1, 4096), (1.0, 2, 1152), (3.0, 2, 4096)):
    pcm = synth.config2_stereo16(secs, 0)
    if ch == 1:
        pcm = pcm[:, :1].copy()
    t = torch.from_numpy(pcm.astype(np.int32)).cuda()
    s = batch.settings(5, ch, 16, 48000, bs)
    out, offs, st = ctx.encode(s, t)
    offs_h = offs.cpu().numpy()
    dec, status, dst = ctx.decode(out[:st.total_bytes], offs_h, ch, 16, len(pcm))
    d = dec.cpu().numpy().reshape(-1, ch)
    bad = np.nonzero((d != pcm).any(axis=1))[0]
    print('secs %.2f ch %d bs %d: frames %d, n %d, status max %d, mismatching rows %d' % (secs, ch, bs, len(offs_h) - 1, len(pcm), int(status[:, 0].max()), len(bad)))
    if len(bad):
        print('   first bad rows', bad[:8], 'last', bad[-3:], ' blocks', np.unique(bad // bs)[:10])
        i = bad[0]
        print('   got', d[i:i + 4].tolist(), 'want', pcm[i:i + 4].tolist())
