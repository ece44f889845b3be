"""Round trips on launches of more than 8192 frames in odd shapes (tiny and odd block sizes, 8..32 bit, 1..4 channels):
tiled size scans and the 48-frame workgroups of the fused decoder.  GPU box: python tests/tools/gpu_odd_shapes.py"""
import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyflac_amd import batch, synth
ctx = batch.Context(0)
bad = 0
for (ch, bps, bs, nfr, lvl) in [(2, 16, 16, 50000, 5), (1, 8, 64, 20000, 3), (2, 24, 192, 12000, 8), (2, 16, 4096, 9000, 0), (4, 16, 128, 10000, 5), (2, 32, 256, 9000, 5), (2, 16, 17, 30000, 2), (1, 12, 100, 15000, 6)]:
    n = nfr * bs - bs // 2
    base = synth.config2_stereo16(n / 48000.0 + 0.01, 5)[:n].astype(np.int64)
    cols = [base[:, i % 2] * (1 if i < 2 else -1) + i for i in range(ch)]
    pcm = np.stack(cols, axis=1)
    if bps < 16: pcm = pcm >> (16 - bps)
    if bps > 16: pcm = pcm * (1 << (bps - 16 - (1 if bps == 32 else 0))) + (np.arange(n)[:, None] % 5)
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    pcm = np.clip(pcm, lo, hi).astype(np.int32)
    t = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    s = batch.settings(lvl, ch, bps, 48000, bs, streamable_subset=(bs >= 16))
    try:
        out, offs, st = ctx.encode(s, t)
        dec, status, _ = ctx.decode(out[:st.total_bytes], offs, ch, bps, n)
        ok = int(status[:, 0].max()) == 0 and torch.equal(dec.reshape(-1, ch), t) and st.nblocks == nfr
        ok2 = True
        if ch <= 2:
            dec2, status2, st2 = ctx.decode_stream(out[:st.total_bytes], ch, bps, n, nframes=nfr)
            ok2 = int(status2[:, 0].max()) == 0 and torch.equal(dec2.reshape(-1, ch), t)
        print((ch, bps, bs, nfr, lvl), 'ok' if ok and ok2 else 'BAD', ok, ok2, st.total_bytes)
        bad += not (ok and ok2)
    except Exception as e:
        print((ch, bps, bs, nfr, lvl), 'EXC', repr(e)[:200]); bad += 1
print('bad', bad)
