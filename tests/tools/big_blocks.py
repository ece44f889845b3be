"""Blocks too large for LDS staging (the generic kernel reads the PCM in place): bytes against the oracle.  GPU box; uses the
oracle as checker, like tests/tools/."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyflac_amd import batch
from oracle import oracle as O
rng = np.random.default_rng(3)
ctx = batch.Context(0)
for bps, bs, ch in [(32, 16384, 2), (24, 16384, 2), (32, 32768, 2), (16, 32768, 2), (32, 16384, 1), (24, 65535, 2)]:
    n = bs * 2 + 100
    t = np.arange(n)
    amp = (1 << (bps - 2)) * 0.6
    x = np.stack([np.round(amp * np.sin(0.01 * (c + 1) * t) + rng.normal(0, amp / 50, n)) for c in range(ch)], axis=1).astype(np.int64)
    x = np.clip(x, -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int32)
    for level in (5,):
        try:
            s = batch.settings(level, ch, bps, 48000, bs, False)
        except Exception as e:
            print(bps, bs, ch, 'settings', e); continue
        cfg, rc = O.config(level, ch, bps, 48000, bs, False)
        out, offs, st = ctx.encode(s, torch.from_numpy(x).cuda())
        want, sizes = O.encode_stream(cfg, x)
        got = out[:st.total_bytes].cpu().numpy().tobytes()
        print(bps, bs, ch, level, 'equal' if got == want[86:] else 'DIFFERENT', len(got), len(want) - 86)
