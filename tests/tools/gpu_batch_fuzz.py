"""Differential fuzz of the batch launch: several streams of random lengths (ragged tails, streams shorter than a block, blocks
too short for the pipeline) in ONE flacgpu_encode_streams call, against the oracle stream by stream, then decoded back through
the device-resident index.  Exercises the block ordering (pipeline blocks first, short blocks, generic blocks), the per-slot
chunk table, the size scan in both forms and the reuse of the block list between calls (every layout is encoded twice, and
layouts alternate).  usage: python tests/tools/gpu_batch_fuzz.py [first] [count]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import oracle as O
from pyflac_amd import batch, synth

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = batch.Context(0)
bad = 0
for seed in range(first, first + count):
    r = np.random.default_rng(500000 + seed)
    level = int(r.integers(0, 9))
    bps = int(r.choice([16, 16, 24]))
    ch = int(r.choice([1, 2, 2]))
    bs = int(r.choice([4096, 4096, 1152, 2304, 4608, 576, 1024]))
    ns = int(r.integers(1, 7))
    lengths = [int(r.choice([r.integers(1, 300), r.integers(bs - 3, bs + 3), r.integers(bs, 6 * bs), r.integers(2 * bs, 40 * bs)])) for _ in range(ns)]
    amp = 10 ** r.uniform(1, np.log10(2 ** (bps - 1) - 1))
    parts = []
    for n in lengths:
        t = np.arange(n)[:, None]
        x = amp * 0.5 * np.sin(t * r.uniform(0.001, 0.3, ch)) + r.normal(0, amp * 10 ** r.uniform(-4, -0.7), (n, ch))
        if r.random() < 0.2:
            x = np.round(x / 16) * 16
        parts.append(np.clip(np.round(x), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int32))
    pcm = np.ascontiguousarray(np.concatenate(parts))
    tag = 'seed %d l%d bps%d ch%d bs%d lengths %s' % (seed, level, bps, ch, bs, lengths)
    cfg, rc = O.config(level, ch, bps, 48000, bs, True)
    if rc:
        continue
    s = batch.settings(level, ch, bps, 48000, bs, True)
    want = b''
    pos = 0
    for n in lengths:
        w, _ = O.encode_stream(cfg, pcm[pos:pos + n])
        want += w[86:]
        pos += n
    t = torch.from_numpy(pcm).cuda()
    ok = True
    for rep in range(2):
        out, offs, st = ctx.encode(s, t, stream_lengths=lengths)
        got = out[:st.total_bytes].cpu().numpy().tobytes()
        if got != want:
            k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
            print('ENCODE DIFF', tag, 'rep', rep, 'len', len(got), len(want), 'first diff byte', k); bad += 1; ok = False
            break
    if not ok:
        continue
    dec, status, _ = ctx.decode(out[:st.total_bytes], offs, ch, bps, len(pcm))
    if int(status[:, 0].max()) != 0 or not torch.equal(dec.reshape(-1, ch)[:len(pcm)], t.reshape(-1, ch)):
        print('DECODE DIFF', tag, 'status', [int(x) for x in np.nonzero(status[:, 0])[0][:8]]); bad += 1
print('batch fuzz %d..%d: %d bad' % (first, first + count - 1, bad))
