"""Differential fuzz of the drop-in stream API: pyflac_amd.StreamEncoder fed with randomly cut input (empty calls, single
samples, pieces of many blocks; all nine levels incl. the loose mid-side ones, whose decisions must carry across calls)
against the oracle's bytes, then pyflac_amd.StreamDecoder fed with the bytes in random pieces against the input.
usage: python tests/tools/gpu_api_fuzz.py [first] [count]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyflac_amd
from oracle import oracle as O
from tests import fuzzgen

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for seed in range(first, first + count):
    r = np.random.default_rng(700000 + seed)
    ch = int(r.choice([1, 2, 2, 2, 3]))
    bps = int(r.choice([16, 16, 32]))
    bs = int(r.choice([0, 576, 1000, 1152, 4096]))
    level = int(r.integers(0, 9))
    n = int(r.choice([r.integers(1, 3000), r.integers(1, 60000)]))
    pcm, _kind = fuzzgen._signal(r, n, ch, 16 if bps == 16 else 24)
    pcm = pcm.astype(np.int16 if bps == 16 else np.int32)
    if bps == 32:
        pcm = pcm * 256
    tag = 'seed %d ch%d bps%d bs%d l%d n%d' % (seed, ch, bps, bs, level, n)
    verify = bool(r.random() < 0.3)
    lmb = bool(r.random() < 0.3)
    tag += ' verify' if verify else ''
    tag += ' lmb' if lmb else ''
    cfg, rc = O.config(level, ch, bps, 44100, bs, True)
    if rc:
        continue
    cfg.limit_min_bitrate = 1 if lmb else 0
    want, _ = O.encode_stream(cfg, pcm.astype(np.int32))
    cuts = sorted(set(int(x) for x in r.integers(0, n + 1, int(r.integers(0, 25)))))
    cuts = [0] + cuts + [n]
    if r.random() < 0.4:
        cuts = cuts[:1] + [cuts[1]] * 2 + cuts[1:]
    chunks = []
    enc = pyflac_amd.StreamEncoder(44100, lambda b, nb, s, f: chunks.append(b), compression_level=level, blocksize=bs, verify=verify,
                                   limit_min_bitrate=lmb)
    for a, b in zip(cuts[:-1], cuts[1:]):
        enc.process(pcm[a:b])
    if not enc.finish() or b''.join(chunks) != want:
        print('ENCODE DIFF', tag, len(b''.join(chunks)), len(want)); bad += 1
        continue
    blocks = []
    dec = pyflac_amd.StreamDecoder(lambda a, sr, c, nn: blocks.append(a))
    m = len(want)
    dcuts = [0] + sorted(set(int(x) for x in r.integers(0, m + 1, int(r.integers(0, 12))))) + [m]
    for a, b in zip(dcuts[:-1], dcuts[1:]):
        if b > a:
            dec.process(want[a:b])
    dec.finish()
    got = np.concatenate(blocks) if blocks else np.zeros((0, ch), pcm.dtype)
    if got.shape != pcm.reshape(-1, ch).shape or not np.array_equal(got, pcm.reshape(-1, ch)):
        print('DECODE DIFF', tag, got.shape, pcm.shape); bad += 1
print('api fuzz %d..%d: %d bad' % (first, first + count - 1, bad))
