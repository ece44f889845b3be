"""Where a StreamDecoder result differs from the input, for one seed of tests/tools/gpu_api_fuzz.py: python tests/tools/api_dec_diff.py <seed>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyflac_amd
from oracle import oracle as O
from tests import fuzzgen
seed = int(sys.argv[1])
r = np.random.default_rng(700000 + seed)
ch = int(r.choice([1, 2, 2, 2, 3])); bps = int(r.choice([16, 16, 32])); bs = int(r.choice([0, 576, 1000, 1152, 4096]))
level = int(r.integers(0, 9)); n = int(r.choice([r.integers(1, 3000), r.integers(1, 60000)]))
pcm, _kind = fuzzgen._signal(r, n, ch, 16 if bps == 16 else 24)
pcm = pcm.astype(np.int16 if bps == 16 else np.int32)
verify = bool(r.random() < 0.3); lmb = bool(r.random() < 0.3)
cfg, rc = O.config(level, ch, bps, 44100, bs, True)
cfg.limit_min_bitrate = 1 if lmb else 0
want, sizes = O.encode_stream(cfg, pcm.astype(np.int32))
print('kind', _kind, 'ch', ch, 'bps', bps, 'bs', bs, 'level', level, 'n', n, 'frames', len(sizes), 'blocksize', cfg.blocksize)
blocks = []
dec = pyflac_amd.StreamDecoder(lambda a, sr, c, nn: blocks.append(a))
dec.process(want); dec.finish()
got = np.concatenate(blocks)
ref = pcm.reshape(-1, ch)
d = np.argwhere(got != ref)
print('differences', len(d))
if len(d):
    rows = np.unique(d[:, 0])
    print('first rows', rows[:20], 'last', rows[-5:])
    B = cfg.blocksize
    print('frames touched', np.unique(rows // B)[:40])
    print('offsets within frame', np.unique(rows % B)[:40])
    for rr in rows[:6]:
        print(rr, 'got', got[rr], 'want', ref[rr])
