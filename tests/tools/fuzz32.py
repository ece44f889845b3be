"""gpu_fuzz over the 32-bit cases of the corpus only (the wide forms of the pipeline): usage python tests/tools/fuzz32.py first count"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import oracle as O
from pyflac_amd import batch
from pyflac_amd.encoder import stream_header_bytes
from tests import fuzzgen
first, count = int(sys.argv[1]), int(sys.argv[2])
ctx = batch.Context(0)
ran = bad = redo = blocks = 0
for seed in range(first, first + count):
    r = np.random.default_rng(7000 + seed)
    ch = int(r.choice([1, 2, 2, 2, 2, 3, 4, 6, 8])); bps = int(r.choice([8, 12, 16, 16, 16, 16, 20, 24, 24, 24, 32]))
    if bps != 32 or ch > 2:
        continue
    c = fuzzgen.case(seed)
    cfg, rc = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
    if rc:
        continue
    s = batch.settings(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
    cfg.limit_min_bitrate = 1 if c['limit_min_bitrate'] else 0
    s.limit_min_bitrate = cfg.limit_min_bitrate
    a32 = np.ascontiguousarray(c['pcm'].astype(np.int32))
    want, _ = O.encode_stream(cfg, a32)
    t = torch.from_numpy(a32).cuda()
    out, offs, st = ctx.encode(s, t)
    ran += 1; redo += st.redo_blocks; blocks += st.nblocks
    got = stream_header_bytes(s) + out[:st.total_bytes].cpu().numpy().tobytes()
    if got != want:
        k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
        h = offs.cpu().numpy()
        fr = int(np.searchsorted(h, k - 86, side='right') - 1) if k >= 86 else -1
        print('ENCODE DIFF seed %d ch%d l%d bs%d n%d %s lmb%d first diff byte %d frame %d' % (seed, c['ch'], c['level'], c['bs'], len(a32), c['kind'], c['limit_min_bitrate'], k, fr)); bad += 1
        continue
    dec, status, _ = ctx.decode(out[:st.total_bytes], offs, c['ch'], c['bps'], len(a32))
    if int(status[:, 0].max()) != 0 or not torch.equal(dec.reshape(-1, c['ch']), t):
        print('DECODE DIFF seed', seed); bad += 1
print('32-bit cases %d..%d: ran %d, %d bad; %d of %d blocks went to the generic kernel' % (first, first + count - 1, ran, bad, redo, blocks))
