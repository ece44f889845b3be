"""One fuzz case in detail: oracle frames vs GPU frames (first differing frame: header bytes, candidate records of the GPU).
usage: python tests/tools/fuzz_one.py <seed>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import oracle as O
from pyflac_amd import batch
from pyflac_amd.encoder import stream_header_bytes
from tests import fuzzgen
seed = int(sys.argv[1])
c = fuzzgen.case(seed)
print({k: v for k, v in c.items() if k != 'pcm'}, c['pcm'].shape, c['pcm'][:4].tolist())
cfg, rc = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
s = batch.settings(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
cfg.limit_min_bitrate = 1 if c['limit_min_bitrate'] else 0
s.limit_min_bitrate = cfg.limit_min_bitrate
a32 = np.ascontiguousarray(c['pcm'].astype(np.int32))
want, _ = O.encode_stream(cfg, a32)
ctx = batch.Context(0)
t = torch.from_numpy(a32).cuda()
for pipe in ('1', '0'):
    os.environ['FLACGPU_PIPE'] = pipe
    out, offs, st = ctx.encode(s, t, debug=True)
    got = stream_header_bytes(s) + out[:st.total_bytes].cpu().numpy().tobytes()
    h = offs.cpu().numpy()
    print('PIPE=%s' % pipe, 'equal' if got == want else 'DIFF', len(got), len(want))
    if got != want:
        k = next(i for i in range(min(len(got), len(want))) if got[i] != want[i])
        fr = int(np.searchsorted(h, k - 86, side='right') - 1)
        print(' first diff byte', k, 'frame', fr, 'got', got[86 + int(h[fr]):86 + int(h[fr]) + 12].hex(), 'want', want[k - (k - 86 - int(h[fr])):][:12].hex())
        rec = ctx.debug_records(fr, 1)[0]
        for ci in range(4):
            cd = rec.cand[ci]
            print('  cand', ci, 'wasted', cd.wasted, 'sbps', cd.sbps, 'type', cd.type, 'order', cd.order, 'bits', cd.bits, 'fixed_bits', cd.fixed_bits, 'guess', cd.fixed_guess)
