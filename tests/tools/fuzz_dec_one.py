"""One fuzz case, decode side: the oracle's stream through the GPU decoder (fused and two-kernel form), per-frame status.
usage: python tests/tools/fuzz_dec_one.py <seed>"""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 2:
    os.environ['FLACGPU_DEC_FUSED'] = sys.argv[2]
import torch
from oracle import oracle as O
from pyflac_amd import batch
from tests import fuzzgen
seed = int(sys.argv[1])
c = fuzzgen.case(seed)
cfg, rc = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
cfg.limit_min_bitrate = 1 if c['limit_min_bitrate'] else 0
a32 = np.ascontiguousarray(c['pcm'].astype(np.int32))
want, _ = O.encode_stream(cfg, a32)
pcm, _res, offs = O.decode_stream(want, want_offsets=True)
offs = list(offs) + [len(want)]
print({k: v for k, v in c.items() if k != 'pcm'}, a32.shape, 'frames', len(offs) - 1, 'fused' if os.environ.get('FLACGPU_DEC_FUSED', '1') != '0' else 'two-kernel')
ctx = batch.Context(0)
body = torch.from_numpy(np.frombuffer(want[86:], np.uint8).copy()).cuda()
o = np.asarray(offs, np.uint64) - 86
dec, status, st = ctx.decode(body, o, c['ch'], c['bps'], len(a32))
print('status', status[:, 0].tolist(), 'equal', bool(torch.equal(dec.reshape(-1, c['ch'])[:len(a32)].cpu(), torch.from_numpy(a32))))
if len(sys.argv) <= 2:
    subprocess.run([sys.executable, __file__, sys.argv[1], '0'])
