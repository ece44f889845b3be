"""Differential fuzz: oracle/flac_oracle.c against the reference's libFLAC 1.4.3 binary on tests/fuzzgen.py cases.
Build container only (needs /root/reference).  usage: python tests/tools/fuzz_oracle_vs_ref.py [first] [count]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import libflac_ref as R
from oracle import oracle as O
from tests import fuzzgen

args = [a for a in sys.argv[1:] if not a.startswith('--')]
first = int(args[0]) if len(args) > 0 else 0
count = int(args[1]) if len(args) > 1 else 200
bad = 0
for seed in range(first, first + count):
    c = fuzzgen.case(seed)
    arr = c['pcm'].astype(np.int16 if c['bps'] == 16 else np.int32)
    extra = []
    if not c['subset']:
        extra.append(('set_streamable_subset', 0))
    if c['limit_min_bitrate']:
        extra.append(('set_limit_min_bitrate', 1))
    cbs, info = R.encode(arr, c['sr'], bps=c['bps'], level=c['level'], blocksize=c['bs'], extra=extra or None)
    cfg, rc = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
    tag = 'seed %d ch%d bps%d l%d bs%d sr%d n%d %s%s' % (seed, c['ch'], c['bps'], c['level'], c['bs'], c['sr'], len(arr), c['kind'],
                                                     ' lmb' if c['limit_min_bitrate'] else '')
    if rc != info['init_status']:
        print('INIT STATUS', tag, rc, info['init_status']); bad += 1
        continue
    if rc:
        continue
    if cfg.do_mid_side and c['bps'] == 32 and '--no33' in sys.argv:
        continue
    cfg.limit_min_bitrate = 1 if c['limit_min_bitrate'] else 0
    ref = b''.join(x[0] for x in cbs)
    mine, _ = O.encode_stream(cfg, arr)
    if mine != ref:
        k = next((i for i in range(min(len(mine), len(ref))) if mine[i] != ref[i]), -1)
        print('ENCODE DIFF', tag, 'len', len(mine), len(ref), 'first diff byte', k); bad += 1
        continue
    out, res = O.decode_stream(ref)
    if res.n_errors or not np.array_equal(out, arr.astype(np.int32).reshape(out.shape)):
        print('DECODE DIFF', tag); bad += 1
print('cases %d..%d: %d bad' % (first, first + count - 1, bad))
