"""One fuzz seed through the pipeline with debug records on: the first stage record that differs from the oracle's (GPU box).
usage: python tests/tools/seed_records.py <seed>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle as O
from pyflac_amd import batch
from tests import fuzzgen
seed = int(sys.argv[1])
c = fuzzgen.case(seed)
print({k: v for k, v in c.items() if k != 'pcm'}, c['pcm'].shape)
s = batch.settings(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
if c['limit_min_bitrate']:
    s.limit_min_bitrate = 1
cfg, _ = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
if c['limit_min_bitrate']:
    cfg.limit_min_bitrate = 1
arr = c['pcm'].astype(np.int32).reshape(-1, c['ch'])
ctx = batch.Context(0)
out, offs, st = ctx.encode(s, torch.from_numpy(arr).cuda(), debug=True)
nb = (len(arr) + c['bs'] - 1) // c['bs']
recs = ctx.debug_records(0, nb)
ncand = 4 if c['ch'] == 2 else c['ch']
for b in range(nb):
    blk = arr[b * c['bs']:(b + 1) * c['bs']]
    _b, info = O.encode_frame(cfg, blk, b, want_info=True)
    for k in range(ncand):
        oc, gc = info.cand[k], recs[b].cand[k]
        if list(oc.fixed_tot) != list(gc.fixed_tot) or oc.fixed_guess != gc.fixed_guess or oc.fixed_bits != gc.fixed_bits:
            print('block', b, 'cand', k, 'fixed', list(oc.fixed_tot), list(gc.fixed_tot), oc.fixed_guess, gc.fixed_guess, oc.fixed_bits, gc.fixed_bits)
        for v in range(oc.n_vectors):
            a, g = list(oc.autoc[v][:cfg.max_lpc_order + 1]), list(gc.autoc[v][:cfg.max_lpc_order + 1])
            if a != g:
                print('block', b, 'cand', k, 'window', v, 'autoc differs at lags', [i for i in range(len(a)) if a[i] != g[i]], a[:3], g[:3])
            if oc.lpc_guess[v] != gc.lpc_guess[v] or oc.lpc_bits[v] != gc.lpc_bits[v]:
                print('block', b, 'cand', k, 'window', v, 'lpc guess/bits', oc.lpc_guess[v], gc.lpc_guess[v], oc.lpc_bits[v], gc.lpc_bits[v])
        o = (oc.type, oc.order, oc.precision, oc.shift, oc.porder, oc.rice_method, oc.bits, oc.wasted)
        g = (gc.type, gc.order, gc.precision, gc.shift, gc.porder, gc.rice_method, gc.bits, gc.wasted)
        if o != g:
            print('block', b, 'cand', k, 'decision (type, order, precision, shift, porder, method, bits, wasted)', o, g)
print('done')
