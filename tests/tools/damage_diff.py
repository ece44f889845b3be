"""What the stream decoder delivers for the damage cases that are not byte-identical to the reference's callback sequence."""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import cases, abi_decode
from oracle import oracle as O
gold = json.load(open(os.path.join(cases.GOLDEN, 'damage_vectors.json')))
for name in sorted(cases.DAMAGE_CASES):
    src = cases.DAMAGE_CASES[name][0]
    clean, _ = O.decode_stream(open(os.path.join(cases.GOLDEN, 'data', src + '.flac'), 'rb').read())
    ch = clean.shape[1] if clean.ndim > 1 else 1
    clean = clean.reshape(-1, ch)
    want = gold[name]
    for rs in (8192, 1000):
        got = abi_decode.decode(cases.damaged_stream(name), rs)
        if got['frames'] == want['frames'] and got['errors'] == want['errors']:
            continue
        def kind(sn, bs, h):
            if h == hashlib.sha256(np.zeros((bs, ch), np.int32).tobytes()).hexdigest()[:16]: return 'Z'
            if h == hashlib.sha256(np.ascontiguousarray(clean[sn:sn + bs]).astype(np.int32).tobytes()).hexdigest()[:16]: return 'ok'
            return '??'
        print(name, rs, cases.DAMAGE_CASES[name][1:] if len(cases.DAMAGE_CASES[name]) > 1 else '')
        print('  want errors', want['errors'], ' got errors', got['errors'])
        w = [(f[0] // 4096 if f[1] == 4096 else f[0], f[1], kind(*f)) for f in want['frames']]
        g = [(f[0] // 4096 if f[1] == 4096 else f[0], f[1], kind(*f)) for f in got['frames']]
        print('  want', [(a, c) for a, b, c in w if c != 'ok'], len(w))
        print('  got ', [(a, c) for a, b, c in g if c != 'ok'], len(g))
