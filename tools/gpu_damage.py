"""Print what our decoder delivers for every damaged-stream case next to the golden reference record."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases, abi_decode
g = json.load(open(os.path.join(cases.GOLDEN, 'damage_vectors.json')))
for name in sorted(cases.DAMAGE_CASES):
    for rs in (8192, 1000):
        got = abi_decode.decode(cases.damaged_stream(name), rs)
        w = g[name]
        same = got['frames'] == w['frames']
        row = ' '.join('%d:%d%s' % (a[0] // 4096, a[1], '' if i < len(w['frames']) and a == w['frames'][i] else '*') for i, a in enumerate(got['frames']))
        print(name, rs, 'frames_same' if same else 'FRAMES_DIFF', got['errors'], 'want', w['errors'], got['state'], w['state'])
        if not same:
            print('   got ', row)
            print('   want', ' '.join('%d:%d' % (a[0] // 4096, a[1]) for a in w['frames']))
