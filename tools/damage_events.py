"""Print, for every damaged-stream vector of tests/golden/damage_vectors.json, whether the stream decoder's callback sequence equals
the recorded one (GPU box).  usage: python tools/damage_events.py [name ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases, abi_decode

G = json.load(open(os.path.join(cases.GOLDEN, 'damage_vectors.json')))
todo = []
for name in sorted(cases.DAMAGE_CASES):
    for rs in cases.DAMAGE_READ_SIZES:
        todo.append((name if rs == 8192 else '%s@%d' % (name, rs), cases.damaged_stream(name), rs))
for seed in cases.DAMAGE_FUZZ_SEEDS:
    _s, data, rs = cases.fuzz_damaged_stream(seed)
    todo.append(('fuzz%d' % seed, data, rs))
bad = 0
for key, data, rs in todo:
    if len(sys.argv) > 1 and key not in sys.argv[1:]:
        continue
    want = G[key]
    got = abi_decode.decode(data, rs)
    ok = got['events'] == want['events'] and got['frames'] == want['frames'] and got['state'] == want['state']
    bad += not ok
    if not ok:
        print('DIFF', key, rs)
        print('  want', want['events'])
        print('  got ', got['events'])
print('%d of %d differ' % (bad, len(todo)))
