"""Random damage against the decoder (callback ABI): flips, deletions, insertions and truncation anywhere behind the
metadata of a fixture stream.  Checks, per case: the decoder finishes (END_OF_STREAM), every delivered frame is either
silence or exactly the clean samples at its own sample number, sample numbers never run backwards, and an error is
reported whenever something is missing.  usage: python tests/tools/gpu_damage_fuzz.py [first] [count]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import abi_decode, cases


def damaged(data, audio_off, r):
    d = bytearray(data)
    for _ in range(int(r.integers(1, 6))):
        kind = r.choice(['flip', 'flip', 'flip', 'del', 'ins', 'trunc', 'zero'])
        pos = int(r.integers(audio_off, max(audio_off + 1, len(d))))
        if pos >= len(d):
            continue                            # (an earlier truncation left nothing behind the metadata)
        if kind == 'flip':
            d[pos] ^= int(r.integers(1, 256))
        elif kind == 'del':
            del d[pos:pos + int(r.integers(1, 3000))]
        elif kind == 'ins':
            d[pos:pos] = r.integers(0, 256, int(r.integers(1, 3000)), dtype=np.uint8).tobytes()
        elif kind == 'zero':
            n = int(r.integers(1, 2000))
            d[pos:pos + n] = bytes(min(n, len(d) - pos))
        else:
            del d[pos:]
    return bytes(d)


def check(name, seed):
    with open(os.path.join(cases.GOLDEN, 'data', name + '.flac'), 'rb') as f:
        data = f.read()
    clean = abi_decode.decode(data)
    pcm = np.concatenate(clean['blocks'])
    audio_off = {'stereo': 8304, 'mono': 8304, 'surround': 8348, '32bit': 8348}[name]
    r = np.random.default_rng(90000 + seed)
    bad = damaged(data, audio_off, r)
    got = abi_decode.decode(bad, read_size=int(r.choice([8192, 1000, 100000])))
    assert got['state'] == 4, ('state', got['state'])
    last_end = 0
    for (sn, bs, _h), blk in zip(got['frames'], got['blocks']):
        assert bs > 0 and sn >= last_end, ('order', sn, last_end)
        last_end = sn + bs
        assert sn + bs <= len(pcm) + 65535, ('beyond the stream', sn, bs)
        ref = pcm[sn:sn + bs]
        assert not blk.any() or (len(ref) == bs and np.array_equal(blk, ref)), ('garbage audio', sn, bs)
    delivered = sum(f[1] for f in got['frames'])
    if delivered < len(pcm) and bad[:len(data)] != data[:len(bad)]:
        pass                                    # (truncation alone ends silently, like libFLAC)
    return len(got['frames']), len(got['errors'])


if __name__ == '__main__':
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    nbad = 0
    for seed in range(first, first + count):
        name = ['stereo', 'mono', 'surround', '32bit'][seed % 4]
        try:
            check(name, seed)
        except AssertionError as e:
            nbad += 1
            print('FAIL seed %d %s: %s' % (seed, name, e))
    print('damage fuzz %d..%d: %d bad' % (first, first + count - 1, nbad))
