"""Robustness fuzz of the host frame index (flacgpu_index_frames, no GPU needed): streams with small blocks -- where the
resynchronisation bound of a frame is a few hundred bytes and every kind of damage crosses it -- are flipped, cut, padded and
truncated at random; the index must return (a watchdog around every chunk of cases catches a search that never ends), its
offsets must rise strictly and stay inside the data.  usage: python tests/tools/index_damage_fuzz.py [first] [count] [--api]   (--api: the same damaged streams through the stream
decoder callbacks on the GPU box: must finish, and deliver silence or the clean samples)"""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def one(seed):
    from oracle import oracle as O
    from pyflac_amd import batch
    r = np.random.default_rng(910000 + seed)
    ch = int(r.choice([1, 2, 2]))
    bps = int(r.choice([8, 16, 16, 24]))
    bs = int(r.choice([16, 64, 192, 256, 576, 1000]))
    level = int(r.integers(0, 9))
    n = int(r.integers(bs, 12 * bs))
    amp = (1 << (bps - 2))
    t = np.arange(n)[:, None]
    pcm = np.round(amp * 0.5 * np.sin(t * r.uniform(0.01, 0.3, ch)) + r.normal(0, amp * 10 ** r.uniform(-3, -0.5), (n, ch))).astype(np.int32)
    pcm = np.clip(pcm, -(1 << (bps - 1)), (1 << (bps - 1)) - 1)
    cfg, rc = O.config(level, ch, bps, 44100, bs, False)
    if rc:
        return
    data, _ = O.encode_stream(cfg, pcm)
    d = bytearray(data)
    for _ in range(int(r.integers(1, 6))):
        kind = r.choice(['flip', 'flip', 'del', 'ins', 'trunc', 'zero', 'ff'])
        if len(d) <= 90:
            break
        pos = int(r.integers(86, len(d)))
        if kind == 'flip':
            d[pos] ^= int(r.integers(1, 256))
        elif kind == 'del':
            del d[pos:pos + int(r.integers(1, 800))]
        elif kind == 'ins':
            d[pos:pos] = r.integers(0, 256, int(r.integers(1, 4000)), dtype=np.uint8).tobytes()
        elif kind == 'zero':
            m = int(r.integers(1, 600))
            d[pos:pos + m] = bytes(min(m, len(d) - pos))
        elif kind == 'ff':
            m = int(r.integers(2, 300))
            d[pos:pos + m] = b'\\xff\\xf8' * (min(m, len(d) - pos) // 2)
        else:
            del d[pos:]
    if API:
        # the whole stream decoder on the damaged bytes (GPU box): it must finish and deliver nothing but silence or clean audio
        from tests import abi_decode
        got = abi_decode.decode(bytes(d), read_size=int(r.choice([8192, 1000, 100000])))
        assert got['state'] == 4, (seed, 'state', got['state'])
        for (sn, nb, _h), blk in zip(got['frames'], got['blocks']):
            ref = pcm[sn:sn + nb]
            assert not blk.any() or (len(ref) == nb and np.array_equal(blk.reshape(ref.shape), ref)), (seed, 'garbage audio', sn, nb)
        return
    offs, _si = batch.index_frames(bytes(d))
    o = [int(x) for x in offs]
    if len(o) <= 1:
        return                                  # (no frame survived)
    assert all(b > a for a, b in zip(o[:-1], o[1:])) and o[0] >= 42 and o[-1] <= len(d), (seed, o[:6], len(d))


API = '--api' in sys.argv
if API:
    sys.argv.remove('--api')

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--chunk':
        a, b = int(sys.argv[2]), int(sys.argv[3])
        for s in range(a, b):
            print('seed', s, flush=True)
            one(s)
        print('chunk done', flush=True)
        sys.exit(0)
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad = 0
    for a in range(first, first + count, 250):
        b = min(a + 250, first + count)
        try:
            p = subprocess.run([sys.executable, __file__, '--chunk', str(a), str(b)] + (['--api'] if API else []), capture_output=True, text=True, timeout=180)
            if 'chunk done' not in p.stdout:
                last = [ln for ln in p.stdout.splitlines() if ln.startswith('seed')][-1:]
                print('FAIL in chunk %d..%d at %s: %s' % (a, b - 1, last, p.stderr.strip().splitlines()[-1:] ))
                bad += 1
        except subprocess.TimeoutExpired as e:
            out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or '')
            last = [ln for ln in out.splitlines() if ln.startswith('seed')][-1:]
            print('HANG in chunk %d..%d at %s' % (a, b - 1, last))
            bad += 1
    print('index damage fuzz %d..%d: %d bad chunks' % (first, first + count - 1, bad))
