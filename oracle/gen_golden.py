"""Generate tests/golden/encode_vectors.json from the reference's libFLAC 1.4.3 binary.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden
For every case of tests/cases.py it records what pyFLAC's write callback
(pyflac/encoder.py:429-450) would see from the bundled library: callback count, byte total,
SHA-256 of the concatenated stream, every frame's size and the encoder decisions recovered by
decoding the output with the same library (channel assignment; per subframe type / order /
wasted bits / partition order).  Small streams are stored whole in tests/golden/small_streams.npz.
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import libflac_ref as R  # noqa: E402
from pyflac_amd import synth  # noqa: E402
from tests import cases  # noqa: E402


def main():
    out = {}
    small = {}
    for name, (spec, sr, level, bs, subset) in sorted(cases.ENCODE_CASES.items()):
        pcm, bps = cases.make_pcm(spec)
        arr = cases.as_int_array(pcm, bps)
        extra = None if subset else [('set_streamable_subset', 0)]
        cbs, info = R.encode(arr, sr, bps=bps, level=level, blocksize=bs, extra=extra)
        stream = b''.join(c[0] for c in cbs)
        _pcm, frames, st = R.decode(stream)
        assert not st['errors'] and (_pcm == np.asarray(pcm).reshape(_pcm.shape)).all(), name
        # file-mode (seekable) STREAMINFO for the same input
        cbs2, info2 = R.encode(arr, sr, bps=bps, level=level, blocksize=bs, extra=extra, seekable=True,
                               want_metadata=True)
        rec = {
            'sample_rate': sr, 'bps': bps, 'channels': int(arr.shape[1]) if arr.ndim > 1 else 1,
            'level': level, 'blocksize_arg': bs, 'blocksize': info['blocksize'], 'subset': subset,
            'pcm_hash': synth.pcm_hash(arr), 'n_callbacks': len(cbs), 'total_bytes': len(stream),
            'sha256': hashlib.sha256(stream).hexdigest(),
            'callbacks': [[len(c[0]), c[1], c[2]] for c in cbs],
            'file_sha256': hashlib.sha256(info2['file']).hexdigest(),
            'streaminfo_file': info2['file'][8:42].hex(),
            'frames': [{'ca': f['channel_assignment'],
                        'sub': [[s['type'], s['wasted'], s.get('order', 0), s.get('porder', 0)]
                                for s in f['subframes']]} for f in frames],
        }
        out[name] = rec
        if len(stream) < 70000:
            small[name] = np.frombuffer(stream, np.uint8)
        print('%-24s %8d bytes %4d cbs  %s' % (name, len(stream), len(cbs), rec['sha256'][:16]))
    # limit_min_bitrate cases (tests/cases.py LIMIT_CASES)
    lim = {}
    for name, (spec, sr, level, bs) in sorted(cases.LIMIT_CASES.items()):
        pcm, bps = cases.make_pcm(spec)
        arr = cases.as_int_array(pcm, bps)
        cbs, info = R.encode(arr, sr, bps=bps, level=level, blocksize=bs, extra=[('set_limit_min_bitrate', 1)])
        stream = b''.join(c[0] for c in cbs)
        _pcm, frames, st = R.decode(stream)
        assert not st['errors'] and (_pcm == np.asarray(pcm).reshape(_pcm.shape)).all(), name
        lim[name] = {'total_bytes': len(stream), 'sha256': hashlib.sha256(stream).hexdigest(),
                     'frame_bytes': [len(c[0]) for c in cbs[3:]], 'pcm_hash': synth.pcm_hash(arr)}
        print('%-24s %8d bytes  %s' % (name, len(stream), lim[name]['sha256'][:16]))
    # seeded random corpus (tests/fuzzgen.py): settings, sample formats and signal families well outside the named cases,
    # incl. 32-bit stereo (33-bit side channel) and ragged blocks above 16 bit (the binary's AVX2 fixed-predictor sums)
    from tests import fuzzgen
    fz = {}
    for seed in range(fuzzgen.GOLDEN_SEEDS):
        c = fuzzgen.case(seed)
        arr = c['pcm'].astype(np.int16 if c['bps'] == 16 else np.int32)
        extra = []
        if not c['subset']:
            extra.append(('set_streamable_subset', 0))
        if c['limit_min_bitrate']:
            extra.append(('set_limit_min_bitrate', 1))
        cbs, info = R.encode(arr, c['sr'], bps=c['bps'], level=c['level'], blocksize=c['bs'], extra=extra or None)
        stream = b''.join(x[0] for x in cbs)
        fz[str(seed)] = {'init_status': int(info['init_status']), 'total_bytes': len(stream),
                         'sha256': hashlib.sha256(stream).hexdigest() if not info['init_status'] else '',
                         'pcm_hash': synth.pcm_hash(c['pcm'].astype(np.int32))}
    with open(os.path.join(cases.GOLDEN, 'fuzz_vectors.json'), 'w') as f:
        json.dump(fz, f, indent=0, sort_keys=True, separators=(',', ':'))
    print('fuzz corpus: %d cases' % len(fz))
    with open(os.path.join(cases.GOLDEN, 'limit_vectors.json'), 'w') as f:
        json.dump(lim, f, indent=0, sort_keys=True, separators=(',', ':'))
    gd = cases.GOLDEN
    with open(os.path.join(gd, 'encode_vectors.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True, separators=(',', ':'))
    np.savez_compressed(os.path.join(gd, 'small_streams.npz'), **small)
    # window tables: glibc cosf is accurate but not guaranteed correctly rounded (hard part 1)
    from oracle import oracle as O
    win = {}
    for lvl, n in ((5, 4096), (6, 4096), (8, 4096), (5, 1152), (5, 1000), (8, 4608)):
        cfg, _ = O.config(lvl, 2, 16, 48000, 4096)
        w = O.window(cfg, n)
        win['l%d_n%d' % (lvl, n)] = hashlib.sha256(w.tobytes()).hexdigest()
    with open(os.path.join(gd, 'window_hashes.json'), 'w') as f:
        json.dump(win, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
