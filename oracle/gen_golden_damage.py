"""Generate tests/golden/damage_vectors.json: how the reference's libFLAC 1.4.3 binary decodes damaged streams.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden_damage
Each case of tests/cases.py DAMAGE_CASES names a committed stream and a list of edits (byte flips, deletions,
insertions, truncation).  The vector records what pyFLAC's callbacks (pyflac/decoder.py:257-313) would see from the
bundled library for the edited stream: the error-callback status sequence and, per delivered frame, its sample number,
block size and a hash of the samples.  The `__md5__` entry records what FLAC__stream_decoder_finish returns with MD5
checking enabled (tests/cases.py MD5_CASES).
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import libflac_ref as R  # noqa: E402
from tests import cases  # noqa: E402


def record(data, read_size):
    """What the callbacks see, in order: per delivered frame its sample number, block size and a hash of the samples; the
    error statuses; and `events`, the two interleaved ('f<sample number>' / 'e<status>')."""
    pcm, frames, st = R.decode(data, read_size=read_size)
    pos = 0
    fr = []
    for f in frames:
        blk = pcm[pos:pos + f['blocksize']]
        pos += f['blocksize']
        fr.append([int(f['sample_number']), int(f['blocksize']),
                   hashlib.sha256(np.ascontiguousarray(blk, np.int32).tobytes()).hexdigest()[:16]])
    return {'errors': [int(e) for e in st['errors']], 'frames': fr, 'state': int(st['state']), 'events': st['events']}


def main():
    out = {}
    for name in sorted(cases.DAMAGE_CASES):
        data = cases.damaged_stream(name)
        for rs in cases.DAMAGE_READ_SIZES:
            key = name if rs == 8192 else '%s@%d' % (name, rs)
            out[key] = record(data, rs)
            print('%-28s frames %3d errors %s' % (key, len(out[key]['frames']), out[key]['errors']))
    for seed in cases.DAMAGE_FUZZ_SEEDS:
        src, data, rs = cases.fuzz_damaged_stream(seed)
        out['fuzz%d' % seed] = record(data, rs)
        out['fuzz%d' % seed]['sha'] = hashlib.sha256(data).hexdigest()[:16]      # (the damaged stream itself, to pin the generator)
        print('%-28s %-8s read %5d frames %3d errors %s' % ('fuzz%d' % seed, src, rs, len(out['fuzz%d' % seed]['frames']), out['fuzz%d' % seed]['errors']))
    for seed in cases.SMALL_DAMAGE_SEEDS:
        data, rs = cases.small_damaged_stream(seed)
        out['small%d' % seed] = record(data, rs)
        out['small%d' % seed]['sha'] = hashlib.sha256(data).hexdigest()[:16]
        print('%-28s %4d B  read %5d frames %3d errors %s' % ('small%d' % seed, len(data), rs, len(out['small%d' % seed]['frames']), out['small%d' % seed]['errors']))
    md5 = {}
    for name in sorted(cases.MD5_CASES):
        _pcm, frames, st = R.decode(cases.md5_stream(name), md5_checking=cases.MD5_CASES[name][2])
        md5[name] = {'finish': st['finish'], 'frames': len(frames), 'errors': [int(e) for e in st['errors']]}
        print('%-28s %s' % (name, md5[name]))
    out['__md5__'] = md5
    with open(os.path.join(cases.GOLDEN, 'damage_vectors.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True, separators=(',', ':'))


if __name__ == '__main__':
    main()
