#!/usr/bin/env python3
"""Golden vectors for FLAC__stream_encoder_set_metadata: what the REFERENCE binary (pyFLAC's bundled libFLAC 1.4.3) writes
through the write callback during init_stream for the block lists of tests/metadata_build.py, and its init status.
Run in the build container:  python oracle/gen_golden_setmeta.py  ->  tests/golden/setmeta_vectors.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import libflac_ref as R     # noqa: E402
from tests import metadata_build as MB  # noqa: E402


def run(lib, blocks, new='FLAC__stream_encoder_new'):
    lib.FLAC__stream_encoder_new.restype = C.c_void_p
    enc = C.c_void_p(lib.FLAC__stream_encoder_new())
    out = []

    def _w(e, buf, n, samples, frame, cd):
        out.append(bytes(C.cast(buf, C.POINTER(C.c_ubyte * n)).contents) if n else b'')
        return 0
    wcb = R.ENC_WRITE_CB(_w)
    lib.FLAC__stream_encoder_set_channels(enc, 2)
    lib.FLAC__stream_encoder_set_bits_per_sample(enc, 16)
    lib.FLAC__stream_encoder_set_sample_rate(enc, 44100)
    lib.FLAC__stream_encoder_set_compression_level(enc, 5)
    arr = MB.block_array(blocks)
    lib.FLAC__stream_encoder_set_metadata.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    ok = lib.FLAC__stream_encoder_set_metadata(enc, arr, len(blocks))
    lib.FLAC__stream_encoder_init_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.FLAC__stream_encoder_init_stream(enc, wcb, None, None, None, None)
    lib.FLAC__stream_encoder_finish.argtypes = [C.c_void_p]
    lib.FLAC__stream_encoder_finish(enc)
    lib.FLAC__stream_encoder_delete.argtypes = [C.c_void_p]
    lib.FLAC__stream_encoder_delete(enc)
    return {'set_ok': int(bool(ok)), 'init_status': int(rc), 'writes': [b.hex() for b in out]}


if __name__ == '__main__':
    res = {name: run(R.lib(), blocks) for name, blocks in MB.cases().items()}
    with open(os.path.join(ROOT, 'tests', 'golden', 'setmeta_vectors.json'), 'w') as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        print('%-24s set %d init %2d  %d writes, %d bytes' % (k, v['set_ok'], v['init_status'], len(v['writes']), sum(len(w) // 2 for w in v['writes'])))
    # CUESHEET / PICTURE legality: the init status only
    leg = {name: run(R.lib(), blocks)['init_status'] for name, blocks in MB.legality_cases().items()}
    with open(os.path.join(ROOT, 'tests', 'golden', 'legality_vectors.json'), 'w') as f:
        json.dump(leg, f, indent=1)
    for k, v in leg.items():
        print('%-36s init %2d' % (k, v))
