"""Seeded parity cases shared by the golden generator (oracle/gen_golden.py) and the tests.

Every case is a deterministic PCM generator + encoder settings.  Expected outputs of the
reference's libFLAC 1.4.3 binary for these cases are committed in tests/golden/encode_vectors.json.
"""
import os

import numpy as np

from pyflac_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _rng(seed):
    return np.random.default_rng(seed)


def _sines(n, ch, amp, seed, noise):
    r = _rng(seed)
    t = np.arange(n)[:, None]
    f = 0.01 * (1 + np.arange(ch)) * (1 + 0.37 * np.arange(ch))
    return (amp * np.sin(t * f) + r.integers(-noise, noise + 1, (n, ch))).astype(np.int64)


def fixture_pcm(name):
    """PCM of a reference test fixture, decoded from tests/golden/data/<name>.flac by the oracle."""
    from oracle import oracle as O
    with open(os.path.join(GOLDEN, 'data', name + '.flac'), 'rb') as f:
        pcm, res = O.decode_stream(f.read())
    return pcm, res.sample_rate, res.bps


def make_pcm(spec):
    """spec: dict(kind=..., ...) -> (int array [frames, ch], bps)."""
    k = spec['kind']
    if k == 'cfg1':
        return synth.config1_sine(), 16
    if k == 'cfg2':
        return synth.config2_stereo16(spec['seconds'], spec.get('seed', 0)), 16
    if k == 'hard16':
        return synth.config2_hard16(spec['seconds'], spec.get('seed', 7)), 16
    if k == 'cfg4':
        return synth.config4_stereo24(spec['seconds'], spec.get('seed', 1)), 24
    if k == 'cfg5':
        return synth.config5_stream(spec['stream'], spec['seconds']), 16
    if k == 'fixture':
        pcm, _sr, bps = fixture_pcm(spec['name'])
        return pcm, bps
    if k == 'sines':
        bps = spec['bps']
        amp = (1 << (bps - 1)) - 1
        return _sines(spec['n'], spec['ch'], amp * spec.get('level', 0.7), spec.get('seed', 3),
                      max(amp // spec.get('snr', 30), 1)), bps
    if k == 'noise':
        bps = spec['bps']
        r = _rng(spec.get('seed', 5))
        return r.integers(-(1 << (bps - 1)), (1 << (bps - 1)), (spec['n'], spec['ch'])), bps
    if k == 'zeros':
        return np.zeros((spec['n'], spec['ch']), np.int64), spec['bps']
    if k == 'const':
        return np.full((spec['n'], spec['ch']), spec['value'], np.int64), spec['bps']
    if k == 'wasted':
        r = _rng(spec.get('seed', 9))
        return r.integers(-2000, 2000, (spec['n'], spec['ch'])) << spec['shift'], spec['bps']
    if k == 'lr_equal':
        r = _rng(11)
        a = r.integers(-20000, 20000, (spec['n'], 1))
        return np.concatenate([a, a * spec.get('sign', 1)], axis=1), 16
    if k == 'constmix':
        # per channel: 'z' zeros, 'c<value>' constant, 's' sine + noise, 'm' constant in even 4096-blocks only
        n, r = spec['n'], _rng(spec.get('seed', 21))
        cols = []
        for i, what in enumerate(spec['chans']):
            t = np.arange(n)
            if what == 'z': x = np.zeros(n)
            elif what[0] == 'c': x = np.full(n, int(what[1:]))
            elif what == 's': x = 2000 * np.sin(t * 0.03 * (i + 1)) + r.normal(0, 20, n)
            else: x = np.where((t // 4096) % 2 == 0, 77, 2000 * np.sin(t * 0.01))
            cols.append(np.round(x))
        return np.stack(cols, 1).astype(np.int64), 16
    if k == 'walk32':
        r = _rng(13)
        x = (np.cumsum(r.integers(-2 ** 27, 2 ** 27, spec['n'])) % 2 ** 31) | 1
        return x.reshape(-1, 1), 32
    raise KeyError(k)


def as_int_array(pcm, bps):
    """What pyFLAC hands to process(): int16 for 16-bit, else int32."""
    pcm = np.asarray(pcm)
    return pcm.astype(np.int16) if bps == 16 else pcm.astype(np.int32)


# name -> (pcm spec, sample_rate, level, blocksize, streamable_subset)
ENCODE_CASES = {
    'cfg1_passthrough': ({'kind': 'cfg1'}, 44100, 5, 0, True),
    'cfg2_2s_l5': ({'kind': 'cfg2', 'seconds': 2.0}, 48000, 5, 4096, True),
    'cfg2_20s_l5': ({'kind': 'cfg2', 'seconds': 20.0}, 48000, 5, 4096, True),
    'cfg4_1s_l8': ({'kind': 'cfg4', 'seconds': 1.0}, 96000, 8, 4096, True),
    'cfg4_10s_l8': ({'kind': 'cfg4', 'seconds': 10.0}, 96000, 8, 4096, True),
    'cfg4_1s_l5': ({'kind': 'cfg4', 'seconds': 1.0}, 96000, 5, 4096, True),
    'cfg5_s0_2s': ({'kind': 'cfg5', 'stream': 0, 'seconds': 2.0}, 48000, 5, 4096, True),
    'cfg5_s777_2s': ({'kind': 'cfg5', 'stream': 777, 'seconds': 2.0}, 48000, 5, 4096, True),
    'fixture_mono_l5': ({'kind': 'fixture', 'name': 'mono'}, 44100, 5, 0, True),
    'fixture_stereo_l5': ({'kind': 'fixture', 'name': 'stereo'}, 44100, 5, 0, True),
    'fixture_stereo_l8': ({'kind': 'fixture', 'name': 'stereo'}, 44100, 8, 0, True),
    'fixture_stereo_bs1024': ({'kind': 'fixture', 'name': 'stereo'}, 44100, 5, 1024, True),
    'fixture_surround_l5': ({'kind': 'fixture', 'name': 'surround'}, 48000, 5, 0, True),
    'fixture_32bit_l5': ({'kind': 'fixture', 'name': '32bit'}, 44100, 5, 0, True),
    'noise16_st': ({'kind': 'noise', 'bps': 16, 'n': 9000, 'ch': 2}, 48000, 5, 4096, True),
    'zeros16_mono': ({'kind': 'zeros', 'bps': 16, 'n': 5000, 'ch': 1}, 44100, 5, 0, True),
    'const16_st': ({'kind': 'const', 'bps': 16, 'n': 8292, 'ch': 2, 'value': -5}, 48000, 5, 4096, True),
    'wasted4_st': ({'kind': 'wasted', 'bps': 16, 'n': 8192, 'ch': 2, 'shift': 4}, 48000, 5, 4096, True),
    'lr_equal': ({'kind': 'lr_equal', 'n': 8192}, 48000, 5, 4096, True),
    'lr_opposite': ({'kind': 'lr_equal', 'n': 8192, 'sign': -1}, 48000, 5, 4096, True),
    'sines24_l8_bs4608': ({'kind': 'sines', 'bps': 24, 'n': 12000, 'ch': 2}, 96000, 8, 4608, True),
    'sines8_st': ({'kind': 'sines', 'bps': 8, 'n': 9000, 'ch': 2}, 22050, 8, 4096, True),
    'sines12_st': ({'kind': 'sines', 'bps': 12, 'n': 9000, 'ch': 2}, 32000, 8, 4096, True),
    'sines20_st': ({'kind': 'sines', 'bps': 20, 'n': 9000, 'ch': 2}, 96000, 8, 4096, True),
    'sines16_ch3': ({'kind': 'sines', 'bps': 16, 'n': 6000, 'ch': 3, 'level': 0.1}, 48000, 5, 4096, True),
    'sines16_ch8': ({'kind': 'sines', 'bps': 16, 'n': 6000, 'ch': 8, 'level': 0.1}, 48000, 5, 4096, True),
    'sines16_bs16': ({'kind': 'sines', 'bps': 16, 'n': 55, 'ch': 2, 'level': 0.1}, 48000, 5, 16, True),
    'sines16_bs1000': ({'kind': 'sines', 'bps': 16, 'n': 3007, 'ch': 2, 'level': 0.1}, 48000, 5, 1000, True),
    'sines16_bs65535_lax': ({'kind': 'sines', 'bps': 16, 'n': 66312, 'ch': 2, 'level': 0.1}, 48000, 8, 65535, False),
    'sines16_sr12345_lax': ({'kind': 'sines', 'bps': 16, 'n': 5000, 'ch': 1, 'level': 0.1}, 12345, 5, 1024, False),
    'sines16_tail3': ({'kind': 'sines', 'bps': 16, 'n': 4099, 'ch': 2, 'level': 0.1}, 48000, 5, 4096, True),
    'walk32_l8': ({'kind': 'walk32', 'n': 12288}, 44100, 8, 4096, True),
}
for _lv in range(9):
    ENCODE_CASES['hard16_l%d' % _lv] = ({'kind': 'hard16', 'seconds': 1.0}, 48000, _lv, 0, True)
    ENCODE_CASES['cfg2_1s_l%d' % _lv] = ({'kind': 'cfg2', 'seconds': 1.0, 'seed': 42}, 48000, _lv, 0, True)


# FLAC__stream_encoder_set_limit_min_bitrate(true): name -> (pcm spec, sample rate, level, blocksize).
# Goldens in tests/golden/limit_vectors.json (reference binary, oracle/gen_golden.py).
_N = 4096 * 5 + 300
LIMIT_CASES = {
    'lmb_zeros_mono_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['z']}, 48000, 5, 4096),
    'lmb_const_mono_l0': ({'kind': 'constmix', 'n': _N, 'chans': ['c-911']}, 48000, 0, 4096),
    'lmb_mixed_mono_l8': ({'kind': 'constmix', 'n': _N, 'chans': ['m']}, 48000, 8, 4096),
    'lmb_zeros_st_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['z', 'z']}, 48000, 5, 4096),
    'lmb_equal_st_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['c123', 'c123']}, 48000, 5, 4096),
    'lmb_equal_st_l1': ({'kind': 'constmix', 'n': _N, 'chans': ['c123', 'c123']}, 48000, 1, 4096),
    'lmb_equal_st_l0': ({'kind': 'constmix', 'n': _N, 'chans': ['c123', 'c123']}, 48000, 0, 4096),
    'lmb_diff_st_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['c123', 'c50']}, 48000, 5, 4096),
    'lmb_diff_st_l4': ({'kind': 'constmix', 'n': _N, 'chans': ['c7', 'c9']}, 48000, 4, 4096),
    'lmb_const_sine_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['c5', 's']}, 48000, 5, 4096),
    'lmb_sine_const_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['s', 'z']}, 48000, 5, 4096),
    'lmb_mixed_st_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['m', 'c77']}, 48000, 5, 4096),
    'lmb_mixed_st_l8': ({'kind': 'constmix', 'n': _N, 'chans': ['m', 'm']}, 48000, 8, 4096),
    'lmb_ch3_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['c1', 'c2', 'c3']}, 48000, 5, 4096),
    'lmb_ch3_mid_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['c1', 's', 'c3']}, 48000, 5, 4096),
    'lmb_ch6_l5': ({'kind': 'constmix', 'n': _N, 'chans': ['z', 'c4', 'z', 'c-4', 'z', 'm']}, 48000, 5, 4096),
    'lmb_bs1152_st_l2': ({'kind': 'constmix', 'n': 1152 * 7 + 5, 'chans': ['c-3', 'c-3']}, 44100, 2, 1152),
}


# ---------------------------------------------------------------------------------------------------------------------
# Damaged streams (decoder resynchronisation).  name -> (fixture stream under golden/data, edits applied in order).
# Edits use absolute byte positions of the ORIGINAL stream (they are applied back to front so they do not shift each
# other): ('flip', pos, xor) | ('del', pos, count) | ('ins', pos, count, seed) | ('trunc', length).
# stereo.flac frames start at 8304, 15616, 21301, 26619, 31110, 35345, 39273, 43054, ...; mono.flac at 8304, 10663,
# 12641, 14513, 15930, 17235, 18551, 19536, ...
DAMAGE_CASES = {
    'body_flip':            ('stereo', [('flip', 21301 + 900, 0x10)]),
    'body_flip_two_frames': ('stereo', [('flip', 15616 + 77, 0x01), ('flip', 35345 + 2000, 0x80)]),
    'body_flip_adjacent':   ('stereo', [('flip', 21301 + 900, 0x10), ('flip', 26619 + 50, 0x04)]),
    'crc16_flip':           ('stereo', [('flip', 26619 - 1, 0x01)]),
    'header_blocksize':     ('stereo', [('flip', 26619 + 2, 0x10)]),
    'header_sync':          ('stereo', [('flip', 26619, 0x0F)]),
    'header_crc8':          ('mono', [('flip', 12641 + 5, 0xFF)]),
    'delete_in_body':       ('stereo', [('del', 31110 + 1000, 37)]),
    'delete_header':        ('mono', [('del', 14513, 8)]),
    'insert_in_body':       ('stereo', [('ins', 35345 + 300, 501, 5)]),
    'insert_between':       ('mono', [('ins', 15930, 3000, 6)]),
    'truncated':            ('stereo', [('trunc', 43054 + 1234)]),
    'truncated_at_frame':   ('mono', [('trunc', 18551)]),
    'first_frame_body':     ('mono', [('flip', 8304 + 100, 0x20)]),
    'last_frame_body':      ('mono', [('flip', 20964 - 100, 0x02)]),
    'many_flips':           ('stereo', [('flip', p, 0x40) for p in range(9000, 73000, 9001)]),
}


def damaged_stream(name):
    src, edits = DAMAGE_CASES[name]
    with open(os.path.join(GOLDEN, 'data', src + '.flac'), 'rb') as f:
        data = bytearray(f.read())
    for e in sorted(edits, key=lambda e: -e[1]):
        if e[0] == 'flip':
            data[e[1]] ^= e[2]
        elif e[0] == 'del':
            del data[e[1]:e[1] + e[2]]
        elif e[0] == 'ins':
            data[e[1]:e[1]] = _rng(e[3]).integers(0, 256, e[2], dtype=np.uint8).tobytes()
        elif e[0] == 'trunc':
            del data[e[1]:]
    return bytes(data)


# Random damage: seed -> (fixture, damaged stream, size of the read callback's answers).  One to four edits (bit flips, deletions,
# insertions of random bytes, stretches of zeros) anywhere behind the first 8300 bytes, and a read size that puts the refills of
# libFLAC's 8 KiB reader in different places.
DAMAGE_FUZZ_SEEDS = list(range(1000, 1200)) + [1225, 1250, 1285, 1309, 5003, 5077, 5210, 5388]
DAMAGE_READ_SIZES = [8192, 1000]


def fuzz_damaged_stream(seed):
    r = np.random.default_rng(seed)
    name = ['stereo', 'mono', 'surround', '32bit'][int(r.integers(0, 4))]
    with open(os.path.join(GOLDEN, 'data', name + '.flac'), 'rb') as f:
        data = bytearray(f.read())
    for _ in range(int(r.integers(1, 5))):
        kind = int(r.integers(0, 4))
        p = int(r.integers(8300, len(data) - 10))
        if kind == 0:
            data[p] ^= 1 << int(r.integers(0, 8))
        elif kind == 1:
            del data[p:p + int(r.integers(1, 300))]
        elif kind == 2:
            data[p:p] = r.integers(0, 256, int(r.integers(1, 600)), dtype=np.uint8).tobytes()
        else:
            n = int(r.integers(1, 200))
            data[p:p + n] = bytes(n)
    rs = int(r.choice([8192, 1000, 4096, 333, 65536]))
    return name, bytes(data), rs


# Small streams: frames of 10..60 bytes read in answers of a few bytes to a few hundred, so that the refills of libFLAC's reader
# fall everywhere around the damaged frames (round 3's replay walked in a circle when a refill fell 2..7 bytes behind a damaged
# frame's sync code; tests/tools/refwalk_small_fuzz.py found 44 such seeds and 8 wrong sequences in its first 3000).  The
# streams come from the oracle's encoder; the vectors pin them by hash.
SMALL_DAMAGE_SEEDS = sorted(set(list(range(0, 152)) + [
    82, 95, 417, 433, 499, 501, 531, 570, 651, 721, 885, 909, 929, 966, 1012, 1066, 1114, 1142, 1152, 1174, 1255, 1274, 1275, 1285,
    1398, 1455, 1473, 1574, 1595, 1831, 1837, 1867, 1884, 1926, 2071, 2103, 2248, 2323, 2412, 2446, 2471, 2564, 2615, 2638, 2694,
    2754, 2798, 2991]))


def small_damaged_stream(seed):
    """(stream bytes, read size) of seed: a short stream of small frames with one to three edits behind the metadata."""
    from oracle import oracle as O
    r = np.random.default_rng(seed)
    ch = int(r.integers(1, 3))
    bps = int(r.choice([8, 16]))
    blk = int(r.choice([16, 16, 24, 32, 64, 192]))
    nfr = int(r.integers(2, 9))
    n = blk * nfr - int(r.integers(0, blk // 2))
    kind = int(r.integers(0, 3))
    amp = (1 << (bps - 1)) - 1
    if kind == 0:
        pcm = np.full((n, ch), int(r.integers(-amp, amp)), np.int32)          # constant frames: 11..14 bytes each
    elif kind == 1:
        pcm = r.integers(-3, 4, (n, ch)).astype(np.int32)
    else:
        pcm = r.integers(-amp, amp, (n, ch)).astype(np.int32)
    cfg, _ = O.config(int(r.integers(0, 9)), ch, bps, 44100, blk, subset=False)
    data, _sizes = O.encode_stream(cfg, pcm, finalize=bool(r.integers(0, 2)))
    data = bytearray(data)
    for _ in range(int(r.integers(1, 4))):
        k = int(r.integers(0, 5))
        p = int(r.integers(86, max(87, len(data) - 1)))
        if k == 0:
            data[p] ^= 1 << int(r.integers(0, 8))
        elif k == 1:
            del data[p:p + int(r.integers(1, 12))]
        elif k == 2:
            data[p:p] = r.integers(0, 256, int(r.integers(1, 20)), dtype=np.uint8).tobytes()
        elif k == 3:
            data[p:p + 2] = b'\xff\xf8'
        else:
            n0 = int(r.integers(1, 10))
            data[p:p + n0] = bytes(n0)
    rs = int(r.choice([int(r.integers(1, 130)), int(r.integers(1, 40)), 8192, int(r.integers(100, 1000))]))
    return bytes(data), rs


# MD5 checking on decode (FLAC__stream_decoder_set_md5_checking): name -> (damage case or fixture, STREAMINFO md5 edit,
# checking enabled).  The signature sits at bytes 26..41 of a stream whose STREAMINFO is the first block.
MD5_CASES = {
    'clean_checked':      ('stereo', None, True),
    'clean_mono_checked': ('mono', None, True),
    'clean_32bit':        ('32bit', None, True),
    'clean_surround':     ('surround', None, True),
    'tampered_checked':   ('stereo', 'flip', True),
    'tampered_unchecked': ('stereo', 'flip', False),
    'zero_signature':     ('stereo', 'zero', True),
    'damaged_checked':    ('body_flip', None, True),
    'truncated_checked':  ('truncated', None, True),
}


def md5_stream(name):
    src, edit, _on = MD5_CASES[name]
    if src in DAMAGE_CASES:
        data = bytearray(damaged_stream(src))
    else:
        with open(os.path.join(GOLDEN, 'data', src + '.flac'), 'rb') as f:
            data = bytearray(f.read())
    if edit == 'flip':
        data[30] ^= 0x55
    elif edit == 'zero':
        data[26:42] = bytes(16)
    return bytes(data)


# ---------------------------------------------------------------------------------------------------------------------
# Metadata pass-through on decode.  A stream with one block of every kind in front of a single small frame, and the
# filter set-ups to read it (and the fixtures) with.
def metadata_stream():
    import struct

    def block(t, body, last=False):
        return bytes([(0x80 if last else 0) | t]) + len(body).to_bytes(3, 'big') + body

    with open(os.path.join(GOLDEN, 'data', 'mono.flac'), 'rb') as f:
        mono = f.read()
    si = mono[8:42]
    frame = mono[8304:10663]                                  # first audio frame of the fixture
    vc = struct.pack('<I', 6) + b'vendor' + struct.pack('<I', 3) + b''.join(struct.pack('<I', len(c)) + c for c in
                                                                           (b'TITLE=probe', b'ARTIST=\xc3\xa9', b'EMPTY='))
    seek = b''.join(struct.pack('>QQH', *p) for p in ((0, 0, 4096), (4096, 2359, 4096), (0xFFFFFFFFFFFFFFFF, 0, 0)))
    cue_track = lambda off, num, isrc, flags, idx: (struct.pack('>QB', off, num) + isrc.ljust(12, b'\0') + bytes([flags]) + bytes(13) +
                                                    bytes([len(idx)]) + b''.join(struct.pack('>QB', o, n) + bytes(3) for o, n in idx))
    cue = (b'1234567890123'.ljust(128, b'\0') + struct.pack('>Q', 88200) + bytes([0x80]) + bytes(258) + bytes([2]) +
           cue_track(0, 1, b'USRC17607839', 0x40, [(0, 0), (588, 1)]) + cue_track(44100 * 60, 170, b'', 0x80, []))
    pic = (struct.pack('>I', 3) + struct.pack('>I', 9) + b'image/png' + struct.pack('>I', 5) + b'cover' +
           struct.pack('>IIII', 4, 3, 24, 0) + struct.pack('>I', 10) + bytes(range(10)))
    blocks = [block(0, si), block(2, b'riff' + b'payload-1'), block(1, bytes(37)), block(3, seek), block(4, vc),
              block(2, b'aiff'), block(5, cue), block(6, pic), block(42, b'unknown block body'), block(1, b'', last=True)]
    return b'fLaC' + b''.join(blocks) + frame


METADATA_SETUPS = {
    'default': [],
    'all': [('respond_all',)],
    'none': [('ignore_all',)],
    'seektable_only': [('ignore_all',), ('respond', 3)],
    'all_but_padding': [('respond_all',), ('ignore', 1)],
    'one_app': [('ignore_all',), ('respond_app', b'aiff')],
    'all_but_one_app': [('respond_all',), ('ignore_app', b'riff')],
    'vc_and_pic': [('respond', 4), ('respond', 6), ('ignore', 0)],
}
METADATA_STREAMS = ['handmade', 'stereo', 'surround', '32bit']


def metadata_input(name):
    if name == 'handmade':
        return metadata_stream()
    with open(os.path.join(GOLDEN, 'data', name + '.flac'), 'rb') as f:
        return f.read()
