"""Pin the CPU oracle (oracle/flac_oracle.c) to the committed libFLAC 1.4.3 golden vectors.

The vectors were produced by the reference's bundled binary (oracle/gen_golden.py); the
reference's own tests hold no encoded-byte assertions, so these
vectors plus the tests/data FLAC fixtures are the known answers for this path.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from pyflac_amd import synth
from tests import cases


@pytest.mark.parametrize('name', sorted(cases.ENCODE_CASES))
def test_encode_matches_golden(name, golden):
    spec, sr, level, bs, subset = cases.ENCODE_CASES[name]
    g = golden[name]
    pcm, bps = cases.make_pcm(spec)
    arr = cases.as_int_array(pcm, bps)
    assert synth.pcm_hash(arr) == g['pcm_hash']
    cfg, rc = O.config(level, g['channels'], bps, sr, bs, subset)
    assert rc == 0 and cfg.blocksize == g['blocksize']
    stream, sizes = O.encode_stream(cfg, arr)
    assert len(stream) == g['total_bytes']
    assert hashlib.sha256(stream).hexdigest() == g['sha256']
    assert [4, 38, 44] + [int(s) for s in sizes] == [c[0] for c in g['callbacks']]
    filed, _ = O.encode_stream(cfg, arr, finalize=True)
    assert hashlib.sha256(filed).hexdigest() == g['file_sha256']


@pytest.mark.parametrize('name', sorted(cases.LIMIT_CASES))
def test_limit_min_bitrate_matches_golden(name, limit_golden):
    """limit_min_bitrate: no frame of CONSTANT subframes only (behaviour recovered from the reference binary, including
    the loose mid-side frames; oracle/flac_oracle.c flo_encode_frame)."""
    spec, sr, level, bs = cases.LIMIT_CASES[name]
    g = limit_golden[name]
    pcm, bps = cases.make_pcm(spec)
    arr = cases.as_int_array(pcm, bps)
    assert synth.pcm_hash(arr) == g['pcm_hash']
    cfg, rc = O.config(level, arr.shape[1], bps, sr, bs, True)
    assert rc == 0
    cfg.limit_min_bitrate = 1
    stream, sizes = O.encode_stream(cfg, arr)
    assert [int(x) for x in sizes] == g['frame_bytes']
    assert hashlib.sha256(stream).hexdigest() == g['sha256']


@pytest.mark.parametrize('chunk', range(6))
def test_fuzz_corpus_matches_reference_hashes(chunk, fuzz_golden):
    """Seeded random corpus (tests/fuzzgen.py): 1-8 channels, 8-32 bit incl. 32-bit stereo with its 33-bit side channel,
    every level, odd block sizes and ragged tails, limit_min_bitrate, non-subset -- stream hashes recorded from the
    reference binary."""
    from tests import fuzzgen
    for seed in range(chunk * 40, chunk * 40 + 40):
        g = fuzz_golden[str(seed)]
        c = fuzzgen.case(seed)
        a32 = np.ascontiguousarray(c['pcm'].astype(np.int32))
        assert synth.pcm_hash(a32) == g['pcm_hash'], seed
        cfg, rc = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
        assert rc == g['init_status'], seed
        if rc:
            continue
        cfg.limit_min_bitrate = 1 if c['limit_min_bitrate'] else 0
        stream, _ = O.encode_stream(cfg, a32)
        assert len(stream) == g['total_bytes'] and hashlib.sha256(stream).hexdigest() == g['sha256'], seed
        out, res = O.decode_stream(stream)
        assert res.n_errors == 0 and np.array_equal(out, a32.reshape(out.shape)), seed


def test_frame_decisions_match_golden(golden):
    name = 'cfg2_2s_l5'
    spec, sr, level, bs, subset = cases.ENCODE_CASES[name]
    pcm, bps = cases.make_pcm(spec)
    arr = cases.as_int_array(pcm, bps).astype(np.int32)
    cfg, _ = O.config(level, 2, bps, sr, bs, subset)
    for i, fr in enumerate(golden[name]['frames']):
        blk = arr[i * cfg.blocksize:(i + 1) * cfg.blocksize]
        _b, info = O.encode_frame(cfg, blk, i, want_info=True)
        assert info.channel_assignment == fr['ca']
        pick = {0: (0, 1), 1: (0, 3), 2: (3, 1), 3: (2, 3)}[fr['ca']]
        names = ['CONSTANT', 'VERBATIM', 'FIXED', 'LPC']
        for c, sub in zip(pick, fr['sub']):
            ci = info.cand[c]
            assert [names[ci.type], ci.wasted, ci.order, ci.porder] == sub


def test_small_streams_decode(small_streams, golden):
    for name, data in small_streams.items():
        spec = cases.ENCODE_CASES[name][0]
        pcm, bps = cases.make_pcm(spec)
        out, res = O.decode_stream(data)
        assert res.n_errors == 0 and res.bps == bps
        assert np.array_equal(out, np.asarray(pcm).astype(np.int32).reshape(out.shape)), name


@pytest.mark.parametrize('name,frames,md5', [
    ('mono', 11, '37fc538d'), ('stereo', 17, 'de497c84'), ('surround', 7, 'c4b11fc1'),
    ('32bit', 11, 'ff470e8e')])
def test_reference_fixture_decode(name, frames, md5):
    """tests/data/*.flac are de-facto known-answer vectors: STREAMINFO MD5 == md5(PCM)."""
    with open(os.path.join(cases.GOLDEN, 'data', name + '.flac'), 'rb') as f:
        data = f.read()
    pcm, res = O.decode_stream(data)
    assert res.n_errors == 0 and res.n_frames == frames
    assert bytes(res.md5).hex().startswith(md5)
    assert O.md5_pcm(pcm, res.bps) == bytes(res.md5)


def test_8bit_fixture_decodes():
    with open(os.path.join(cases.GOLDEN, 'data', '8bit.flac'), 'rb') as f:
        pcm, res = O.decode_stream(f.read())
    assert res.bps == 8 and res.n_frames == 11 and res.n_errors == 0


def test_window_hashes():
    import json
    with open(os.path.join(cases.GOLDEN, 'window_hashes.json')) as f:
        want = json.load(f)
    for key, h in want.items():
        lvl, n = key.split('_')
        cfg, _ = O.config(int(lvl[1:]), 2, 16, 48000, 4096)
        assert hashlib.sha256(O.window(cfg, int(n[1:])).tobytes()).hexdigest() == h, key


def test_crc_known_answers():
    buf = np.frombuffer(b'123456789', np.uint8)
    assert O.lib().flo_crc8(buf.ctypes.data, 9) == 0xF4          # CRC-8 poly 0x07
    assert O.lib().flo_crc16(buf.ctypes.data, 9) == 0xFEE8       # CRC-16/BUYPASS poly 0x8005


def test_md5_known_answer():
    a = np.frombuffer(b'ab', np.int16).reshape(1, 1)
    assert O.md5_pcm(a, 16).hex() == hashlib.md5(b'ab').hexdigest()


@pytest.mark.parametrize('name', ['eight_ch', 'escape16', 'rice2_24', 'side33', 'variable'])
def test_decoder_on_handmade_streams(name, handmade_streams):
    """Escape-coded partitions, RICE2, partition order 8, LPC order 32, variable block sizes and every header form,
    33-bit side channels: the oracle decoder returns what the reference binary returned."""
    data, pcm = handmade_streams[name]
    out, res = O.decode_stream(data)
    assert res.n_errors == 0
    assert np.array_equal(out.reshape(pcm.shape), pcm)
