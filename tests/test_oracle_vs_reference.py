"""Pin the oracle against the reference's bundled libFLAC binary on a wider seeded corpus.

Runs only where /root/reference exists (the build container); the GPU box relies on the
committed golden vectors instead.
"""
import numpy as np
import pytest

from oracle import libflac_ref as R
from oracle import oracle as O

pytestmark = pytest.mark.skipif(not R.available(), reason='reference binary not present')


def _check(pcm, sr, bps, level, bs=0, subset=True):
    pcm = np.asarray(pcm)
    ch = 1 if pcm.ndim == 1 else pcm.shape[1]
    arr = pcm.astype(np.int16 if bps == 16 else np.int32)
    extra = None if subset else [('set_streamable_subset', 0)]
    cbs, info = R.encode(arr, sr, bps=bps, level=level, blocksize=bs, extra=extra)
    cfg, rc = O.config(level, ch, bps, sr, bs, subset)
    assert rc == info['init_status']
    if rc:
        return
    ref = b''.join(c[0] for c in cbs)
    mine, _ = O.encode_stream(cfg, arr)
    assert mine == ref
    out, res = O.decode_stream(ref)
    assert res.n_errors == 0 and np.array_equal(out, arr.astype(np.int32).reshape(out.shape))
    rpcm, _frames, st = R.decode(mine, want_frames=False)
    assert not st['errors'] and np.array_equal(rpcm, out)


@pytest.mark.parametrize('seed', range(12))
def test_random_corpus(seed):
    r = np.random.default_rng(100 + seed)
    ch = int(r.choice([1, 2, 2, 2, 3, 6]))
    bps = int(r.choice([8, 12, 16, 16, 16, 20, 24, 24]))
    level = int(r.integers(0, 9))
    bs = int(r.choice([0, 0, 256, 576, 1000, 1152, 2304, 4096, 4608]))
    n = int(r.integers(3000, 20000))
    amp = (1 << (bps - 1)) - 1
    t = np.arange(n)[:, None]
    f = r.uniform(0.001, 0.3, ch)
    x = amp * r.uniform(0.05, 0.9) * np.sin(t * f + r.uniform(0, 6, ch))
    x = x * (0.3 + 0.7 * np.abs(np.sin(t * r.uniform(1e-4, 1e-3))))
    x = x + r.normal(0, amp * 10 ** r.uniform(-4, -1), (n, ch))
    if ch == 2 and r.random() < 0.5:
        x[:, 1] = 0.8 * x[:, 0] + 0.2 * x[:, 1]
    x = np.clip(np.round(x), -amp - 1, amp).astype(np.int64)
    _check(x, int(r.choice([44100, 48000, 96000, 22050])), bps, level, bs)


@pytest.mark.parametrize('chunk', range(4))
def test_fuzz_corpus_beyond_the_golden_seeds(chunk):
    """tests/fuzzgen.py seeds past the ones with committed hashes (tests/tools/fuzz_oracle_vs_ref.py runs thousands)."""
    from tests import fuzzgen
    for seed in range(fuzzgen.GOLDEN_SEEDS + chunk * 50, fuzzgen.GOLDEN_SEEDS + chunk * 50 + 50):
        c = fuzzgen.case(seed)
        arr = c['pcm'].astype(np.int16 if c['bps'] == 16 else np.int32)
        extra = []
        if not c['subset']:
            extra.append(('set_streamable_subset', 0))
        if c['limit_min_bitrate']:
            extra.append(('set_limit_min_bitrate', 1))
        cbs, info = R.encode(arr, c['sr'], bps=c['bps'], level=c['level'], blocksize=c['bs'], extra=extra or None)
        cfg, rc = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
        assert rc == info['init_status'], seed
        if rc:
            continue
        cfg.limit_min_bitrate = 1 if c['limit_min_bitrate'] else 0
        mine, _ = O.encode_stream(cfg, arr)
        assert mine == b''.join(x[0] for x in cbs), seed


@pytest.mark.parametrize('status_case', [
    (5, 2, 16, 2000000, 0, True), (5, 2, 16, 44100, 1000000, True), (5, 2, 16, 44100, 65535, True),
    (5, 2, 16, 44100, 65535, False), (5, 9, 16, 44100, 0, True), (5, 2, 3, 44100, 0, True),
    (5, 2, 64, 44100, 0, True), (5, 2, 16, 48000, 4609, True), (5, 2, 16, 96000, 16385, True),
    (5, 2, 17, 48000, 0, True), (5, 2, 16, 48000, 15, True), (8, 2, 16, 48000, 8, False)])
def test_init_status(status_case):
    level, ch, bps, sr, bs, subset = status_case
    x = np.zeros((100, min(ch, 8)), np.int32)
    L = R.lib()
    import ctypes as C
    enc = C.c_void_p(L.FLAC__stream_encoder_new())
    L.FLAC__stream_encoder_set_channels(enc, ch)
    L.FLAC__stream_encoder_set_bits_per_sample(enc, bps)
    L.FLAC__stream_encoder_set_sample_rate(enc, sr)
    L.FLAC__stream_encoder_set_compression_level(enc, level)
    L.FLAC__stream_encoder_set_blocksize(enc, bs)
    L.FLAC__stream_encoder_set_streamable_subset(enc, 1 if subset else 0)
    wcb = R.ENC_WRITE_CB(lambda *a: 0)
    rc = L.FLAC__stream_encoder_init_stream(enc, wcb, None, None, None, None)
    L.FLAC__stream_encoder_delete(enc)
    _cfg, mine = O.config(level, ch, bps, sr, bs, subset)
    assert mine == rc
    del x
