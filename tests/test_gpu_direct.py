"""GPU tests of the direct packing path (round 5): fg_pipe_pack_kernel<DIRECT> assembles a frame in LDS, takes the CRC-16 there
and stores the bytes at the frame's final place, which a decoupled look-back over the frame sizes supplies.

Bar: byte-identical to the chunk form of rounds 2-4 (flacgpu_set_direct(ctx, 0): chunks through HBM, sizes scan, assembly
kernel) and to the CPU oracle; the frame index and the statistics equal too.  Layouts: one stream (the tail block keeps the
chunk form and is placed behind the direct frames), many streams with ragged tails in FRONT of direct blocks (their chain runs
beside the analysis and the direct kernels wait for it), tiny streams between long ones, mono, noise (verbatim frames: the
largest frames there are), int16 ingest, loose mid-side, a launch of two groups.
"""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=[1, 2], ids=['direct', 'fused'])
def ctxs(request):
    """(direct, chunk form).  direct = 1: the packing kernel places its frames behind a separate evaluation kernel; 2: the evaluation
    runs inside the packing kernel (fg_pipe_pack_kernel<FUSED>, blocks of 4096 samples)."""
    import torch
    from pyflac_amd import batch, _lib
    assert torch.cuda.is_available()
    a, b = batch.Context(0), batch.Context(0)
    _lib.lib().flacgpu_set_direct(a._h, request.param)
    _lib.lib().flacgpu_set_direct(b._h, 0)
    return a, b


def _pcm(seed, n, ch, kind='music', bps=16):
    rng = np.random.default_rng(seed)
    if kind == 'noise':
        lim = 1 << (bps - 1)
        return rng.integers(-lim, lim, size=(n, ch), dtype=np.int64).astype(np.int32)
    t = np.arange(n)
    x = np.zeros((n, ch))
    for c in range(ch):
        x[:, c] = 9000 * np.sin(2 * np.pi * (220 + 37 * c + seed % 50) * t / 48000.0) + 3000 * np.sin(2 * np.pi * 1733.0 * t / 48000.0 + c)
    x += rng.normal(0, 300, size=(n, ch))
    return np.clip(np.round(x), -32768, 32767).astype(np.int32)


def _both(ctxs, s, pcm, lengths=None, i16=False):
    import torch
    a, b = ctxs
    t = torch.from_numpy(np.ascontiguousarray(pcm.astype(np.int16 if i16 else np.int32))).cuda()
    oa, fa, sa = a.encode(s, t, stream_lengths=lengths)
    ob, fb, sb = b.encode(s, t, stream_lengths=lengths)
    assert sb.direct_path == 0
    assert sa.total_bytes == sb.total_bytes and sa.nblocks == sb.nblocks
    assert torch.equal(fa, fb), 'frame index differs'
    assert torch.equal(oa[:sa.total_bytes], ob[:sb.total_bytes]), 'bytes differ'
    assert sa.log_guard_subframes == sb.log_guard_subframes and sa.lpc_order_min_margin == sb.lpc_order_min_margin
    return oa[:sa.total_bytes].cpu().numpy().tobytes(), fa.cpu().numpy(), sa


def _check_oracle(level, ch, bps, sr, bs, pcm, lengths, got):
    from oracle import oracle as O
    cfg, _ = O.config(level, ch, bps, sr, bs)
    want, pos = [], 0
    for n in lengths:
        stream, sizes = O.encode_stream(cfg, pcm[pos:pos + n])
        want.append(stream[len(stream) - int(sizes.astype(np.int64).sum()):])          # (the frames, without the stream header)
        pos += n
    want = b''.join(want)
    assert len(got) == len(want) and hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()


def test_one_stream_with_a_tail(ctxs):
    from pyflac_amd import batch
    n = 4096 * 37 + 1000
    pcm = _pcm(1, n, 2)
    s = batch.settings(5, 2, 16, 48000, 4096)
    got, offs, st = _both(ctxs, s, pcm)
    assert st.direct_path == 1 and st.nblocks == 38
    _check_oracle(5, 2, 16, 48000, 4096, pcm, [n], got)


def test_many_streams_tails_in_front_of_direct_blocks(ctxs):
    from pyflac_amd import batch
    lengths = [4096 * 5 + 512, 4096 * 3, 700, 4096 * 9 + 4095, 49, 4096 * 2 + 33, 4096, 4096 * 4 + 2048, 65]
    pcm = _pcm(2, sum(lengths), 2)
    s = batch.settings(5, 2, 16, 48000, 4096)
    got, offs, st = _both(ctxs, s, pcm, lengths)
    assert st.direct_path == 1
    _check_oracle(5, 2, 16, 48000, 4096, pcm, lengths, got)


def test_a_generic_block_in_front_disables_the_direct_path(ctxs):
    """A 20-sample stream takes the generic kernel; in front of direct blocks the call keeps the chunk form."""
    from pyflac_amd import batch
    lengths = [20, 4096 * 3]
    pcm = _pcm(3, sum(lengths), 2)
    s = batch.settings(5, 2, 16, 48000, 4096)
    got, offs, st = _both(ctxs, s, pcm, lengths)
    assert st.direct_path == 0
    lengths = [4096 * 3, 20]          # behind them: its size is published in front of the assembly of the rest
    got, offs, st = _both(ctxs, s, pcm, lengths)
    assert st.direct_path == 1
    _check_oracle(5, 2, 16, 48000, 4096, pcm, lengths, got)


@pytest.mark.parametrize('level', [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_levels(ctxs, level):
    from pyflac_amd import batch
    lengths = [4096 * 6 + 100, 4096 * 7]
    pcm = _pcm(10 + level, sum(lengths), 2)
    s = batch.settings(level, 2, 16, 44100, 4096)
    got, offs, st = _both(ctxs, s, pcm, lengths)
    assert st.direct_path == 1
    _check_oracle(level, 2, 16, 44100, 4096, pcm, lengths, got)


def test_noise_gives_verbatim_frames_that_fill_the_buffer(ctxs):
    from pyflac_amd import batch
    lengths = [4096 * 4, 4096 * 3 + 64]
    pcm = _pcm(4, sum(lengths), 2, 'noise')
    s = batch.settings(5, 2, 16, 48000, 4096)
    got, offs, st = _both(ctxs, s, pcm, lengths)
    assert st.direct_path == 1
    _check_oracle(5, 2, 16, 48000, 4096, pcm, lengths, got)


def test_residuals_beyond_16_bits_take_the_second_walk(ctxs):
    """Full-scale spikes on a quiet signal: the residual of a spike needs 17+ bits, so the wave cannot keep its residuals two to a
    register and walks its samples a second time -- with the frame buffer over the staged samples (ALIAS) from memory."""
    from pyflac_amd import batch
    n = 4096 * 12
    pcm = (_pcm(21, n, 2) // 40).astype(np.int32)
    rng = np.random.default_rng(5)
    for p in rng.integers(10, n - 10, size=40):
        pcm[p, rng.integers(0, 2)] = -32768
        pcm[p + 1, rng.integers(0, 2)] = 32767
    s = batch.settings(5, 2, 16, 48000, 4096)
    got, offs, st = _both(ctxs, s, pcm, [4096 * 7, 4096 * 5])
    assert st.direct_path == 1
    _check_oracle(5, 2, 16, 48000, 4096, pcm, [4096 * 7, 4096 * 5], got)
    got, offs, st = _both(ctxs, s, pcm, [4096 * 7, 4096 * 5], i16=True)
    _check_oracle(5, 2, 16, 48000, 4096, pcm, [4096 * 7, 4096 * 5], got)


def test_mono_and_small_bit_depths(ctxs):
    from pyflac_amd import batch
    for ch, bps, bs in [(1, 16, 4096), (1, 8, 4096), (2, 12, 4096), (2, 16, 4608), (1, 16, 1152), (2, 16, 2304)]:
        lengths = [bs * 5 + 77, bs * 4]
        pcm = _pcm(5 + ch + bps, sum(lengths), ch)
        pcm = pcm >> (16 - bps)
        s = batch.settings(5, ch, bps, 48000, bs)
        got, offs, st = _both(ctxs, s, pcm, lengths)
        # (blocks of 1152 / 2304 samples are packed by one wave per subframe -- the spare waves only join the barriers and the CRC pass)
        assert st.direct_path == 1, (ch, bps, bs)
        _check_oracle(5, ch, bps, 48000, bs, pcm, lengths, got)


def test_int16_ingest_and_silence(ctxs):
    from pyflac_amd import batch
    lengths = [4096 * 5, 4096 * 5 + 9]
    pcm = _pcm(6, sum(lengths), 2)
    pcm[4096:4096 * 3] = 0                   # constant subframes: chunks of a few bits, two of them empty
    pcm[4096 * 6:4096 * 7, 0] = 1234
    s = batch.settings(5, 2, 16, 48000, 4096)
    got, offs, st = _both(ctxs, s, pcm, lengths, i16=True)
    assert st.direct_path == 1
    _check_oracle(5, 2, 16, 48000, 4096, pcm, lengths, got)


def test_two_groups_and_repeated_calls(ctxs):
    """4096 blocks and more are cut into two groups on two streams; the second group's packing kernel looks back into the first's.
    Repeated calls reuse the look-back words under a new epoch."""
    from pyflac_amd import batch
    n = 4096 * 4300 + 321
    pcm = _pcm(7, n, 2)
    s = batch.settings(5, 2, 16, 48000, 4096)
    first = None
    for _ in range(3):
        got, offs, st = _both(ctxs, s, pcm)
        assert st.direct_path == 1 and st.nblocks == 4301
        h = hashlib.sha256(got).hexdigest()
        assert first is None or h == first
        first = h
    # every frame's CRC-16 and the decoder agree
    import torch
    a, _ = ctxs
    t = torch.from_numpy(pcm).cuda()
    out, fo, st = a.encode(s, t)
    dec, status, _ = a.decode(out[:st.total_bytes], fo, 2, 16, n)
    assert int(status[:, 0].max()) == 0 and torch.equal(dec.reshape(-1, 2), t)


def test_short_output_buffer_is_reported(ctxs):
    """Frames that do not fit the caller's capacity are not written (nothing lands behind it) and the call says so."""
    import torch
    from pyflac_amd import batch, _lib
    L = _lib.lib()
    a, _ = ctxs
    n = 4096 * 8
    pcm = _pcm(8, n, 2, 'noise')
    s = batch.settings(5, 2, 16, 48000, 4096)
    t = torch.from_numpy(pcm).cuda()
    out = torch.zeros(400000, dtype=torch.uint8, device='cuda')
    offs = torch.zeros(9, dtype=torch.int64, device='cuda')
    d = (_lib.StreamDesc * 1)()
    d[0].pcm_offset = 0; d[0].nsamples = n; d[0].first_frame = 0
    st = _lib.EncodeStats()
    torch.cuda.synchronize()
    cap = 40000
    rc = L.flacgpu_encode_streams(a._h, C.byref(s), t.data_ptr(), 0, d, 1, out.data_ptr(), cap, offs.data_ptr(), C.byref(st))
    assert rc != 0 and b'too small' in _lib.last_error().encode()
    assert int(out[cap:].max()) == 0
    assert int(out[:cap].max()) != 0                 # (the frames that fit are there)
