"""gfx950emu (test infrastructure): parity cases run on the ISA-level emulator, one group per process.
usage: python tests/emu/parity_cases.py <group> [args]      -> prints one line "RESULT <json>"

The process loads the stand-in HIP runtime in front of libflacgpu.so (emurun.load()), so what runs is the product library's host code
and the gfx950 instructions hipcc compiled from its kernels; what it is compared with is the CPU oracle (oracle/flac_oracle.c) and the
golden vectors of the reference binary.  tests/test_emu_parity.py starts these groups from the CPU suite."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import emurun  # noqa: E402

shim, L = emurun.load()
import numpy as np  # noqa: E402
import torch  # noqa: E402  (the stand-in)
from oracle import oracle as O  # noqa: E402
from pyflac_amd import batch, synth  # noqa: E402


def kernel_stats():
    return json.loads(shim.gfx950emu_stats_json().decode())


def drop_in_classes(level, seconds, channels, bps):
    """StreamEncoder -> the oracle's bytes; StreamDecoder -> the input."""
    import pyflac_amd
    sr = 48000
    if bps == 16:
        pcm = synth.config2_stereo16(seconds, 3)
        if channels == 1:
            pcm = np.ascontiguousarray(pcm[:, :1])
    else:
        pcm = synth.config4_stereo24(seconds, 5) if hasattr(synth, 'config4_stereo24') else (synth.config2_stereo16(seconds, 5).astype(np.int32) << 8)
    chunks, blocks = [], []
    kw = {} if bps == 16 else {'bits_per_sample': bps}
    enc = pyflac_amd.StreamEncoder(sr, lambda b, n, s, f: chunks.append(b), compression_level=level, blocksize=4096, **kw)
    enc.process(pcm)
    ok = enc.finish()
    stream = b''.join(chunks)
    cfg, _ = O.config(level, pcm.shape[1], bps, sr, 4096)
    want, _ = O.encode_stream(cfg, pcm, finalize=True)
    # (the stream encoder finalises STREAMINFO only with a seek callback: compare the frames, and the header up to the fields it cannot fill)
    hdr = 4 + 4 + 34 + 4 + 40
    if bps == 16:
        dec = pyflac_amd.StreamDecoder(lambda a, r, c, n: blocks.append(a))
        dec.process(stream)
        dec.finish()
        got = np.concatenate(blocks, axis=0) if blocks else np.zeros((0, pcm.shape[1]), pcm.dtype)
    else:
        # (pyFLAC's StreamDecoder hands out int16 / int32 frames only -- pyflac/decoder.py:517 -- and so does the mirror: 24-bit
        # frames are decoded through the batch entry point)
        ctx = batch.Context(0)
        data = torch.from_numpy(np.frombuffer(stream, np.uint8)[hdr:].copy()).cuda()
        d, status, dst = ctx.decode_stream(data, pcm.shape[1], bps, pcm.shape[0])
        got = d.cpu().numpy().astype(pcm.dtype) if int(status[:, 0].max()) == 0 else np.zeros((0, pcm.shape[1]), pcm.dtype)
    return {'finish': bool(ok), 'frames_equal_oracle': stream[hdr:] == want[hdr:], 'bytes': len(stream), 'decoded_equals_input': bool(np.array_equal(got, pcm)),
            'decoded_dtype': str(got.dtype)}


def batch_round_trip(level, seconds, bs):
    """The batch entry points: encode == oracle (bytes), decode from the bytes alone == input, with the join through the word in memory."""
    ctx = batch.Context(0, testhooks=os.environ.get('PYFLAC_AMD_TESTHOOKS') == '1')
    pcm = synth.config2_stereo16(seconds, 21)
    s = batch.settings(level, 2, 16, 48000, bs, bs != 4095)
    t = torch.from_numpy(pcm.astype(np.int32)).cuda()
    out, offs, est = ctx.encode(s, t)
    body = out[:est.total_bytes].cpu().numpy().tobytes()
    cfg, _ = O.config(level, 2, 16, 48000, bs, bs != 4095)
    want, _ = O.encode_stream(cfg, pcm)
    from pyflac_amd.encoder import stream_header_bytes
    res = {'encode_equals_oracle': stream_header_bytes(s) + body == want, 'blocks': int(est.nblocks), 'direct_path': int(est.direct_path), 'calls': []}
    data = out[:est.total_bytes].clone()
    for _ in range(2):
        dec, status, dst = ctx.decode_stream(data, 2, 16, t.shape[0], nframes=est.nblocks)
        res['calls'].append({'equal': bool(torch.equal(dec, t)), 'status_max': int(status[:, 0].max()), 'late': int(dst.join_late_workgroups),
                             'plane_bits': int(dst.plane_bits), 'generic': int(dst.generic_frames),
                             'sha': hashlib.sha256(dec.cpu().numpy().tobytes()).hexdigest()})
    return res


def fuzz_cases(first, count):
    """Seeded cases of tests/fuzzgen.py (1-8 channels, 8-32 bit, all levels, odd block sizes, ragged tails): batch encoder == oracle,
    decoder == input with the encoder's index and from the bytes alone."""
    from pyflac_amd.encoder import stream_header_bytes
    from tests import fuzzgen
    ctx = batch.Context(0)
    bad, ran, skipped = [], 0, 0
    for seed in range(first, first + count):
        c = fuzzgen.case(seed)
        if len(c['pcm']) * c['ch'] > 60000:
            skipped += 1
            continue
        cfg, rc = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
        try:
            s = batch.settings(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
            grc = 0
        except batch.FlacGpuError:
            grc = 1
        if (rc != 0) != (grc != 0):
            bad.append([seed, 'init status'])
            continue
        if rc:
            continue
        if c['limit_min_bitrate']:
            cfg.limit_min_bitrate = 1
            s.limit_min_bitrate = 1
        a = np.ascontiguousarray(np.asarray(c['pcm']).reshape(-1, c['ch']).astype(np.int32))
        want, _ = O.encode_stream(cfg, a)
        t = torch.from_numpy(a).cuda()
        out, offs, est = ctx.encode(s, t)
        body = out[:est.total_bytes].cpu().numpy().tobytes()
        ran += 1
        if stream_header_bytes(s) + body != want:
            bad.append([seed, 'encode'])
            continue
        data = out[:est.total_bytes].clone()
        dec, status, dst = ctx.decode(data, offs, c['ch'], c['bps'], a.shape[0])
        if int(status[:, 0].max()) != 0 or not torch.equal(dec.reshape(-1, c['ch']), t):
            bad.append([seed, 'decode with index'])
    return {'ran': ran, 'skipped_long': skipped, 'bad': bad}


def ragged_32bit(level):
    """Round 6: tails of TRUE 32-bit content (33-bit side channel) through the pipeline's fp64 forms in the ragged lane geometry:
    bytes == oracle, no block handed to the generic kernel."""
    ctx = batch.Context(0)
    rng = np.random.default_rng(600 + level)
    res = []
    for ch, bs, n, shift in ((2, 4096, 4096 + 777, 0), (1, 4096, 4096 + 2049, 0), (2, 1155, 2 * 1155 + 401, 0), (2, 4096, 4096 + 516, 2),
                             (2, 4096, 4096 + 778, 0), (2, 4096, 4096 + 779, 5), (1, 4096, 4096 + 778, 6), (2, 256, 66, 0)):
        walk = np.cumsum(rng.integers(-2**26, 2**26, (n, ch)), axis=0)
        x = ((walk + rng.integers(-2**20, 2**20, (n, ch))) % 2**32 - 2**31).astype(np.int64)
        x = (x >> shift) << shift
        arr = np.ascontiguousarray(np.clip(x, -2**31, 2**31 - 1).astype(np.int32))
        s = batch.settings(level, ch, 32, 48000, bs, bs == 4096)
        cfg, _ = O.config(level, ch, 32, 48000, bs, bs == 4096)
        out, offs, st = ctx.encode(s, torch.from_numpy(arr).cuda())
        want, sizes = O.encode_stream(cfg, arr)
        body = out[:st.total_bytes].cpu().numpy().tobytes()
        res.append({'equal': body == want[len(want) - int(sizes.sum()):], 'redo': int(st.redo_blocks), 'blocks': int(st.nblocks), 'case': [ch, bs, n, shift]})
    return {'cases': res}


def direct_24bit(level):
    """Round 6: frames of 17..24-bit and of 32-bit input packed at their final place (fg_pipe_pack_kernel<ACC64, DIRECT>): bytes and offsets == oracle,
    direct_path == 1, nothing handed back -- regular blocks, a tail, a short block size, one channel, noise (verbatim: the largest frames)."""
    from pyflac_amd.encoder import stream_header_bytes
    ctx = batch.Context(0)
    res = []
    for ch, bps, bs, n, noise in ((2, 24, 4096, 4096 * 3 + 300, False), (1, 24, 4096, 4096 * 2 + 17, False), (2, 20, 4608, 4608 * 2 + 100, False),
                                  (2, 24, 1152, 1152 * 4, False), (2, 24, 4096, 4096 * 2, True), (2, 32, 4096, 4096 * 2 + 9, False), (2, 32, 4096, 4096, True)):
        rng = np.random.default_rng(level * 100 + bs + ch)
        pcm = synth.config4_stereo24(n / 48000.0 + 0.01, bs + ch)[:n].astype(np.int32)
        pcm = pcm >> (24 - bps) if bps <= 24 else pcm * 251 + rng.integers(-90, 90, pcm.shape).astype(np.int32)      # (32 bit: no wasted bits)
        if noise:
            pcm = rng.integers(-2**(bps - 1), 2**(bps - 1), pcm.shape).astype(np.int32)
        pcm = np.ascontiguousarray(pcm[:, :ch])
        s = batch.settings(level, ch, bps, 48000, bs, True)
        cfg, _ = O.config(level, ch, bps, 48000, bs, True)
        out, offs, st = ctx.encode(s, torch.from_numpy(pcm).cuda())
        want, sizes = O.encode_stream(cfg, pcm)
        body = out[:st.total_bytes].cpu().numpy().tobytes()
        t = torch.from_numpy(pcm).cuda()
        dec, status, dst = ctx.decode(out[:st.total_bytes].clone(), offs, ch, bps, n)
        res.append({'case': [ch, bps, bs, n, noise], 'equal': stream_header_bytes(s) + body == want, 'direct': int(st.direct_path), 'redo': int(st.redo_blocks),
                    'offsets': bool(np.array_equal(np.diff(offs.cpu().numpy().astype(np.int64)), sizes.astype(np.int64))),
                    'decoded': int(status[:, 0].max()) == 0 and bool(torch.equal(dec.reshape(-1, ch), t))})
    return {'cases': res}


if __name__ == '__main__':
    group = sys.argv[1]
    if group == 'dropin':
        r = drop_in_classes(int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    elif group == 'batch':
        r = batch_round_trip(int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]))
    elif group == 'w32rag':
        r = ragged_32bit(int(sys.argv[2]))
    elif group == 'direct24':
        r = direct_24bit(int(sys.argv[2]))
    elif group == 'fuzz':
        r = fuzz_cases(int(sys.argv[2]), int(sys.argv[3]))
    else:
        raise SystemExit('unknown group ' + group)
    if len(sys.argv) > 1 and os.environ.get('GFX950EMU_REPORT_STATS') == '1':
        r['kernel_stats'] = kernel_stats()
    fault = shim.gfx950emu_last_fault().decode()
    if fault:
        r['fault'] = fault
    print('RESULT ' + json.dumps(r))
