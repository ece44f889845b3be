"""Edge cases at the limits of the format: GPU encoder vs CPU oracle (and, where /root/reference exists, the oracle vs
the reference binary).  usage: python tests/tools/extremes.py [ref|gpu]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O

r = np.random.default_rng(5)


def sig(n, ch, bps, kind):
    amp = (1 << (bps - 1)) - 1
    if kind == 'noise':
        return r.integers(-amp - 1, amp + 1, (n, ch)).astype(np.int64)
    t = np.arange(n)[:, None]
    x = amp * 0.8 * np.sin(t * r.uniform(0.001, 0.1, ch)) + r.normal(0, amp * 1e-3, (n, ch))
    return np.clip(np.round(x), -amp - 1, amp).astype(np.int64)


CASES = [
    # name, n, ch, bps, sr, level, bs, subset, kind
    ('one_sample', 1, 2, 16, 48000, 5, 4096, True, 'sine'),
    ('two_samples_8ch', 2, 8, 24, 48000, 8, 4096, True, 'sine'),
    ('bs65535_8ch_32bit', 70000, 8, 32, 96000, 5, 65535, False, 'sine'),
    ('bs65535_stereo32_noise', 66000, 2, 32, 192000, 8, 65535, False, 'noise'),
    ('bs32768_mono16', 40000, 1, 16, 96000, 8, 32768, False, 'sine'),
    ('bs16_stereo', 1000, 2, 16, 8000, 8, 16, False, 'sine'),
    ('sr_1hz', 5000, 1, 16, 1, 5, 4096, False, 'sine'),
    ('sr_655350', 5000, 2, 16, 655350, 5, 4096, False, 'sine'),
    ('sr_1048575', 5000, 2, 24, 1048575, 3, 1152, False, 'sine'),
    ('sr_65535_subset', 5000, 2, 16, 65535, 5, 4096, True, 'sine'),
    ('sr_255000', 5000, 2, 16, 255000, 5, 4096, False, 'sine'),
    ('bps8_noise', 9000, 2, 8, 44100, 8, 4608, True, 'noise'),
    ('bps12_8ch', 9000, 8, 12, 44100, 6, 576, True, 'sine'),
    ('stereo32_fullscale_noise', 9000, 2, 32, 48000, 5, 4096, True, 'noise'),
    ('mono32_fullscale_noise', 9000, 1, 32, 48000, 8, 4096, True, 'noise'),
    ('blocksize_17', 1000, 2, 24, 48000, 8, 17, False, 'sine'),
    ('long_tail_4095', 4096 + 4095, 2, 24, 48000, 5, 4096, True, 'sine'),
    ('tail_5', 4096 + 5, 2, 32, 48000, 8, 4096, True, 'sine'),
]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'gpu'
    bad = 0
    if mode == 'gpu':
        import torch
        from pyflac_amd import batch
        from pyflac_amd.encoder import stream_header_bytes
        ctx = batch.Context(0)
    else:
        from oracle import libflac_ref as R
    for name, n, ch, bps, sr, level, bs, subset, kind in CASES:
        pcm = sig(n, ch, bps, kind)
        cfg, rc = O.config(level, ch, bps, sr, bs, subset)
        a32 = np.ascontiguousarray(pcm.astype(np.int32))
        if mode == 'ref':
            arr = pcm.astype(np.int16 if bps == 16 else np.int32)
            cbs, info = R.encode(arr, sr, bps=bps, level=level, blocksize=bs, extra=None if subset else [('set_streamable_subset', 0)])
            if rc != info['init_status']:
                print('INIT', name, rc, info['init_status']); bad += 1
                continue
            if rc:
                print('%-28s init status %d (both)' % (name, rc)); continue
            want = b''.join(c[0] for c in cbs)
            got, _ = O.encode_stream(cfg, a32)
            out, res = O.decode_stream(want)
            ok = got == want and res.n_errors == 0 and np.array_equal(out.reshape(a32.shape), a32)
        else:
            try:
                s = batch.settings(level, ch, bps, sr, bs, subset)
                grc = 0
            except batch.FlacGpuError:
                grc = 1
            if (rc != 0) != (grc != 0):
                print('INIT', name, rc, grc); bad += 1
                continue
            if rc:
                print('%-28s init status %d (both)' % (name, rc)); continue
            want, _ = O.encode_stream(cfg, a32)
            t = torch.from_numpy(a32).cuda()
            out, offs, st = ctx.encode(s, t)
            got = stream_header_bytes(s) + out[:st.total_bytes].cpu().numpy().tobytes()
            dec, status, _ = ctx.decode(out[:st.total_bytes], offs, ch, bps, n)
            ok = got == want and int(status[:, 0].max()) == 0 and torch.equal(dec.reshape(-1, ch), t)
        print('%-28s %s (%d bytes)' % (name, 'ok' if ok else 'DIFF', len(want)))
        bad += 0 if ok else 1
    print('%d bad' % bad)


if __name__ == '__main__':
    main()
