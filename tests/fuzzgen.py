"""Seeded random encode cases for differential testing (oracle <-> reference binary here, GPU <-> oracle on the GPU box).

case(seed) -> dict(pcm[n, ch] int64, sr, bps, level, bs, subset, ch).  The signal families aim at the encoder's decision
boundaries: near-silence and DC (CONSTANT / order-0), wasted low bits, full-scale noise (VERBATIM, Rice parameter limits,
RICE2 at > 16 bit), clipped squares and impulses (fixed vs LPC ties, residual overflow guards at >= 28 bit), strongly
correlated and anti-correlated channel pairs (all four stereo assignments), short and ragged final blocks.
"""
import numpy as np

GOLDEN_SEEDS = 240     # seeds 0..239 have reference-recorded hashes in tests/golden/fuzz_vectors.json
SRS = [8000, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000]


def _signal(r, n, ch, bps):
    amp = (1 << (bps - 1)) - 1
    t = np.arange(n, dtype=np.float64)[:, None]
    kind = r.choice(['sines', 'noise', 'full_noise', 'silence', 'dc', 'square', 'impulses', 'ramp', 'wasted', 'decay', 'steps',
                     'sines', 'sines', 'mixed'])
    if kind == 'sines' or kind == 'mixed':
        f = r.uniform(0.0005, 0.45, ch)
        x = amp * r.uniform(0.01, 0.95) * np.sin(t * f + r.uniform(0, 6, ch))
        x += amp * r.uniform(0, 0.3) * np.sin(t * r.uniform(0.001, 1.0, ch))
        x += r.normal(0, amp * 10 ** r.uniform(-5, -1), (n, ch))
        if kind == 'mixed':
            k = int(r.integers(1, max(2, n // 2)))
            x[k:] = r.normal(0, amp * 10 ** r.uniform(-3, -0.5), (n - k, ch))
    elif kind == 'noise':
        x = r.normal(0, amp * 10 ** r.uniform(-4, -0.3), (n, ch))
    elif kind == 'full_noise':
        x = r.integers(-amp - 1, amp + 1, (n, ch)).astype(np.float64)
    elif kind == 'silence':
        x = np.zeros((n, ch))
        if r.random() < 0.5:
            x += r.integers(-1, 2, (n, ch))
    elif kind == 'dc':
        x = np.ones((n, ch)) * r.integers(-amp - 1, amp + 1, ch)
        if r.random() < 0.4:
            x[int(r.integers(0, n))] += 1
    elif kind == 'square':
        per = int(r.integers(2, 400))
        x = np.where((np.arange(n) // per) % 2 == 0, 1.0, -1.0)[:, None] * amp * r.uniform(0.2, 1.0, ch)
        if r.random() < 0.5:
            x = np.where(x > 0, amp, -amp - 1) * np.ones((1, ch))
    elif kind == 'impulses':
        x = np.zeros((n, ch))
        idx = r.integers(0, n, max(1, n // int(r.integers(20, 2000))))
        x[idx] = r.integers(-amp - 1, amp + 1, (len(idx), ch))
    elif kind == 'ramp':
        x = (t * r.uniform(-3, 3, ch) * amp / max(n, 1)) + r.normal(0, r.uniform(0, 3), (n, ch))
    elif kind == 'wasted':
        w = int(r.integers(1, max(2, bps - 2)))
        x = np.round(r.normal(0, amp * 10 ** r.uniform(-3, -0.5), (n, ch)) / (1 << w)) * (1 << w)
        if r.random() < 0.3:
            x[:, -1] = np.round(x[:, -1] / (1 << min(bps - 2, w + 2))) * (1 << min(bps - 2, w + 2))
    elif kind == 'decay':
        x = amp * np.exp(-t / r.uniform(n / 50 + 1, max(n, n / 50 + 1))) * np.sin(t * r.uniform(0.01, 0.4, ch)) + r.normal(0, 0.6, (n, ch))
    else:  # steps
        lv = r.integers(-amp - 1, amp + 1, (n // int(r.integers(50, 3000)) + 2, ch))
        x = lv[np.minimum(np.arange(n) * len(lv) // max(n, 1), len(lv) - 1)].astype(np.float64)
    if ch >= 2:
        m = r.random()
        if m < 0.25:
            x[:, 1] = x[:, 0] + r.normal(0, 2, n)
        elif m < 0.4:
            x[:, 1] = -x[:, 0] + r.normal(0, 2, n)
        elif m < 0.5:
            x[:, 1] = x[:, 0]
    return np.clip(np.round(x), -amp - 1, amp).astype(np.int64), str(kind)


def case(seed):
    r = np.random.default_rng(7000 + seed)
    ch = int(r.choice([1, 2, 2, 2, 2, 3, 4, 6, 8]))
    bps = int(r.choice([8, 12, 16, 16, 16, 16, 20, 24, 24, 24, 32]))
    level = int(r.integers(0, 9))
    sr = int(r.choice(SRS))
    subset = bool(r.random() < 0.8)
    bs = int(r.choice([0, 0, 0, 16, 64, 192, 256, 576, 1000, 1024, 1152, 2048, 2304, 4096, 4096, 4608, 8192, 16384]))
    if subset and sr <= 48000 and bs > 4608:
        bs = 4096
    nominal = bs if bs else (1152 if level < 3 else 4096)
    n = int(r.choice([nominal * int(r.integers(1, 5)) + int(r.integers(0, nominal)), int(r.integers(1, 3 * nominal)),
                      nominal * int(r.integers(1, 4))]))
    n = max(1, min(n, 40000))
    pcm, kind = _signal(r, n, ch, bps)
    return {'pcm': pcm, 'sr': sr, 'bps': bps, 'level': level, 'bs': bs, 'subset': subset, 'ch': ch, 'kind': kind,
            'limit_min_bitrate': bool(r.random() < 0.15)}
