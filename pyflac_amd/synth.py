"""Deterministic synthetic PCM generators for parity tests and bench.py.

These are the inputs SURVEY.md §8(d) defines for BASELINE.json's configs; the
SHA-256 checkpoints in that section pin them (tests/test_synth.py).
"""
import hashlib

import numpy as np


def pcm_hash(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def config1_sine(sample_rate=44100, seconds=1.0):
    """Config 1 (examples/passthrough.py shape): mono 16-bit 440 Hz sine."""
    n = int(sample_rate * seconds)
    t = np.arange(n) / sample_rate
    return np.round(0.5 * 32767 * np.sin(2 * np.pi * 440 * t)).astype(np.int16).reshape(-1, 1)


def config2_stereo16(seconds=20.0, seed=0, sample_rate=48000, base=440.0):
    """Config 2/3: stereo 16-bit 48 kHz sines + Gaussian noise (oracle ratio ~0.68)."""
    n = int(round(sample_rate * seconds))
    t = np.arange(n) / sample_rate
    rng = np.random.default_rng(seed)
    nl = rng.normal(0, 300, n)
    nr = rng.normal(0, 300, n)
    L = 8000 * np.sin(2 * np.pi * base * t) + 3000 * np.sin(2 * np.pi * (base * 1333.3 / 440.0) * t + 0.3) + nl
    R = 0.8 * L + 2000 * np.sin(2 * np.pi * 97 * t) + nr
    return np.stack([np.round(L), np.round(R)], axis=1).astype(np.int16)


def config2_hard16(seconds=5.0, seed=7, sample_rate=48000):
    """'Hard' input: noise whose sigma switches every 512 samples (exercises partition orders)."""
    n = int(round(sample_rate * seconds))
    rng = np.random.default_rng(seed)
    sig = np.array([3, 40, 500, 6000])[rng.integers(0, 4, (n + 511) // 512)]
    sig = np.repeat(sig, 512)[:n]
    L = rng.normal(0, 1, n) * sig
    R = 0.5 * L + rng.normal(0, 1, n) * sig * 0.5
    return np.clip(np.stack([np.round(L), np.round(R)], axis=1), -32768, 32767).astype(np.int16)


def config4_stereo24(seconds=10.0, seed=1, sample_rate=96000):
    """Config 4: stereo 24-bit 96 kHz (int32 container, values in +-2^23)."""
    n = int(round(sample_rate * seconds))
    t = np.arange(n) / sample_rate
    rng = np.random.default_rng(seed)
    env = 0.2 + 0.8 * np.abs(np.sin(2 * np.pi * 0.7 * t))
    nl = rng.normal(0, 2000, n)
    nr = rng.normal(0, 2000, n)
    L = env * (2.0e6 * np.sin(2 * np.pi * 220 * t) + 6e5 * np.sin(2 * np.pi * 3520.1 * t)) + env * nl
    R = 0.6 * L + env * 4e5 * np.sin(2 * np.pi * 55 * t) + nr
    a = np.stack([np.round(L), np.round(R)], axis=1)
    return np.clip(a, -(1 << 23), (1 << 23) - 1).astype(np.int32)


def config5_stream(s, seconds=60.0, sample_rate=48000):
    """Config 5: stream ``s`` of the 1024-stream batch."""
    return config2_stereo16(seconds=seconds, seed=1000 + s, sample_rate=sample_rate,
                            base=220.0 * 2 ** (s / 1024.0))
