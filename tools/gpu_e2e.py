"""End-to-end wall time of the drop-in (callback) API: StreamEncoder.process(numpy) -> bytes -> StreamDecoder -> numpy."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import pyflac_amd
from pyflac_amd import synth, _lib

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
pcm = synth.config2_stereo16(secs, 0)
for md5 in (1, 0):
    for rep in range(2):
        chunks = []
        enc = pyflac_amd.StreamEncoder(48000, lambda b, n, s, f: chunks.append(b), compression_level=5, blocksize=4096)
        t0 = time.perf_counter()
        if not md5:
            _lib.lib().FLAC__stream_encoder_set_do_md5(enc._encoder, 0)
        enc.process(pcm)
        enc.finish()
        t1 = time.perf_counter()
        stream = b''.join(chunks)
        blocks = []
        dec = pyflac_amd.StreamDecoder(lambda a, sr, ch, n: blocks.append(a))
        t2 = time.perf_counter()
        dec.process(stream)
        dec.finish()
        t3 = time.perf_counter()
        n = pcm.size
        print('md5=%d rep %d: encode %.3f s (%.1f Msamples/s)  decode %.3f s (%.1f Msamples/s)  frames %d' %
              (md5, rep, t1 - t0, n / (t1 - t0) / 1e6, t3 - t2, n / (t3 - t2) / 1e6, len(chunks) - 3))
