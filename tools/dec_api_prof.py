"""StreamDecoder on a 600 s stream: where the wall time of the drop-in decode goes (GPU box).
FLACGPU_API_PROF=1 adds the library's own phase times; PYPROF=1 runs the decoder thread under cProfile."""
import cProfile, os, pstats, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyflac_amd
from pyflac_amd import synth, _lib
sec = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
pcm = synth.config2_stereo16(sec, 0, 48000)
chunks = []
enc = pyflac_amd.StreamEncoder(48000, lambda b, n, s, f: chunks.append(b), compression_level=5, blocksize=4096)
_lib.lib().FLAC__stream_encoder_set_do_md5(enc._encoder, 0)
enc.process(pcm); enc.finish()
stream = b''.join(chunks)
if os.environ.get('FILEMODE'):
    # what a finished file holds (FLAC__stream_encoder_finish seeks back and fills STREAMINFO in): min / max frame size
    from pyflac_amd import batch
    offs, _si = batch.index_frames(stream)
    sz = np.diff(np.asarray(offs, np.int64))
    stream = bytearray(stream)
    stream[12:15] = int(sz.min()).to_bytes(3, 'big'); stream[15:18] = int(sz.max()).to_bytes(3, 'big')
    stream = bytes(stream)
    print('file mode: min/max frame size', int(sz.min()), int(sz.max()), file=sys.stderr)
if os.environ.get('PYPROF'):
    threading.setprofile(None)
    prof = cProfile.Profile()
    orig = pyflac_amd.StreamDecoder._process
    def _p(self):
        prof.enable(); orig(self); prof.disable()
    pyflac_amd.StreamDecoder._process = _p
for rep in range(3):
    blocks = []
    dec = pyflac_amd.StreamDecoder(lambda a, r, c, n: blocks.append(a))
    t0 = time.perf_counter()
    dec.process(stream)
    t1 = time.perf_counter()
    dec.finish()
    t2 = time.perf_counter()
    print('rep %d: %.1f ms (process %.1f, finish %.1f), %.1f Msamples/s, %d blocks' % (rep, (t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, pcm.size / (t2 - t0) / 1e6, len(blocks)), file=sys.stderr)
if os.environ.get('PYPROF'):
    pstats.Stats(prof, stream=sys.stderr).sort_stats('cumulative').print_stats(18)
