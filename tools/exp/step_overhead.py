"""Where a step's wall time goes that is not GPU time: the Python wrappers, torch's stream synchronisation, the C calls themselves.
python tools/exp/step_overhead.py [seconds] [reps]"""
import os, sys, time
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyflac_amd import batch, synth, _lib
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pcm = synth.config2_stereo16(secs, 3)
t = torch.from_numpy(pcm.astype(np.int32)).cuda()
a = batch.Context(0)
s = batch.settings(5, 2, 16, 48000, 4096)
o, f, st = a.encode(s, t)
n = t.shape[0]
dec = None
for _ in range(5):
    o, f, st = a.encode(s, t, out=o, offsets=f)
    dec, status, ds = a.decode_stream(o[:st.total_bytes], 2, 16, n, nframes=st.nblocks, out=dec)
torch.cuda.synchronize()
T = time.perf_counter
eg = dg = 0.0
t0 = T()
for _ in range(reps):
    o, f, st = a.encode(s, t, out=o, offsets=f)
    dec, status, ds = a.decode_stream(o[:st.total_bytes], 2, 16, n, nframes=st.nblocks, out=dec)
    eg += st.total_gpu_ms; dg += ds.total_gpu_ms
wall = (T() - t0) / reps * 1e3
print('step through the wrappers: wall %.3f ms, GPU encode %.3f + decode %.3f = %.3f, rest %.3f' % (wall, eg / reps, dg / reps, (eg + dg) / reps, wall - (eg + dg) / reps))
# the C calls alone
L = _lib.lib()
descs = (_lib.StreamDesc * 1)(); descs[0].pcm_offset = 0; descs[0].nsamples = n; descs[0].first_frame = 0
est = _lib.EncodeStats(); dst = _lib.DecodeStats()
stat = np.zeros((st.nblocks, 2), np.uint32)
data = o[:st.total_bytes]
eg = dg = 0.0
t0 = T()
for _ in range(reps):
    L.flacgpu_encode_streams(a._h, C.byref(s), t.data_ptr(), 0, descs, 1, o.data_ptr(), o.numel(), f.data_ptr(), C.byref(est))
    L.flacgpu_decode_stream_dev(a._h, data.data_ptr(), est.total_bytes, est.nblocks, 0, 2, 16, dec.data_ptr(), n, stat.ctypes.data, stat.shape[0], None, C.byref(dst))
    eg += est.total_gpu_ms; dg += dst.total_gpu_ms
wall = (T() - t0) / reps * 1e3
print('the two C calls alone:     wall %.3f ms, GPU encode %.3f + decode %.3f = %.3f, rest %.3f' % (wall, eg / reps, dg / reps, (eg + dg) / reps, wall - (eg + dg) / reps))
t0 = T()
for _ in range(2000): torch.cuda.current_stream(t.device).synchronize()
print('torch.cuda.current_stream().synchronize(): %.1f us' % ((T() - t0) / 2000 * 1e6))
t0 = T()
for _ in range(2000): x = o[:st.total_bytes]
print('tensor slice: %.1f us' % ((T() - t0) / 2000 * 1e6))
t0 = T()
for _ in range(2000): z = np.zeros((st.nblocks, 2), np.uint32)
print('np.zeros status: %.1f us' % ((T() - t0) / 2000 * 1e6))
# the C calls with torch's synchronisation in front of each, as the wrappers have it
eg = dg = 0.0
t0 = T()
for _ in range(reps):
    torch.cuda.current_stream(t.device).synchronize()
    L.flacgpu_encode_streams(a._h, C.byref(s), t.data_ptr(), 0, descs, 1, o.data_ptr(), o.numel(), f.data_ptr(), C.byref(est))
    torch.cuda.current_stream(t.device).synchronize()
    L.flacgpu_decode_stream_dev(a._h, data.data_ptr(), est.total_bytes, est.nblocks, 0, 2, 16, dec.data_ptr(), n, stat.ctypes.data, stat.shape[0], None, C.byref(dst))
    eg += est.total_gpu_ms; dg += dst.total_gpu_ms
wall = (T() - t0) / reps * 1e3
print('C calls + torch syncs:     wall %.3f ms, GPU %.3f, rest %.3f' % (wall, (eg + dg) / reps, wall - (eg + dg) / reps))
# the wrappers with the synchronisation taken out
class _NoSync:
    def synchronize(self): pass
real = torch.cuda.current_stream
torch.cuda.current_stream = lambda *a_, **k_: _NoSync()
eg = dg = 0.0
t0 = T()
for _ in range(reps):
    o, f, st = a.encode(s, t, out=o, offsets=f)
    dec, status, ds = a.decode_stream(o[:st.total_bytes], 2, 16, n, nframes=st.nblocks, out=dec)
    eg += st.total_gpu_ms; dg += ds.total_gpu_ms
wall = (T() - t0) / reps * 1e3
torch.cuda.current_stream = real
print('wrappers without the sync: wall %.3f ms, GPU %.3f, rest %.3f' % (wall, (eg + dg) / reps, wall - (eg + dg) / reps))
