"""How much do the encoder's kernels gain from running beside each other?  Two (or four) contexts on one device encode the same
stream from two threads: if the pair finishes in less than twice the time of one, kernels of different stages overlap usefully
(the matrix-core chains of the autocorrelation beside the VALU work of evaluation and packing).  Experiment, not product."""
import sys, os, time, threading, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflac_amd import batch, synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
dev = torch.device('cuda', 0)
s = batch.settings(5, 2, 16, 48000, 4096, True)
K = 100
ref = None
for nctx in (1, 2, 4, 8):
    ctxs = [batch.Context(0) for _ in range(nctx)]
    pcms = [torch.from_numpy(synth.config2_stereo16(secs, i, 48000).astype(np.int32)).to(dev) for i in range(nctx)]
    outs = [None] * nctx
    gpu_ms = [0.0] * nctx
    def work(i, k):
        o = f = None
        for _ in range(k):
            o, f, st = ctxs[i].encode(s, pcms[i], out=o, offsets=f)
            gpu_ms[i] += st.total_gpu_ms
        outs[i] = (o, f, st)
    for i in range(nctx):
        work(i, 3)
    torch.cuda.synchronize()
    gpu_ms = [0.0] * nctx
    th = [threading.Thread(target=work, args=(i, K)) for i in range(nctx)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    shas = [hashlib.sha256(outs[i][0][:outs[i][2].total_bytes].cpu().numpy().tobytes()).hexdigest()[:12] for i in range(nctx)]
    if ref is None: ref = shas[0]
    print('contexts %d: wall %.3f ms per round of %d encodes (%.3f ms per stream); device-stamp ms per call %s; sha[0] %s same-as-single %s, %d distinct' %
          (nctx, dt / K * 1e3, nctx, dt / K / nctx * 1e3, ['%.3f' % (g / K) for g in gpu_ms], shas[0], shas[0] == ref, len(set(shas))), flush=True)
