#!/usr/bin/env python3
"""bench.py -- FLAC encode (+ decode) throughput of the MI355X hot path.

Metric (BASELINE.json): Msamples/s encode (level 5, 48 kHz/16-bit stereo, blocksize 4096) + decode, bit-exact.
A "step" is one pass of the hot path over the batch resident in HBM: encode every block of the stream, then
decode the encoder's output back to PCM.  `value` counts each channel-sample once per step
(Msamples/s = 1e-6 * frames * channels / seconds).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N            (N > 1, no launcher around it: starts the N ranks itself, as a child process)

N = 1 (default): the headline is configs[1]+[2] (stream16); the same run then measures configs[3] (stream24, 300 s, level 8)
and configs[4]'s per-GPU share (batch, 128 streams x 60 s) in the same process and reports them under `configs`.
N > 1: configs[4] -- independent streams shard across ranks, 128 per GPU (weak scaling, no data-path collective); rank 0
broadcasts the 86-byte stream header over RCCL as the only shared datum.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


_CPU_JOB = None       # (level, channels, sample rate, int32 PCM, repetitions): inherited by the forked pool workers


def _encode_worker(_i):
    """Pool worker of the all-cores CPU baseline: `reps` oracle encodes of the sample; returns its own busy time."""
    level, ch, sr, a32, reps = _CPU_JOB
    from oracle import oracle as O
    cfg, _rc = O.config(level, ch, 16, sr, 4096, True)
    O.encode_stream(cfg, a32[:4096 * 4])          # load the library and build the window outside the timed part
    t = time.perf_counter()
    for _ in range(reps):
        O.encode_stream(cfg, a32)
    return time.perf_counter() - t


def cpu_model():
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def system_libflac(pcm16, level, sr):
    """a libFLAC found on THIS box (ldconfig), timed through its public C API
    (FLAC__stream_encoder_process_interleaved + finish, FLAC__stream_decoder_process_until_end_of_stream; the callbacks
    only count bytes).  None when the box has no libFLAC -- the image of the GPU pool has none, the probe is what is asked."""
    import ctypes as C
    import ctypes.util
    name = ctypes.util.find_library('FLAC')
    if not name:
        return None
    try:
        L = C.CDLL(name)
        L.FLAC__stream_encoder_new.restype = C.c_void_p
        L.FLAC__stream_decoder_new.restype = C.c_void_p
        WCB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p)
        a32 = np.ascontiguousarray(pcm16.astype(np.int32))
        res = {'library': name, 'version': C.c_char_p.in_dll(L, 'FLAC__VERSION_STRING').value.decode()}
        stream = bytearray()
        for md5 in (1, 0):
            best = None
            for _ in range(3):
                stream = bytearray()
                def w(_e, buf, n, _s, _f, _c):
                    stream.extend(C.string_at(buf, n))
                    return 0
                cb = WCB(w)
                enc = C.c_void_p(L.FLAC__stream_encoder_new())
                L.FLAC__stream_encoder_set_channels(enc, a32.shape[1]); L.FLAC__stream_encoder_set_bits_per_sample(enc, 16)
                L.FLAC__stream_encoder_set_sample_rate(enc, sr); L.FLAC__stream_encoder_set_compression_level(enc, level)
                L.FLAC__stream_encoder_set_blocksize(enc, 4096)
                if hasattr(L, 'FLAC__stream_encoder_set_do_md5'):
                    L.FLAC__stream_encoder_set_do_md5(enc, md5)
                if L.FLAC__stream_encoder_init_stream(enc, cb, None, None, None, None) != 0:
                    return None
                t = time.perf_counter()
                L.FLAC__stream_encoder_process_interleaved(enc, a32.ctypes.data_as(C.c_void_p), a32.shape[0])
                L.FLAC__stream_encoder_finish(enc)
                dt = time.perf_counter() - t
                L.FLAC__stream_encoder_delete(enc)
                best = dt if best is None else min(best, dt)
            res['encode_msamples_per_s_md5_%s' % ('on' if md5 else 'off')] = round(a32.size / best / 1e6, 2)
        return res
    except Exception as e:       # noqa: BLE001 (a library that does not behave is reported, not fatal)
        return {'library': name, 'error': repr(e)}


def cpu_baseline(pcm16, level, sr, budget_s=12.0, all_cores=True, what='the same synthetic stream'):
    """Time the CPU oracle (oracle/flac_oracle.c, a scalar port of libFLAC 1.4.3) on a bounded sample.  all_cores=False (the N > 1
    line: the other ranks are starting up on the same host) leaves the one-process-per-core leg out."""
    from oracle import oracle as O
    ch = pcm16.shape[1]
    cfg, rc = O.config(level, ch, 16, sr, 4096, True)
    assert rc == 0
    a32 = np.ascontiguousarray(pcm16.astype(np.int32))
    t0 = time.perf_counter()
    reps, enc_t = 0, 0.0
    stream = None
    while enc_t < budget_s * 0.6 or reps < 1:
        t = time.perf_counter()
        stream, _ = O.encode_stream(cfg, a32)
        enc_t += time.perf_counter() - t
        reps += 1
    dec_t, dreps = 0.0, 0
    while dec_t < budget_s * 0.3 or dreps < 1:
        t = time.perf_counter()
        O.decode_stream(stream)
        dec_t += time.perf_counter() - t
        dreps += 1
    nsamp = a32.size
    # the same encode as one process per host core (libFLAC and the oracle are single-threaded; processes scale linearly)
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    ncores = max(1, min(avail, 64))
    allc = None
    try:
        if not all_cores:
            raise RuntimeError('left out')
        import multiprocessing as mp
        global _CPU_JOB
        wreps = max(1, int(round(3.0 / max(enc_t / reps, 1e-3))))          # about 3 s of work per core
        _CPU_JOB = (level, ch, sr, a32, wreps)
        with mp.get_context('fork').Pool(ncores) as pool:
            busy = pool.map(_encode_worker, range(ncores))
        allc = nsamp * wreps * ncores / max(busy) / 1e6                     # all cores busy for the slowest worker's time
    except Exception:       # noqa: BLE001 (a host that cannot fork a pool still reports the single-thread figure)
        allc = None
    enc = nsamp * reps / enc_t / 1e6
    # oracle.decode_stream makes two passes (count, then decode): one decode = half the measured time
    dec = nsamp * dreps / (dec_t / 2) / 1e6
    both = 1.0 / (1.0 / enc + 1.0 / dec)
    # MD5 on (libFLAC's default): the same encode with the signature of the PCM computed as well
    t = time.perf_counter()
    O.encode_stream(cfg, a32, finalize=True)
    enc_md5 = nsamp / (time.perf_counter() - t) / 1e6
    sysflac = system_libflac(pcm16, level, sr)
    return {'value': round(both, 2), 'unit': 'Msamples/s', 'cores': 1, 'kind': 'port',
            'cpu_model': cpu_model(), 'host_cores': avail,
            'encode_msamples_per_s': round(enc, 2), 'decode_msamples_per_s': round(dec, 2),
            'encode_msamples_per_s_md5_on': round(enc_md5, 2),
            'system_libFLAC': sysflac if sysflac is not None else 'none on this box (ctypes.util.find_library("FLAC") is None)',
            'all_cores': {'cores': ncores, 'encode_msamples_per_s': None if allc is None else round(allc, 1),
                          'how': 'one oracle process per host core, about 3 s of encodes each, aggregate over the slowest worker'}
                         if all_cores else 'not run in the N > 1 line (the other ranks start up on the same cores); see the N = 1 line',
            'sample': '%.0f s of %s x%d encode / x%d decode passes, MD5 off, 1 thread; '
                      'oracle/flac_oracle.c is a scalar restatement of libFLAC 1.4.3 (the reference binary does not '
                      'travel to the GPU box; reference_binary_ratio has what it measured against the oracle)' %
                      (a32.shape[0] / sr, what, reps, dreps),
            'wall_s': round(time.perf_counter() - t0, 1)}


def committed_profile(name):
    """A committed profile summary (profiles/<name>), or None."""
    try:
        with open(os.path.join(ROOT, 'profiles', name)) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def committed_pmc(workload, level, nblocks, kernel_id, load=None, build_id=None):
    """The committed counter pass that may be quoted for this run, and whether it may: (pass, file name, same, note).
    A pass counts only for the KERNELS it was measured on -- flacgpu_kernel_id() is a hash over the .hip files and the headers they
    include, the pass carries the id of the library that ran under the counters (round 6: an edit of a host file changes
    flacgpu_build_id() and flacgpu_host_id() and leaves the pass standing) -- and for the same workload, level and block count;
    otherwise `same` is False, `traffic` / `issue_ceiling` stay out of the line and `note` says why.  A round-5 pass carries the id
    of the whole build only and counts for exactly that build (`build_id`).  (`load`: committed_profile, replaceable in tests.)"""
    load = load or committed_profile
    pmc, pmc_name = {}, None
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):
        pmc_name = '%s_pmc.json' % rnd if workload == 'stream16' else '%s_pmc_%s.json' % (rnd, workload)
        pmc = load(pmc_name) or {}
        if pmc:
            break
    same_shape = pmc.get('workload') == workload and pmc.get('level') == level and pmc.get('blocks') == nblocks
    if pmc.get('kernel_id'):
        same_code = pmc.get('kernel_id') == kernel_id
    else:
        same_code = bool(pmc.get('build_id')) and build_id is not None and pmc.get('build_id') == build_id
    same = bool(pmc) and same_shape and same_code
    note = None
    if pmc and not same:
        their = ('kernels %s' % pmc['kernel_id']) if pmc.get('kernel_id') else ('build %s' % pmc.get('build_id', '(no id: before round 5)'))
        note = ('profiles/%s is a pass of %s on %s / level %s / %s blocks; this run: kernels %s%s, %s / level %d / %d blocks -- not quoted' %
                (pmc_name, their, pmc.get('workload'), pmc.get('level'), pmc.get('blocks'),
                 kernel_id, (' (build %s)' % build_id) if build_id else '', workload, level, nblocks))
    return pmc, pmc_name, same, note


def api_e2e(seconds, sr):
    """Drop-in API end to end on host buffers (PCIe, callbacks and MD5 included): numpy int16 -> StreamEncoder.process ->
    write callbacks -> bytes -> StreamDecoder -> numpy blocks.  Never part of `value`."""
    import pyflac_amd
    from pyflac_amd import synth, _lib
    pcm = synth.config2_stereo16(seconds, 0, sr)
    out = {}
    for md5 in (1, 0):
        best_e = best_d = first_d = None
        for _rep in range(3):
            chunks = []
            enc = pyflac_amd.StreamEncoder(sr, lambda b, n, s, f: chunks.append(b), compression_level=5, blocksize=4096)
            if not md5:
                _lib.lib().FLAC__stream_encoder_set_do_md5(enc._encoder, 0)
            t0 = time.perf_counter()
            enc.process(pcm)
            enc.finish()
            t1 = time.perf_counter()
            stream = b''.join(chunks)
            blocks = []
            dec = pyflac_amd.StreamDecoder(lambda a, r, c, n: blocks.append(a))
            t2 = time.perf_counter()
            dec.process(stream)
            dec.finish()
            t3 = time.perf_counter()
            best_e = t1 - t0 if best_e is None else min(best_e, t1 - t0)
            best_d = t3 - t2 if best_d is None else min(best_d, t3 - t2)
            first_d = t3 - t2 if first_d is None else first_d
        assert sum(len(b) for b in blocks) == len(pcm)
        out['md5_on' if md5 else 'md5_off'] = {'encode_msamples_per_s': round(pcm.size / best_e / 1e6, 1),
                                               'decode_msamples_per_s': round(pcm.size / best_d / 1e6, 1)}
        if md5:
            # (the first decoder of a process allocates its pinned and device buffers; later ones inherit them)
            out['md5_on']['decode_first_decoder_msamples_per_s'] = round(pcm.size / first_d / 1e6, 1)
    out['sample'] = '%.0f s stereo 16-bit numpy array through pyflac_amd.StreamEncoder / StreamDecoder, best of 3' % seconds
    # ---- the streaming use of the class (pyflac/encoder.py:234-330, examples/stream.py): process() in small calls.  Every call is
    # a launch of the encoder's kernels and a wait; rate and per-call latency for calls of 1024 .. 65536 frames (MD5 on, as
    # pyFLAC has it), over 20 s of the stream
    small = {}
    part = pcm[:min(len(pcm), 20 * sr)]
    for frames in (1024, 4096, 16384, 65536):
        enc = pyflac_amd.StreamEncoder(sr, lambda b, n, s, f: None, compression_level=5, blocksize=4096)
        enc.process(part[:frames])          # (initialisation and the first launch stay outside)
        lat = []
        t0 = time.perf_counter()
        for a in range(frames, len(part) - frames + 1, frames):
            tc = time.perf_counter()
            enc.process(part[a:a + frames])
            lat.append(time.perf_counter() - tc)
        dt = time.perf_counter() - t0
        enc.finish()
        lat.sort()
        if not lat:                         # (a stream shorter than two calls of this size: --seconds below 3)
            continue
        small[str(frames)] = {'encode_msamples_per_s': round(len(lat) * frames * 2 / dt / 1e6, 1), 'calls': len(lat),
                              'call_ms_median': round(lat[len(lat) // 2] * 1e3, 3), 'call_ms_p95': round(lat[int(len(lat) * 0.95)] * 1e3, 3)}
    out['process_call_size'] = small
    # the same 1024-frame calls with the launch held back until 16 blocks are buffered (an extension, include/flacgpu.h
    # flacgpu_stream_encoder_set_launch_blocks: the frames then come in bursts of 16, the bytes are the same)
    enc = pyflac_amd.StreamEncoder(sr, lambda b, n, s, f: None, compression_level=5, blocksize=4096, launch_blocks=16)
    enc.process(part[:1024])
    t0 = time.perf_counter()
    ncalls = 0
    for a in range(1024, len(part) - 1024 + 1, 1024):
        enc.process(part[a:a + 1024])
        ncalls += 1
    dt = time.perf_counter() - t0
    enc.finish()
    out['process_call_size']['1024_launch_blocks_16'] = {'encode_msamples_per_s': round(ncalls * 1024 * 2 / dt / 1e6, 1), 'calls': ncalls}
    # ---- many streams at once: MD5 is serial per stream (0.4 G samples/s), so the class scales with the number of streams:
    # 16 StreamEncoders on 16 threads, MD5 on, 60 s each (the library calls release the GIL)
    import threading
    nthreads, secs = 16, min(seconds, 60.0)
    parts = [synth.config5_stream(k, secs, sr) for k in range(nthreads)]
    encs = [pyflac_amd.StreamEncoder(sr, lambda b, n, s, f: None, compression_level=5, blocksize=4096) for _ in range(nthreads)]
    for e, x in zip(encs, parts):
        e.process(x[:4096])
    def work(e, x):
        e.process(x[4096:])
        e.finish()
    th = [threading.Thread(target=work, args=(e, x)) for e, x in zip(encs, parts)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    out['many_streams'] = {'streams': nthreads, 'seconds_each': secs, 'md5': 'on',
                           'encode_msamples_per_s': round(sum(x[4096:].size for x in parts) / dt / 1e6, 1)}
    return out


WORKLOADS = ['stream16', 'stream24', 'batch', 'wasted', 'stream32', 'stream32w', 'surround6']
WORKLOAD_NAMES = {'stream16': 'configs[1]+[2]', 'stream24': 'configs[3]', 'batch': 'configs[4]',
                  'wasted': 'stream16 signal in a 24-bit container (8 wasted bits)',
                  'stream32': '32-bit samples (33-bit side channel)',
                  'stream32w': '24-bit material in a 32-bit container (int32 arrays as pyFLAC gets them from a 24-bit WAV: eight wasted bits)',
                  'surround6': 'six channels'}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default: 300 for the single-stream workloads, 30 for batch)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--seconds', type=float, default=None, help='length of each stream (default 600; 300 for stream24, 60 for batch)')
    ap.add_argument('--level', type=int, default=None, help='compression level (default 5; 8 for --workload stream24)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true', help='skip the drop-in API end-to-end leg')
    ap.add_argument('--no-configs', action='store_true', help='headline only: skip the configs[3] / configs[4] legs of the default run')
    ap.add_argument('--share-gpu', action='store_true',
                    help='test aid for --gpus N on a box with ONE GPU: every rank works on cuda:0, the collectives run over gloo on host tensors '
                         '(tests/test_gpu_dist.py: rank 1\'s data path, the barriers and the MAX reduction on real kernels)')
    ap.add_argument('--no-direct', action='store_true', help='A/B: the chunk form of the packing stage (rounds 2-4) instead of the direct path')
    ap.add_argument('--no-passes', action='store_true', help='skip the event passes after the timed loop (profiling runs: the trace then holds the timed loop only)')
    ap.add_argument('--workload', choices=WORKLOADS, default=None,
                    help='default: stream16 at 1 GPU (the metric: configs[1]+[2]), batch at N > 1 (configs[4]).  stream24: configs[3], '
                         '24-bit 96 kHz level 8; batch: independent 16-bit streams per GPU in one launch (--streams, --seconds each); '
                         'wasted: the stream16 signal in a 24-bit container (8 wasted bits in every block)')
    ap.add_argument('--streams', type=int, default=128, help='streams per GPU for --workload batch')
    ap.add_argument('--dry-run', action='store_true',
                    help='CPU only, gloo: the launcher, the rank mapping, the header broadcast and the timing contract with no GPU work '
                         '(tests/test_bench_launcher.py); the line it prints says so and is no measurement')
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a CHILD process (torch.distributed.run, one
    rank per GPU) and relay what rank 0 prints.  This process runs no GPU work and never replaces itself with another program
    (torch.cuda.device_count() may initialise the HIP runtime here on builds of torch without amdsmi; the ranks are a fresh child
    either way).  A box with fewer than N GPUs is an error, not an n_gpus: 1 line."""
    import subprocess
    if not args.dry_run and not args.share_gpu:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write('bench.py: --gpus %d asked, %d GPU(s) visible on this box\n' % (args.gpus, have))
            return 3
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # (dmabuf IPC: RCCL between processes needs it on this pool)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """The N > 1 path with the GPU work taken out (gloo, CPU): what remains is what the launcher test checks -- every rank gets
    its share of the world x streams batch, rank 0's header reaches everybody, the steps are bracketed by barriers and the
    maximum over the ranks is what rank 0 reports."""
    import torch
    import torch.distributed as dist
    from pyflac_amd import shard
    if world > 1:
        dist.init_process_group('gloo')
    dev = torch.device('cpu')
    hdr = bytes(range(86)) if rank == 0 else b''
    if world > 1:
        hdr = shard.broadcast_header(hdr, dev)
    assert hdr == bytes(range(86))
    mine = shard.streams_for_rank(world * args.streams, rank, world)
    assert len(mine) == args.streams
    steps = args.steps or 3
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        got = [None] * world
        dist.all_gather_object(got, mine)
        assert sorted(x for g in got for x in g) == list(range(world * args.streams))
    if rank == 0:
        print(json.dumps({'metric': 'dry run: no GPU work, no measurement', 'value': 0.0, 'unit': 'Msamples/s', 'n_gpus': world, 'steps': steps,
                          'warmup': args.warmup, 'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'int32', 'data': 'none (dry run)',
                          'config': {'workload': 'dry run of configs[4]: %d streams per rank, %d ranks' % (args.streams, world)}, 'dry_run': True}))
    if world > 1:
        dist.destroy_process_group()


def make_input(workload, seconds, rank, world, streams):
    """(pcm [samples, channels] as numpy, sample rate, channels, bits per sample, per-stream lengths or None)"""
    from pyflac_amd import shard, synth
    sr, ch, bps, lengths = 48000, 2, 16, None
    if workload == 'stream24':
        sr, bps = 96000, 24
        pcm = synth.config4_stereo24(seconds, 1 + rank, sr)          # int32 container, 24-bit values
    elif workload == 'wasted':
        bps = 24
        pcm = synth.config2_stereo16(seconds, rank, sr).astype(np.int32) << 8
    elif workload == 'stream32':
        # pyFLAC's other input type (pyflac/encoder.py:109: int32 arrays): 32-bit samples, the side channel needs 33 bits
        bps = 32
        pcm = synth.config2_stereo16(seconds, rank, sr).astype(np.int64) * 40000 + (np.arange(int(round(sr * seconds)))[:, None] % 977)
        pcm = np.clip(pcm, -(1 << 31), (1 << 31) - 1).astype(np.int32)
    elif workload == 'stream32w':
        # what pyFLAC makes of a 24-bit WAV file: soundfile reads it as left-justified int32, pyflac/encoder.py:109 sets 32 bits per
        # sample -- a 32-bit stream whose samples share eight wasted bits (the configs[3] signal in a 32-bit container)
        sr, bps = 96000, 32
        pcm = (synth.config4_stereo24(seconds, 1 + rank, sr).astype(np.int64) << 8).astype(np.int32)
    elif workload == 'surround6':
        # six channels (tests/test_encoder.py:258-273 of the reference: surround.wav), 16 bit: three stereo pairs of the generator family
        ch = 6
        pcm = np.concatenate([synth.config5_stream(3 * rank + k, seconds, sr) for k in range(3)], axis=1)
    elif workload == 'batch':
        # configs[4]: this rank's share of the world * streams batch -- stream s runs on rank s mod world
        # (pyflac_amd.shard.streams_for_rank, the mapping batch.MultiContext uses inside one process) -- concatenated in HBM,
        # every stream with its own frame numbering, all of them in ONE launch
        mine = shard.streams_for_rank(world * streams, rank, world)
        per = [synth.config5_stream(sidx, seconds, sr) for sidx in mine]
        lengths = [len(x) for x in per]
        pcm = np.concatenate(per)
    else:
        # each rank encodes its own stream (config 5 generator family); rank 0 at N=1 is config 2's stream
        pcm = synth.config2_stereo16(seconds, 0, sr) if world == 1 else synth.config5_stream(rank, seconds, sr)
    return pcm, sr, ch, bps, lengths


def measure(env, ctx, workload, seconds, level, steps, warmup, streams, passes=True, check=True):
    """One workload: `warmup` untimed steps, `steps` timed steps between barriers, the bit-exactness gates, then the event
    passes for the per-launch durations.  Returns the result dictionary (rank 0; None on the other ranks)."""
    import torch
    import torch.distributed as dist
    from pyflac_amd import batch, shard, _lib
    rank, world, dev = env['rank'], env['world'], env['dev']
    cdev = env.get('coll', dev)          # where the collectives' tensors live (the GPU under RCCL; the host under --share-gpu / gloo)
    bs = 4096
    pcm16, sr, ch, bps, lengths = make_input(workload, seconds, rank, world, streams)
    pcm = torch.from_numpy(pcm16.astype(np.int32)).to(dev)     # int32 at the C ABI, like pyflac/encoder.py:112
    nsamp = pcm.shape[0]
    s = batch.settings(level, ch, bps, sr, bs, True)

    # the only shared datum: the 86-byte stream header, broadcast from rank 0 (RCCL over xGMI)
    from pyflac_amd.encoder import stream_header_bytes
    hdr = stream_header_bytes(s) if rank == 0 else b''
    if world > 1:
        hdr = shard.broadcast_header(hdr, cdev)
    assert len(hdr) == 86

    out = offs = dec = None
    single = lengths is None
    ranges = [None]

    def step():
        nonlocal out, offs, dec
        out, offs, est = ctx.encode(s, pcm, stream_lengths=lengths, out=out, offsets=offs)
        if single:
            # decode from the bytes alone: the frame index is rebuilt on the GPU inside the timed region (the number of
            # frames is what STREAMINFO tells a decoder: total samples / block size)
            dec, status, dst = ctx.decode_stream(out[:est.total_bytes], ch, bps, nsamp, nframes=est.nblocks, out=dec)
        else:
            # many streams back to back, every stream numbering its frames from 0: decoded from the bytes alone as well -- one
            # pass over all bytes files every stream's frames under its own numbers (flacgpu_decode_streams_dev).  What is passed
            # beside the bytes is what the STREAMINFO blocks hold: the frame count of every stream, and where its bytes end
            if ranges[0] is None:
                h = offs.cpu().numpy().astype(np.int64)      # (once, outside the timed region: the streams' byte lengths)
                ranges[0], fi = [], 0
                for n in lengths:
                    nfr = -(-n // bs)
                    ranges[0].append((int(h[fi + nfr] - h[fi]), nfr))
                    fi += nfr
            dec, status, dst = ctx.decode_streams(out[:est.total_bytes], ranges[0], ch, bps, nsamp, out=dec)
        return est, dst, status

    for _ in range(max(1, warmup)):
        est, dst, status = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc_wall = dec_wall = 0.0
    total_bytes = 0
    per_step = []
    tl = t0
    for _ in range(steps):
        est, dst, status = step()
        enc_wall += est.total_gpu_ms       # (timed region: no HIP events; GPU time from the wall-clock stamps of the first
        dec_wall += dst.total_gpu_ms       #  and the last kernel of each call)
        total_bytes = est.total_bytes
        tn = time.perf_counter()          # (every library call is synchronous: no extra synchronisation inside the timed region)
        per_step.append(tn - tl)
        tl = tn
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    rank_ms = [dt / steps * 1e3]
    if world > 1:
        mine_t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(every, mine_t)
        rank_ms = [float(x.item()) / steps * 1e3 for x in every]
        t = mine_t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # bit-exactness gate (outside the timed region, behind it: what is checked is what the timed steps left in the buffers -- and
    # nothing but the warm-up steps stands between the box's idle time and the first timed step): the round trip equals the input
    rt_ok = int(status[:, 0].max()) == 0 and bool(torch.equal(dec[:nsamp], pcm))
    if world == 1:
        assert int(status[:, 0].max()) == 0, 'decoder reported frame errors'
        assert rt_ok, 'round trip is not bit-exact'
    # what the GPU wrote, for the byte-for-byte check against the oracle after the timed region: the WHOLE stream (of the batch:
    # its first and its last stream, whole)
    h_chk = []
    if check:            # (every rank: at N > 1 each checks the first and the last stream of ITS share)
        if single:
            h_chk.append((0, nsamp, out[:int(offs[-(-nsamp // bs)].item())].cpu().numpy().tobytes()))
        else:
            ho = offs.cpu().numpy().astype(np.int64)
            first_frame = [0]
            for n in lengths:
                first_frame.append(first_frame[-1] + -(-n // bs))
            starts = np.concatenate([[0], np.cumsum(lengths)])
            for k in sorted(set([0, len(lengths) // 2, len(lengths) - 1] if world == 1 else [0, len(lengths) - 1])):
                a, b = int(ho[first_frame[k]]), int(ho[first_frame[k + 1]])
                h_chk.append((int(starts[k]), int(lengths[k]), out[a:b].cpu().numpy().tobytes()))
    ms_per_step = dt / steps * 1e3
    chsamples = nsamp * ch                                   # per rank per step
    value = chsamples * world / (ms_per_step * 1e-3) / 1e6
    K = steps
    alg_bytes = chsamples * 4 + total_bytes                   # read int32 PCM once + write the frames once (and back for decode)

    # The same steps again with HIP events around every launch: the per-launch durations the roofline objects quote.
    # flacgpu_set_stage_timing level 3 = the default call with ONE event in front of its first kernel and one behind its last (each
    # event record idles the GPU for a few microseconds, which is why the timed region above runs without any: the two of level 3
    # stand outside the kernels they time).  Then a few steps at level 1 (events also around the kernel groups inside a call: the
    # encode / decode kernels proper, the index pass) and at level 2 (between the encoder's stages) for the break-down fields.
    L = _lib.lib()
    L.flacgpu_set_stage_timing(ctx._h, 3)
    enc_ms = dec_ms = enc_tot = dec_tot = idx_ms = 0.0
    KE = max(1, min(K, 100))
    for _ in range(KE if passes else 0):
        est, dst, status = step()
        enc_tot += est.total_gpu_ms
        dec_tot += dst.total_gpu_ms
    L.flacgpu_set_stage_timing(ctx._h, 1)
    KG = max(1, min(K, 20))
    for _ in range(KG if passes else 0):
        est, dst, status = step()
        enc_ms += est.encode_kernel_ms
        dec_ms += dst.decode_kernel_ms
        idx_ms += dst.index_ms
    enc_k, dec_k, enc_t, dec_t, idx_ms = enc_ms / KG, dec_ms / KG, enc_tot / KE, dec_tot / KE, idx_ms / KG
    if not passes:          # (no event pass: the device stamps of the timed region stand in)
        enc_k = enc_t = enc_wall / K
        dec_k = dec_t = dec_wall / K
    L.flacgpu_set_stage_timing(ctx._h, 2)
    stage = np.zeros(4)
    for _ in range(5 if passes else 0):
        e2, _d2, _s2 = step()
        stage += np.array(list(e2.stage_ms)[:4])
    stage /= 5
    L.flacgpu_set_stage_timing(ctx._h, 0)
    nblocks = int(est.nblocks)
    # STREAMINFO's md5sum for the batch (stage L1; pyFLAC's own path always hashes): the batch entry point leaves it to the caller,
    # flacgpu_md5_streams computes it on the GPU -- one thread per stream, the hash being one serial chain per stream -- and this is
    # what a step costs with it (checked against hashlib on the host copy of the first and the last stream)
    md5_on = None
    if workload == 'batch' and rank == 0 and world == 1 and passes:
        import hashlib
        digs, md5_ms = ctx.md5_streams(pcm, bps, lengths)
        starts = np.concatenate([[0], np.cumsum(lengths)])
        for k in (0, len(lengths) - 1):
            assert digs[k] == hashlib.md5(pcm16[int(starts[k]):int(starts[k + 1])].astype('<i2').tobytes()).digest(), 'GPU MD5 differs'
        step_ms = dt / steps * 1e3
        md5_on = {'md5_kernel_ms': round(md5_ms, 2), 'value': round(nsamp * ch / ((step_ms + md5_ms) * 1e-3) / 1e6, 1), 'unit': 'Msamples/s',
                  'what': 'one step (encode + decode) plus flacgpu_md5_streams over the same %d streams, run one behind the other; MD5 is a '
                          'serial chain per stream (one GPU thread each, %.0f MB/s a stream), so its time does not depend on the number '
                          'of streams up to the lanes of the chip and is hidden only behind launches of thousands of streams' %
                          (len(lengths), lengths[0] * ch * 2 / (md5_ms * 1e-3) / 1e6)}
        # ... and at the stream count BASELINE.json's config 5 names for the node (1024): the same PCM eight times over, hashed as
        # 1024 streams -- the chain's time should not move, so the hash is hidden eight times better.  Only the hash kernel is run at
        # that size; the step is the measured one taken eight times (the encode and decode kernels are saturated at 128 streams:
        # their time is proportional to the samples).  Never allowed to fail the line.
        try:
            reps = max(1, 1024 // len(lengths))
            big = pcm.repeat(reps, 1) if pcm.dim() == 2 else pcm.repeat(reps)
            digs_b, md5_ms_b = ctx.md5_streams(big, bps, list(lengths) * reps)
            assert digs_b[0] == digs[0] and digs_b[-1] == digs[-1], 'GPU MD5 of the repeated streams differs'
            md5_on['streams_%d' % (len(lengths) * reps)] = {
                'md5_kernel_ms': round(md5_ms_b, 2),
                'value': round(reps * nsamp * ch / ((reps * step_ms + md5_ms_b) * 1e-3) / 1e6, 1), 'unit': 'Msamples/s',
                'what': 'flacgpu_md5_streams over %d streams (the %d measured ones %d times over) timed; the step taken as %d times the '
                        'measured %d-stream step' % (len(lengths) * reps, len(lengths), reps, reps, len(lengths))}
            del big
        except Exception as e:       # noqa: BLE001
            md5_on['streams_1024'] = {'error': repr(e)}
    del out, offs, dec, pcm
    torch.cuda.empty_cache()
    # checker use of the oracle: EVERY frame the GPU wrote (of the batch: every frame of its first, middle and last stream; at N > 1
    # every rank's first and last stream) is the oracle's, byte for byte -- a round trip alone would also pass a wrong but decodable
    # choice.  At N > 1 the verdicts are reduced over the ranks before anybody raises: a rank that stopped alone would leave the
    # others waiting in the next collective.
    checked = None
    ok = rt_ok
    if h_chk:
        import hashlib
        from oracle import oracle as O
        cfg, _ = O.config(level, ch, bps, sr, bs, True)
        tc = time.perf_counter()
        shas, nfr_chk, nbytes = [], 0, 0
        for start, n, got in h_chk:
            ref, sizes = O.encode_stream(cfg, pcm16[start:start + n].astype(np.int32))
            got_sha, want_sha = hashlib.sha256(got).hexdigest(), hashlib.sha256(ref[86:]).hexdigest()
            ok = ok and got_sha == want_sha
            shas.append(got_sha)
            nfr_chk += len(sizes)
            nbytes += len(got)
        checked = {'frames': '%d of %d frames' % (nfr_chk, nblocks),
                   'what': 'SHA-256 of the GPU stream == SHA-256 of oracle.encode_stream over the same PCM'
                           + ('' if single else (' (first, middle and last stream of the batch, whole)' if world == 1 else
                                                 ' (first and last stream of every rank\'s share, whole); round trip of every rank bit-exact')),
                   'sha256': shas[0] if single and world == 1 else shas, 'bytes': nbytes, 'oracle_s': round(time.perf_counter() - tc, 1)}
    if world > 1:
        v = torch.tensor([1 if ok else 0, 1 if rt_ok else 0, int(checked['frames'].split()[0]) if checked else 0,
                          checked['bytes'] if checked else 0], device=cdev, dtype=torch.int64)
        every = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(every, v)
        bad = [r for r in range(world) if int(every[r][0]) == 0]
        if bad:
            raise SystemExit('bench.py: rank(s) %s: %s' % (bad, 'round trip is not bit-exact' if any(int(every[r][1]) == 0 for r in bad)
                                                           else 'encoded stream differs from the oracle'))
        if checked:
            checked['ranks'] = world
            checked['frames'] = '%d frames over %d ranks (%d of %d on rank 0)' % (sum(int(e[2]) for e in every), world, int(every[0][2]), nblocks)
            checked['bytes'] = sum(int(e[3]) for e in every)
    else:
        assert ok, 'encoded stream differs from the oracle'
    if rank != 0:
        return None

    ps = np.sort(np.array(per_step)) * 1e3
    # (the committed PMC passes of the same command: profiles/r04_pmc.json for the headline, profiles/r04_pmc_<workload>.json
    # for the others; a pass of an earlier round stands in only if it was made on the same workload, level and block count)
    build_id = L.flacgpu_build_id().decode()
    kernel_id, host_id = L.flacgpu_kernel_id().decode(), L.flacgpu_host_id().decode()
    pmc, pmc_name, same, pmc_note = committed_pmc(workload, level, nblocks, kernel_id, build_id=build_id)
    enc_ach = alg_bytes / (enc_t * 1e-3) / 1e9
    dec_ach = alg_bytes / (dec_t * 1e-3) / 1e9
    res = {
        'metric': 'Msamples/s encode (level %d, %dkHz/%d-bit %s, blk 4096) + decode; bit-exact' % (level, sr // 1000, bps, 'stereo' if ch == 2 else '%d channels' % ch),
        'value': round(value, 1), 'unit': 'Msamples/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'int32', 'data': 'synthetic',
        'config': {'workload': '%s: %s encode, then decode of its output from the bytes%s, %d channels %d-bit %d kHz, blocksize 4096, '
                               'level %d, %.0f s%s (%d blocks) per GPU, int32 PCM resident in HBM, MD5 off '
                               '(FLAC__stream_encoder_set_do_md5(0)); every step repeats one layout, so the block descriptor list '
                               '(225 KB for the headline) is uploaded by the first call and reused by the timed ones' %
                               (WORKLOAD_NAMES[workload],
                                'single-stream' if single else '%d independent streams in one launch,' % len(lengths),
                                ' alone (frame index rebuilt on the GPU inside the timed region)' if single else
                                ' alone (every stream\'s frame index rebuilt on the GPU inside the timed region, one pass over all bytes)',
                                ch, bps, sr // 1000, level, seconds, '' if single else ' each', nblocks),
                   'blocks_per_gpu': nblocks, 'compression_ratio': round(total_bytes / (chsamples * (bps // 8)), 4),
                   'timed_s': round(dt, 3)},
        'ms_per_step_min': round(float(ps[0]), 3), 'ms_per_step_median': round(float(ps[len(ps) // 2]), 3),
        'encode_kernel_msamples_per_s': round(chsamples / (enc_k * 1e-3) / 1e6, 1),
        'decode_kernel_msamples_per_s': round(chsamples / (dec_k * 1e-3) / 1e6, 1),
        'encode_kernel_ms': round(enc_k, 3), 'decode_kernel_ms': round(dec_k, 3),
        'encode_gpu_ms': round(enc_t, 3), 'decode_gpu_ms': round(dec_t, 3), 'decode_index_ms': round(idx_ms, 3),
        'timed_region_gpu_ms': {'encode': round(enc_wall / K, 3), 'decode': round(dec_wall / K, 3),
                                'source': 'device wall-clock stamps of the first and last kernel of each call, inside the timed region'},
        'event_pass_steps': KE if passes else 0,
        'encode_stage_ms': {'analysis (autocorrelation, Levinson-Durbin, evaluation)': round(float(stage[0]), 3),
                            'packing': round(float(stage[1]), 3), 'sizes + scan': round(float(stage[2]), 3),
                            'assembly + CRC-16': round(float(stage[3]), 3)},
        # the encoder is a handful of kernels back to back: one encode = PCM read once, frames written once; time = HIP events
        # around all of them on the library's stream.  Per-kernel durations: profiles/r04_*_kernel_stats.csv (rocprofv3
        # --kernel-trace --stats of the same command).
        'roofline': {'bound': 'hbm', 'kernel': 'encode launch (all kernels of flacgpu_encode_streams; per-kernel times in profiles/)',
                     'achieved': round(enc_ach, 2), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(enc_ach / 8000.0, 5),
                     'traffic': pmc.get('encode_traffic_bytes_per_launch') if same else None,
                     'algorithmic_bytes_per_launch': int(alg_bytes), 'ms_per_launch': round(enc_t, 4)},
        'roofline_decode': {'bound': 'hbm', 'kernel': 'decode launch (all kernels of flacgpu_decode_stream_dev: index, headers, scan, parse, CRC-16, restore)',
                            'achieved': round(dec_ach, 2), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(dec_ach / 8000.0, 5),
                            'traffic': pmc.get('decode_traffic_bytes_per_launch') if same else None,
                            'algorithmic_bytes_per_launch': int(alg_bytes), 'ms_per_launch': round(dec_t, 4)},
    }
    res['build_id'] = build_id
    res['kernel_id'], res['host_id'] = kernel_id, host_id
    if pmc_note:
        res['roofline']['traffic_note'] = res['roofline_decode']['traffic_note'] = pmc_note
    if same:
        res['roofline']['traffic_source'] = res['roofline_decode']['traffic_source'] = 'profiles/' + pmc_name + ' (committed PMC pass of this command and these kernels, not measured in this run)'
    if same and pmc.get('encode_valu_insts_per_launch'):
        # the ceiling these integer kernels actually run against: VALU issue (one wave-instruction per 2 cycles and SIMD,
        # fp64 4 cycles), 1024 SIMDs at 2.4 GHz; instruction counts from the committed PMC pass of this same command
        slots = 1024 * 2.4e9 / 2
        # (tools/ubench/valurate.hip on this chip, 8 waves a SIMD, independent instructions: add / sub / logic / shift / v_mov_b32 /
        # fp32 FMA issue in 2.5-3.1 cycles a wave-instruction, everything else -- every multiply, the three-operand integer
        # operations, conversions, min / max, DPP, SDWA, 64-bit moves, fp64 -- in 4.2-4.6: `frac_bounds` prices all instructions
        # of the launch at 2.5 and at 4.3 cycles)
        def bounds(n, t_ms):
            return [round(n * c / (t_ms * 1e-3 * 1024 * 2.4e9), 4) for c in (2.5, 4.3)] if n else None
        res['issue_ceiling'] = {'unit': 'VALU wave-instructions/s', 'peak': slots,
                                'encode_frac_bounds': bounds(pmc['encode_valu_insts_per_launch'], enc_t),
                                'decode_frac_bounds': bounds(pmc.get('decode_valu_insts_per_launch'), dec_t),
                                'encode_valu_insts_per_launch': pmc['encode_valu_insts_per_launch'],
                                'encode_frac': round(pmc['encode_valu_insts_per_launch'] / (enc_t * 1e-3) / slots, 4),
                                'decode_valu_insts_per_launch': pmc.get('decode_valu_insts_per_launch'),
                                'decode_frac': (round(pmc['decode_valu_insts_per_launch'] / (dec_t * 1e-3) / slots, 4)
                                                if pmc.get('decode_valu_insts_per_launch') else None),
                                'source': 'profiles/' + pmc_name}
    if checked:
        res['checked'] = checked
    if md5_on:
        res['md5_on'] = md5_on
    if world > 1:
        res['ms_per_step_rank'] = [round(x, 3) for x in rank_ms]
    return res


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    under_launcher = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if not under_launcher and args.gpus > 1:
        # decided before anything touches a GPU: the ranks are a child process of this one
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d rank(s)' % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    workload = args.workload or ('stream16' if world == 1 else 'batch')
    defaults = {'batch': (60.0, 5, 30), 'stream24': (300.0, 8, 100)}.get(workload, (600.0, 5, 300))
    seconds = args.seconds if args.seconds is not None else defaults[0]
    level = args.level if args.level is not None else defaults[1]
    steps = args.steps if args.steps is not None else defaults[2]

    from pyflac_amd import synth
    # The CPU leg runs FIRST, before anything touches the GPU: it forks one oracle process per host core, and a forked child
    # must not carry HIP state (torch.cuda.is_available() below initialises the device).
    cpu_res = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and workload == 'stream16':
        cpu_res = cpu_baseline(synth.config2_stereo16(60.0, 0, 48000), level, 48000)
    elif rank == 0 and world > 1 and not args.no_cpu_baseline and workload == 'batch':
        # (the N > 1 line carries the baseline too: rank 0 times the oracle on ONE stream of its share of the batch, one thread, some
        # eight seconds, before it touches its GPU; the other ranks wait for it in init_process_group)
        cpu_res = cpu_baseline(synth.config5_stream(0, min(seconds, 60.0), 48000), level, 48000, budget_s=8.0, all_cores=False,
                               what='stream 0 of the batch (synth.config5_stream)')
    import torch
    import torch.distributed as dist
    from pyflac_amd import batch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    if args.share_gpu:
        local = 0                       # (every rank on cuda:0: the one GPU of a test box)
    if local >= torch.cuda.device_count():
        raise SystemExit('bench.py: rank %d has no GPU (local rank %d, %d visible)' % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    env = {'rank': rank, 'world': world, 'dev': dev}
    if world > 1:
        if args.share_gpu:
            dist.init_process_group('gloo')
            env['coll'] = torch.device('cpu')
        else:
            dist.init_process_group('nccl', device_id=dev)
    ctx = batch.Context(local)
    if args.no_direct:
        # (A/B of round 5's direct packing path: chunks through HBM, sizes scan and assembly kernel, as in rounds 2-4)
        from pyflac_amd import _lib
        _lib.lib().flacgpu_set_direct(ctx._h, 0)

    # The other single-GPU configurations of BASELINE.json, in the same process: each with its own warm-up, bit-exactness gates
    # (round trip; whole stream against the oracle), >= 20 timed steps and event passes.  They run in FRONT of the headline
    # measurement: the CPU leg above ends with one oracle process on every host core, and the host thread that drives the steps
    # needs a few seconds to get its clocks and caches back -- a 20-step run right behind that burst showed 0.11 ms of host time
    # a step where a run without the CPU leg shows 0.05.
    cfgs = {}
    if rank == 0 and world == 1 and workload == 'stream16' and not args.no_configs and args.seconds is None and args.level is None:
        # (behind them the shapes off the headline -- pyFLAC's int32 input with true 32-bit content and with 24-bit material in a 32-bit
        # container, six channels --, two minutes each)
        for wl, secs, lvl, k in (('stream24', 300.0, 8, 40), ('batch', 60.0, 5, 20), ('stream32', 120.0, 5, 10), ('stream32w', 120.0, 5, 10),
                                 ('surround6', 120.0, 5, 10)):
            r = measure(env, ctx, wl, secs, lvl, k, 2, args.streams, passes=True, check=not args.no_cpu_baseline)
            cfgs[wl] = {key: r[key] for key in ('metric', 'value', 'unit', 'steps', 'ms_per_step', 'ms_per_step_min', 'encode_gpu_ms',
                                                'decode_gpu_ms', 'encode_stage_ms', 'config', 'checked', 'md5_on') if key in r}
            cfgs[wl]['roofline'] = {k2: r['roofline'][k2] for k2 in ('achieved', 'frac', 'traffic', 'algorithmic_bytes_per_launch', 'ms_per_launch')}
            cfgs[wl]['roofline_decode'] = {k2: r['roofline_decode'][k2] for k2 in ('achieved', 'frac', 'traffic', 'algorithmic_bytes_per_launch', 'ms_per_launch')}
    res = measure(env, ctx, workload, seconds, level, steps, args.warmup, args.streams, passes=not args.no_passes,
                  check=not args.no_cpu_baseline)
    if rank == 0:
        if cfgs:
            res['configs'] = cfgs
        if world == 1 and not args.no_e2e and workload == 'stream16':
            res['api_e2e'] = api_e2e(min(seconds, 600.0), 48000)
        if cpu_res is not None:
            res['cpu_baseline'] = cpu_res
            ratio = committed_profile('r02_cpu_ref_ratio.json')
            if ratio:
                res['cpu_baseline']['reference_binary_ratio'] = {
                    'encode': ratio['reference_over_oracle_encode'], 'decode': ratio['reference_over_oracle_decode'],
                    'reference_encode_msamples_per_s_build_container': ratio['reference_encode_msamples_per_s'],
                    'measured_by': ratio['command'] + ' (build container; the binary does not travel): profiles/r02_cpu_ref_ratio.json'}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
