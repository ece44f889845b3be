#!/usr/bin/env python3
"""bench.py -- FLAC encode (+ decode) throughput of the MI355X hot path.

Metric (BASELINE.json): Msamples/s encode (level 5, 48 kHz/16-bit stereo, blocksize 4096) + decode, bit-exact.
A "step" is one pass of the hot path over the batch resident in HBM: encode every block of the stream, then
decode the encoder's output back to PCM.  `value` counts each channel-sample once per step
(Msamples/s = 1e-6 * frames * channels / seconds).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: independent streams shard across ranks (weak scaling, no data-path collective); rank 0 broadcasts the
86-byte stream header over RCCL as the only shared datum.
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


def cpu_baseline(pcm16, level, sr, budget_s=12.0):
    """Time the CPU oracle (oracle/flac_oracle.c, a scalar port of libFLAC 1.4.3) on a bounded sample."""
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
    return {'value': round(both, 2), 'unit': 'Msamples/s', 'cores': 1, 'kind': 'port',
            'encode_msamples_per_s': round(enc, 2), 'decode_msamples_per_s': round(dec, 2),
            'all_cores': {'cores': ncores, 'encode_msamples_per_s': None if allc is None else round(allc, 1),
                          'how': 'one oracle process per host core, about 3 s of encodes each, aggregate over the slowest worker'},
            'sample': '%.0f s of the same synthetic stream x%d encode / x%d decode passes, MD5 off, 1 thread; '
                      'oracle/flac_oracle.c is a scalar restatement of libFLAC 1.4.3 (the reference binary does not '
                      'travel to the GPU box; it measured ~1.5x the oracle in the build container)' %
                      (a32.shape[0] / sr, reps, dreps),
            'wall_s': round(time.perf_counter() - t0, 1)}


def pmc_traffic(args, est):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_traffic.json, written by
    tools/rocprof_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command); None when the file does
    not describe this workload."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_traffic.json')) as fh:
            t = json.load(fh)
        if (args.workload == 'stream16' and t.get('blocks') == int(est.nblocks) and t.get('level') == args.level and
                t.get('kernel') == 'fg_encode_fast_kernel'):
            return int(t['traffic_bytes_per_launch'])
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--seconds', type=float, default=600.0, help='length of the stream each GPU encodes')
    ap.add_argument('--level', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', choices=['stream16', 'stream24', 'batch'], default='stream16',
                    help='stream16 (default, the metric): configs[1]+[2]; stream24: configs[3], 24-bit 96 kHz (use --level 8); '
                         'batch: configs[4], independent 16-bit streams per GPU in one launch (--streams, --seconds each)')
    ap.add_argument('--streams', type=int, default=128, help='streams per GPU for --workload batch')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pyflac_amd import batch, synth

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)

    sr, ch, bps, bs = 48000, 2, 16, 4096
    lengths = None
    if args.workload == 'stream24':
        sr, bps = 96000, 24
        pcm16 = synth.config4_stereo24(args.seconds, 1 + rank, sr)          # int32 container, 24-bit values
    elif args.workload == 'batch':
        # configs[4]: the rank's share of the 1024-stream batch, concatenated in HBM, every stream its own frame numbering
        per = [synth.config5_stream(rank * args.streams + i, args.seconds, sr) for i in range(args.streams)]
        lengths = [len(x) for x in per]
        pcm16 = np.concatenate(per)
    else:
        # each rank encodes its own stream (config 5 generator family); rank 0 at N=1 is config 2's stream
        pcm16 = synth.config2_stereo16(args.seconds, 0, sr) if world == 1 else synth.config5_stream(rank, args.seconds, sr)
    pcm = torch.from_numpy(pcm16.astype(np.int32)).to(dev)     # int32 at the C ABI, like pyflac/encoder.py:112
    nsamp = pcm.shape[0]
    ctx = batch.Context(local)
    s = batch.settings(args.level, ch, bps, sr, bs, True)

    # the only shared datum: the 86-byte stream header, broadcast from rank 0 (RCCL over xGMI)
    hdr = torch.zeros(86, dtype=torch.uint8, device=dev)
    if rank == 0:
        from pyflac_amd.encoder import stream_header_bytes
        hdr.copy_(torch.frombuffer(bytearray(stream_header_bytes(s)), dtype=torch.uint8))
    if world > 1:
        dist.broadcast(hdr, 0)

    out = offs = dec = None

    def step():
        nonlocal out, offs, dec
        out, offs, est = ctx.encode(s, pcm, stream_lengths=lengths, out=out, offsets=offs)
        # decode from the bytes alone: the frame index is rebuilt on the GPU inside the timed region (the number of frames is
        # what STREAMINFO tells a decoder: total samples / block size)
        dec, status, dst = ctx.decode_stream(out[:est.total_bytes], ch, bps, nsamp, nframes=est.nblocks, out=dec)
        return est, dst, status

    for _ in range(args.warmup):
        est, dst, status = step()
    # bit-exactness gate (outside the timed region): the round trip equals the input
    assert int(status[:, 0].max()) == 0, 'decoder reported frame errors'
    assert torch.equal(dec[:nsamp], pcm), 'round trip is not bit-exact'
    # (the comparison of the encoded frames with the CPU oracle's is part of the cpu_baseline leg below: the oracle is only
    # touched there)
    h_chk = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == 'stream16':
        nchk = min(nsamp, 40 * bs) // bs * bs
        h_offs = offs.cpu().numpy()
        h_chk = (nchk, out[:int(h_offs[nchk // bs])].cpu().numpy().tobytes())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc_ms = dec_ms = 0.0
    total_bytes = 0
    for _ in range(args.steps):
        est, dst, status = step()
        enc_ms += est.encode_kernel_ms
        dec_ms += dst.decode_kernel_ms
        total_bytes = est.total_bytes
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    chsamples = nsamp * ch                                   # per rank per step
    value = chsamples * world / (ms_per_step * 1e-3) / 1e6
    enc_k = enc_ms / args.steps
    dec_k = dec_ms / args.steps
    alg_bytes = chsamples * 4 + total_bytes                   # read int32 PCM once + write the frames once
    achieved = alg_bytes / (enc_k * 1e-3) / 1e9
    if rank == 0:
        res = {
            'metric': 'Msamples/s encode (level %d, %dkHz/%d-bit stereo, blk 4096) + decode; bit-exact' % (args.level, sr // 1000, bps),
            'value': round(value, 1), 'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'int32', 'data': 'synthetic',
            'config': {'workload': '%s: %s encode then decode of its output, stereo %d-bit %d kHz, '
                                   'blocksize 4096, level %d, %.0f s (%d blocks) per GPU, int32 PCM resident in HBM, '
                                   'MD5 off (FLAC__stream_encoder_set_do_md5(0)); decoder uses the frame index the '
                                   'encoder produced (device-resident)' %
                                   ({'stream16': 'configs[1]+[2]', 'stream24': 'configs[3]', 'batch': 'configs[4]'}[args.workload],
                                    'single-stream' if lengths is None else '%d independent streams in one launch,' % len(lengths),
                                    bps, sr // 1000, args.level, args.seconds, est.nblocks),
                       'blocks_per_gpu': int(est.nblocks), 'compression_ratio': round(total_bytes / (chsamples * (bps // 8)), 4)},
            'encode_kernel_msamples_per_s': round(chsamples / (enc_k * 1e-3) / 1e6, 1),
            'decode_kernel_msamples_per_s': round(chsamples / (dec_k * 1e-3) / 1e6, 1),
            'encode_kernel_ms': round(enc_k, 3), 'decode_kernel_ms': round(dec_k, 3),
            'roofline': {'bound': 'hbm', 'kernel': 'fg_encode_fast_kernel', 'achieved': round(achieved, 2), 'peak': 8000.0,
                         'unit': 'GB/s', 'frac': round(achieved / 8000.0, 5), 'traffic': pmc_traffic(args, est),
                         'algorithmic_bytes_per_launch': int(alg_bytes)},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == 'stream16':
            res['cpu_baseline'] = cpu_baseline(synth.config2_stereo16(60.0, 0, 48000), args.level, 48000)
            # checker use of the same oracle: the first 40 frames the GPU wrote are the oracle's, byte for byte
            from oracle import oracle as O
            cfg, _ = O.config(args.level, ch, bps, sr, bs, True)
            ref, _sizes = O.encode_stream(cfg, pcm16[:h_chk[0]].astype(np.int32))
            assert h_chk[1] == ref[86:86 + len(h_chk[1])], 'encoded frames differ from the oracle'
            res['cpu_baseline']['checked'] = '%d GPU frames byte-identical to the oracle' % (h_chk[0] // bs)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
