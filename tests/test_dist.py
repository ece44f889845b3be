"""world_size-2 gloo tests of the multi-GPU sharding helpers (pyflac_amd/shard.py): streams and block ranges
partition without overlap, and the stream header broadcast delivers rank 0's bytes."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from pyflac_amd import shard
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    hdr = bytes(range(86)) if rank == 0 else b''
    got = shard.broadcast_header(hdr, torch.device('cpu'))
    mine = shard.streams_for_rank(1024, rank, world)
    rng = shard.block_range_for_rank(7032, rank, world)
    mn, mx, cnt = shard.gather_frame_sizes([100 + rank, 200 + rank], torch.device('cpu'))
    # the config-5 mapping driven end to end with a stub encoder: every rank "encodes" its round-robin share, the shares are
    # gathered (gloo) and put back into stream order exactly as bench.py / MultiContext do
    nstreams = 11
    stub = [('rank%d:stream%d' % (rank, sidx)) for sidx in shard.streams_for_rank(nstreams, rank, world)]
    gathered = [None] * world
    dist.all_gather_object(gathered, stub)
    ordered = shard.assemble_in_stream_order(nstreams, world, gathered)
    # the decode side of the same mapping: every rank "decodes" the frames of ITS streams (stub: the stream's name reversed), the
    # PCM is gathered and lands in stream order -- what MultiContext.decode_streams / bench.py --workload batch do per device
    frames = ['frames-of-stream%d' % sidx for sidx in range(nstreams)]
    dec = [frames[sidx][::-1] for sidx in shard.streams_for_rank(nstreams, rank, world)]
    gathered2 = [None] * world
    dist.all_gather_object(gathered2, dec)
    ordered2 = shard.assemble_in_stream_order(nstreams, world, gathered2)
    q.put((rank, got, mine[:3], len(mine), rng, (mn, mx, cnt), ordered, ordered2))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_sharding_and_header_broadcast():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == bytes(range(86)) and res[1][1] == bytes(range(86))
    assert res[0][2] == [0, 2, 4] and res[1][2] == [1, 3, 5] and res[0][3] == res[1][3] == 512
    assert res[0][4] == (0, 3516) and res[1][4] == (3516, 7032)
    assert res[0][5] == res[1][5] == (100, 201, 4)
    want = ['rank%d:stream%d' % (sidx % 2, sidx) for sidx in range(11)]
    assert res[0][6] == want and res[1][6] == want
    want2 = [('frames-of-stream%d' % sidx)[::-1] for sidx in range(11)]
    assert res[0][7] == want2 and res[1][7] == want2


def test_partitions_cover_everything():
    from pyflac_amd import shard
    for world in (1, 2, 3, 8):
        seen = sorted(s for r in range(world) for s in shard.streams_for_rank(1024, r, world))
        assert seen == list(range(1024))
        edges = [shard.block_range_for_rank(7031, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == 7031
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def test_encode_sharded_returns_stream_order():
    """pyflac_amd.shard.encode_sharded (the engine of batch.MultiContext) with stub per-device encoders: stream s runs on
    device s mod world, the devices run concurrently, results come back in stream order; an error on one device surfaces."""
    import threading
    import pytest
    from pyflac_amd import shard
    seen = {}
    together = threading.Barrier(4)

    def make(r, meet=False):
        def run(mine):
            if meet:
                together.wait(timeout=30)          # only passes when the four devices run at the same time
            seen[r] = list(mine)
            return [('dev%d' % r, x) for x in mine]
        return run
    streams = ['s%d' % i for i in range(13)]
    out = shard.encode_sharded(streams, [make(r, True) for r in range(4)])
    assert out == [('dev%d' % (i % 4), 's%d' % i) for i in range(13)]
    assert seen[1] == ['s1', 's5', 's9']
    assert shard.encode_sharded(['a'], [make(0), make(1)]) == [('dev0', 'a')]

    def bad(_mine):
        raise RuntimeError('device lost')
    with pytest.raises(RuntimeError, match='device lost'):
        shard.encode_sharded(streams, [make(0), bad])
