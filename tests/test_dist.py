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
    q.put((rank, got, mine[:3], len(mine), rng, (mn, mx, cnt)))
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


def test_partitions_cover_everything():
    from pyflac_amd import shard
    for world in (1, 2, 3, 8):
        seen = sorted(s for r in range(world) for s in shard.streams_for_rank(1024, r, world))
        assert seen == list(range(1024))
        edges = [shard.block_range_for_rank(7031, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == 7031
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
