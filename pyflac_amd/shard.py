"""Sharding of independent streams / blocks across the GPUs of one node.

FLAC frames are independently encodable and decodable, so the path shards with no data-path collective:
whole streams go round-robin to ranks (config 5: stream s -> rank s mod world), and one long stream is cut
into contiguous block ranges (frame number = block index, so a rank only needs its offset).  The only shared
datum is the stream header; rank 0 broadcasts it (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def streams_for_rank(nstreams, rank, world):
    """Indices of the streams rank `rank` encodes (round-robin: 128 per GPU for 1024 streams on 8 GPUs)."""
    return list(range(rank, nstreams, world))


def block_range_for_rank(nblocks, rank, world):
    """Contiguous [first, last) block range of one long stream for this rank; sizes differ by at most one."""
    base, extra = divmod(nblocks, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def broadcast_header(header_bytes, device, src=0):
    """Broadcast the fLaC+STREAMINFO+VORBIS_COMMENT header (<= 86 bytes) from rank `src`; returns bytes."""
    n = 86
    buf = torch.zeros(n, dtype=torch.uint8, device=device)
    if dist.get_rank() == src:
        assert len(header_bytes) == n
        buf.copy_(torch.frombuffer(bytearray(header_bytes), dtype=torch.uint8))
    dist.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def gather_frame_sizes(sizes, device):
    """All ranks learn every rank's frame sizes (for min/max frame size in STREAMINFO).  Host-side reduction
    in the single-process API; provided for multi-rank single-stream encodes."""
    world = dist.get_world_size()
    t = torch.tensor([min(sizes) if sizes else 0xFFFFFF, max(sizes) if sizes else 0, len(sizes)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    mn = min(int(o[0]) for o in out if int(o[2]) > 0) if any(int(o[2]) for o in out) else 0
    mx = max(int(o[1]) for o in out)
    return mn, mx, sum(int(o[2]) for o in out)


def assemble_in_stream_order(nstreams, world, per_rank_results):
    """Inverse of `streams_for_rank`: per_rank_results[r][i] is the result of the i-th stream of rank r (stream
    r + i * world); returns the list of results in stream order ("outputs gathered by the host in
    stream order")."""
    out = [None] * nstreams
    for r in range(world):
        mine = streams_for_rank(nstreams, r, world)
        assert len(per_rank_results[r]) == len(mine), (r, len(per_rank_results[r]), len(mine))
        for i, sidx in enumerate(mine):
            out[sidx] = per_rank_results[r][i]
    assert all(o is not None for o in out)
    return out


def run_sharded(streams, workers):
    """`encode_sharded` under the name the decode side uses: the same round-robin mapping and stream-order reassembly, whatever
    the per-device workers do with their share."""
    return encode_sharded(streams, workers)


def encode_sharded(streams, encoders):
    """Encode `streams` (a list of per-stream inputs) on `len(encoders)` devices of this process: stream s goes to device
    s mod world; encoders[r](list_of_streams) -> list of per-stream outputs, called concurrently (one thread per device; the
    library calls release the GIL).  Returns the outputs in stream order.  pyflac_amd.batch.MultiContext passes one GPU
    context per device; the CPU test passes stubs."""
    import threading
    world = len(encoders)
    results = [None] * world
    errors = [None] * world

    def run(r):
        try:
            mine = streams_for_rank(len(streams), r, world)
            results[r] = encoders[r]([streams[i] for i in mine])
        except BaseException as e:     # noqa: BLE001 (re-raised in the caller's thread)
            errors[r] = e

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    return assemble_in_stream_order(len(streams), world, results)
