"""The N > 1 bench path on real kernels: `bench.py --gpus 2 --share-gpu` starts two ranks as a child
process before anything touches the GPU; both work on cuda:0, the collectives (header broadcast, barriers around the timed loop,
MAX reduction of the time, the gathered bit-exactness verdicts) run over gloo on host tensors.  What an 8-GPU node would run over
RCCL is the same code with `nccl` in place of `gloo` and one GPU per rank.

Checked here: rank 1's data path (its own seeds: streams 1, 3, 5, ... of the batch), `shard.streams_for_rank` on real PCM, the
line's `checked` object (every rank compared the first and the last stream of ITS share with the oracle and the round trip of
all its streams with the input), per-rank step times, and value = all ranks' samples / the slowest rank's time.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith('{')]
    return p, lines


def test_two_ranks_on_one_gpu_batch_workload():
    p, lines = _bench(['--gpus', '2', '--share-gpu', '--streams', '3', '--seconds', '2.5', '--steps', '3', '--warmup', '1', '--no-e2e', '--no-configs'])
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1                               # rank 0 only
    r = lines[0]
    assert r['n_gpus'] == 2 and r['steps'] == 3 and r['scaling'] == 'weak'
    assert 'configs[4]' in r['config']['workload'] and '3 independent streams' in r['config']['workload']
    c = r['checked']
    assert c['ranks'] == 2 and 'first and last stream of every rank' in c['what']
    # two streams (first, last) of 2.5 s = 30 frames each, on each of the two ranks
    assert c['frames'].startswith('120 frames over 2 ranks') and len(c['sha256']) == 2
    assert len(r['ms_per_step_rank']) == 2 and max(r['ms_per_step_rank']) == pytest.approx(r['ms_per_step'], abs=2e-3)
    # value = the samples of BOTH ranks over the slowest rank's time
    per_rank = 3 * int(2.5 * 48000) * 2
    assert r['value'] == pytest.approx(per_rank * 2 / (r['ms_per_step'] * 1e-3) / 1e6, rel=5e-3)      # (ms_per_step is printed to three decimals)


def test_two_ranks_single_stream_workload():
    """--workload stream16 at N > 1: every rank encodes its own stream (generator family of configs[4], seed = rank)."""
    p, lines = _bench(['--gpus', '2', '--share-gpu', '--workload', 'stream16', '--seconds', '6', '--steps', '3', '--warmup', '1',
                       '--no-e2e', '--no-configs'])
    assert p.returncode == 0, p.stderr[-3000:]
    r = lines[0]
    assert r['n_gpus'] == 2 and r['checked']['ranks'] == 2
    assert r['checked']['frames'].startswith('142 frames over 2 ranks')          # 71 blocks of 4096 in 6 s at 48 kHz, on each rank
