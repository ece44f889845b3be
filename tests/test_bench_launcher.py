"""bench.py --gpus N without a launcher around it: it starts the N ranks itself as a child process
and relays rank 0's line; it never prints an n_gpus: 1 line for N > 1.  CPU only: --dry-run takes the GPU work out (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_itself():
    p = _run(['--gpus', '2', '--dry-run', '--steps', '2', '--streams', '5'])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith('{')]
    assert len(lines) == 1                       # rank 0 only
    assert lines[0]['n_gpus'] == 2 and lines[0]['dry_run'] is True and lines[0]['steps'] == 2
    assert '5 streams per rank, 2 ranks' in lines[0]['config']['workload']


def test_gpus_2_on_a_box_without_two_gpus_is_an_error_not_a_one_gpu_line():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('this box has two GPUs')
    p = _run(['--gpus', '2', '--steps', '1'])
    assert p.returncode != 0
    assert not any(x.startswith('{') for x in p.stdout.splitlines())
    assert '--gpus 2 asked' in p.stderr


def test_world_size_must_match_gpus():
    """Under a launcher (WORLD_SIZE set) the rank count must be what --gpus says."""
    p = _run(['--gpus', '2', '--dry-run'], {'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'})
    assert p.returncode != 0 and 'the launcher started 1 rank' in (p.stderr + p.stdout)
