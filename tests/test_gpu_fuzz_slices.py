"""A slice of every differential fuzzer under tests/tools/ as part of the GPU suite (the long runs are quoted in DESIGN.md
section 0 / 2; these keep the tools themselves and the paths they cover from rotting).  Every tool prints "<n> bad" at the end."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    p = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    tail = (p.stdout + p.stderr).strip().splitlines()[-12:]
    assert p.returncode == 0, tail
    last = [ln for ln in p.stdout.splitlines() if re.search(r'\b\d+ bad', ln)]
    assert last, tail
    assert re.search(r'\b0 bad', last[-1]), tail


@pytest.mark.parametrize('args', [
    ['tests/tools/gpu_fuzz.py', '30000', '400'],             # encoder vs oracle, decoder with the encoder's index and from the bytes
    ['tests/tools/gpu_fuzz.py', '79100', '100'],             # (holds seed 79157: codes of hundreds of bytes)
    ['tests/tools/gpu_batch_fuzz.py', '0', '120'],           # several ragged streams per launch
    ['tests/tools/gpu_api_fuzz.py', '0', '150'],             # StreamEncoder / StreamDecoder, randomly cut input, verify, limit_min_bitrate
    ['tests/tools/dec_stream_fuzz.py', 'gpu', '0', '60'],    # constructed streams no encoder writes, every GPU decode path
    ['tests/tools/gpu_damage_fuzz.py', '0', '120'],          # random damage on the fixtures through the stream decoder
    ['tests/tools/index_damage_fuzz.py', '0', '500', '--api'],   # random damage on small-block streams through the stream decoder
], ids=lambda a: os.path.basename(a[0])[:-3] + '_' + a[-2])
def test_fuzzer_slice(args):
    _run(args)
