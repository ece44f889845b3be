import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import json
    from tests import cases
    with open(os.path.join(cases.GOLDEN, 'encode_vectors.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def damage_golden():
    """What clients of the reference's libFLAC 1.4.3 see for damaged streams (tests/cases.py DAMAGE_CASES)."""
    import json
    from tests import cases
    with open(os.path.join(cases.GOLDEN, 'damage_vectors.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def handmade_streams():
    """Streams outside any encoder's repertoire with the PCM the reference binary decoded from them
    (oracle/gen_golden_handmade.py): name -> (stream bytes, pcm int32[n, ch])."""
    import numpy as np
    from tests import cases
    z = np.load(os.path.join(cases.GOLDEN, 'handmade_streams.npz'))
    return {k[:-5]: (z[k].tobytes(), z[k[:-5] + '.pcm']) for k in z.files if k.endswith('.flac')}


@pytest.fixture(scope='session')
def fuzz_golden():
    """Reference hashes of the seeded random corpus (tests/fuzzgen.py seeds 0..GOLDEN_SEEDS-1; oracle/gen_golden.py)."""
    import json
    from tests import cases
    with open(os.path.join(cases.GOLDEN, 'fuzz_vectors.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def limit_golden():
    """Reference output with FLAC__stream_encoder_set_limit_min_bitrate(true) (tests/cases.py LIMIT_CASES)."""
    import json
    from tests import cases
    with open(os.path.join(cases.GOLDEN, 'limit_vectors.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def small_streams():
    import numpy as np
    from tests import cases
    z = np.load(os.path.join(cases.GOLDEN, 'small_streams.npz'))
    return {k: z[k].tobytes() for k in z.files}
