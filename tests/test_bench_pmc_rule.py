"""bench.py quotes the committed counter passes (roofline.traffic, issue_ceiling) only for the build they were measured on
(VERDICT round 4, item 7): the rule as a function, and the committed passes against the library in the tree."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    import importlib
    return importlib.import_module('bench')


def test_a_pass_of_other_kernels_is_not_quoted_and_the_line_says_why():
    b = _bench()
    store = {'r06_pmc.json': {'workload': 'stream16', 'level': 5, 'blocks': 7032, 'build_id': 'aaaaaaaaaaaaaaaa', 'kernel_id': 'cccccccccccccccc',
                              'host_id': 'dddddddddddddddd', 'encode_traffic_bytes_per_launch': 1, 'decode_traffic_bytes_per_launch': 2}}
    pmc, name, same, note = b.committed_pmc('stream16', 5, 7032, 'cccccccccccccccc', load=store.get, build_id='aaaaaaaaaaaaaaaa')
    assert same and note is None and name == 'r06_pmc.json' and pmc['decode_traffic_bytes_per_launch'] == 2
    # (round 6) an edit of a host file: another build id, the same kernels -- the pass stands
    pmc, name, same, note = b.committed_pmc('stream16', 5, 7032, 'cccccccccccccccc', load=store.get, build_id='eeeeeeeeeeeeeeee')
    assert same and note is None
    # other kernels: not quoted, and the note names both
    pmc, name, same, note = b.committed_pmc('stream16', 5, 7032, 'bbbbbbbbbbbbbbbb', load=store.get, build_id='aaaaaaaaaaaaaaaa')
    assert not same and 'cccccccccccccccc' in note and 'bbbbbbbbbbbbbbbb' in note and 'not quoted' in note
    # another shape of the same workload: not quoted either
    pmc, name, same, note = b.committed_pmc('stream16', 8, 7032, 'cccccccccccccccc', load=store.get)
    assert not same and note
    # a round-5 pass names a whole build and counts for exactly that build
    r5 = {'r05_pmc.json': {'workload': 'stream16', 'level': 5, 'blocks': 7032, 'build_id': 'aaaaaaaaaaaaaaaa'}}
    assert b.committed_pmc('stream16', 5, 7032, 'x', load=r5.get, build_id='aaaaaaaaaaaaaaaa')[2]
    pmc, name, same, note = b.committed_pmc('stream16', 5, 7032, 'x', load=r5.get, build_id='bbbbbbbbbbbbbbbb')
    assert not same and 'build aaaaaaaaaaaaaaaa' in note
    assert not b.committed_pmc('stream16', 5, 7032, 'x', load=r5.get)[2]
    # a pass from before the build ids existed stands for no build
    old = {'r04_pmc.json': {'workload': 'stream16', 'level': 5, 'blocks': 7032}}
    pmc, name, same, note = b.committed_pmc('stream16', 5, 7032, 'cccccccccccccccc', load=old.get, build_id='aaaaaaaaaaaaaaaa')
    assert name == 'r04_pmc.json' and not same and 'before round 5' in note
    # no pass at all: nothing to quote, nothing to say
    pmc, name, same, note = b.committed_pmc('wasted', 5, 7032, 'cccccccccccccccc', load={}.get)
    assert pmc == {} and not same and note is None


def test_the_other_workloads_have_files_of_their_own():
    b = _bench()
    seen = []
    b.committed_pmc('stream24', 8, 7032, 'x', load=lambda n: seen.append(n))
    assert seen[0] == 'r06_pmc_stream24.json' and seen[1] == 'r05_pmc_stream24.json'


def test_the_library_names_its_kernels_and_its_host_files_apart():
    """flacgpu_kernel_id() is the hash the Makefile takes over the .hip files and the headers they include; recomputed here from the
    tree, so that a stale fg_build_id.inc (a library older than its sources) shows."""
    import glob, hashlib
    from pyflac_amd import _lib
    L = _lib.lib()
    ids = [getattr(L, n)().decode() for n in ('flacgpu_build_id', 'flacgpu_kernel_id', 'flacgpu_host_id')]
    assert all(len(i) == 16 and int(i, 16) >= 0 for i in ids) and len(set(ids)) == 3
    src = os.path.join(ROOT, 'pyflac_amd', 'csrc')
    ksrc = sorted([os.path.basename(f) for f in glob.glob(os.path.join(src, '*.hip'))] +
                  ['fg_dev.h', 'fg_types.h', 'fg_dec_hdr.h', 'flac_enc_pipe_impl.h', 'pipe_shape.inc'])
    h = hashlib.sha256()
    for f in ksrc:
        with open(os.path.join(src, f), 'rb') as fh:
            h.update(fh.read())
    assert h.hexdigest()[:16] == ids[1]


def test_the_committed_passes_name_a_build():
    """Every round-5 pass carries the id of the library it ran on (sixteen hex digits); the three of a collection share it."""
    ids = set()
    for name in ('r05_pmc.json', 'r05_pmc_stream24.json', 'r05_pmc_batch.json'):
        with open(os.path.join(ROOT, 'profiles', name)) as fh:
            p = json.load(fh)
        assert len(p['build_id']) == 16 and int(p['build_id'], 16) >= 0
        ids.add(p['build_id'])
    assert len(ids) == 1
