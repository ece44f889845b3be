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


def test_a_pass_of_another_build_is_not_quoted_and_the_line_says_why():
    b = _bench()
    store = {'r05_pmc.json': {'workload': 'stream16', 'level': 5, 'blocks': 7032, 'build_id': 'aaaaaaaaaaaaaaaa',
                              'encode_traffic_bytes_per_launch': 1, 'decode_traffic_bytes_per_launch': 2}}
    pmc, name, same, note = b.committed_pmc('stream16', 5, 7032, 'aaaaaaaaaaaaaaaa', load=store.get)
    assert same and note is None and name == 'r05_pmc.json' and pmc['decode_traffic_bytes_per_launch'] == 2
    pmc, name, same, note = b.committed_pmc('stream16', 5, 7032, 'bbbbbbbbbbbbbbbb', load=store.get)
    assert not same and 'aaaaaaaaaaaaaaaa' in note and 'bbbbbbbbbbbbbbbb' in note and 'not quoted' in note
    # another shape of the same workload: not quoted either
    pmc, name, same, note = b.committed_pmc('stream16', 8, 7032, 'aaaaaaaaaaaaaaaa', load=store.get)
    assert not same and note
    # a pass from before the build ids existed stands for no build
    old = {'r04_pmc.json': {'workload': 'stream16', 'level': 5, 'blocks': 7032}}
    pmc, name, same, note = b.committed_pmc('stream16', 5, 7032, 'aaaaaaaaaaaaaaaa', load=old.get)
    assert name == 'r04_pmc.json' and not same and 'before round 5' in note
    # no pass at all: nothing to quote, nothing to say
    pmc, name, same, note = b.committed_pmc('wasted', 5, 7032, 'aaaaaaaaaaaaaaaa', load={}.get)
    assert pmc == {} and not same and note is None


def test_the_other_workloads_have_files_of_their_own():
    b = _bench()
    seen = []
    b.committed_pmc('stream24', 8, 7032, 'x', load=lambda n: seen.append(n))
    assert seen[0] == 'r05_pmc_stream24.json'


def test_the_committed_passes_name_a_build():
    """Every round-5 pass carries the id of the library it ran on (sixteen hex digits); the three of a collection share it."""
    ids = set()
    for name in ('r05_pmc.json', 'r05_pmc_stream24.json', 'r05_pmc_batch.json'):
        with open(os.path.join(ROOT, 'profiles', name)) as fh:
            p = json.load(fh)
        assert len(p['build_id']) == 16 and int(p['build_id'], 16) >= 0
        ids.add(p['build_id'])
    assert len(ids) == 1
