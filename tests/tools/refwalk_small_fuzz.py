"""fg_refwalk.h against the reference binary on TINY streams: frames of 10..60 bytes, read sizes of a few bytes to a few hundred, so
that refills of libFLAC's reader fall everywhere around damaged frames (a refill a few bytes behind a damaged frame's sync code
made round 3's walker loop forever).  Every probe runs in a child process under a watchdog: a hang is a finding, not a stuck
run.  Build container only.
usage: python tests/tools/refwalk_small_fuzz.py first count [--dump]"""
import multiprocessing as mp
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def small_stream(seed):
    from tests import cases
    return cases.small_damaged_stream(seed)


def _worker(first, count, q):
    from tests.tools import refwalk_vs_ref as V
    for seed in range(first, first + count):
        data, rs = small_stream(seed)
        q.put(('start', seed))
        try:
            ok = V.compare('small %d' % seed, data, rs, verbose=False)
        except Exception as e:        # noqa: BLE001
            ok = False
            q.put(('exc', seed, repr(e)))
        q.put(('done', seed, ok))
    q.put(('end',))


def run(first, count, timeout=10.0):
    """Returns (differing seeds, hanging seeds)."""
    bad, hung = [], []
    pos = first
    while pos < first + count:
        q = mp.Queue()
        p = mp.Process(target=_worker, args=(pos, first + count - pos, q))
        p.start()
        cur = None
        while True:
            try:
                m = q.get(timeout=timeout)
            except Exception:        # noqa: BLE001  (queue.Empty: the probe does not return)
                hung.append(cur)
                p.kill(); p.join()
                pos = cur + 1
                break
            if m[0] == 'start':
                cur = m[1]
            elif m[0] == 'done':
                if not m[2]:
                    bad.append(m[1])
            elif m[0] == 'end':
                p.join()
                pos = first + count
                break
    return bad, hung


if __name__ == '__main__':
    first, count = int(sys.argv[1]), int(sys.argv[2])
    bad, hung = run(first, count)
    print('small-stream refwalk fuzz %d..%d: %d differ, %d hang' % (first, first + count - 1, len(bad), len(hung)))
    print('  differ:', bad[:40])
    print('  hang:', hung[:40])
    if '--dump' in sys.argv:
        from tests.tools import refwalk_vs_ref as V
        for s in bad[:6]:
            d, rs = small_stream(s)
            print(s, len(d), rs)
            V.compare('small %d' % s, d, rs)
