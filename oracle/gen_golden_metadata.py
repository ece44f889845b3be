"""Generate tests/golden/metadata_vectors.json: the blocks the reference's libFLAC 1.4.3 hands to a metadata callback.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden_metadata
For every stream of tests/cases.py METADATA_STREAMS and every filter set-up of METADATA_SETUPS the decoder of the bundled
binary is driven through FLAC__stream_decoder_set_metadata_respond* / init_stream / process_until_end_of_metadata and the
parsed FLAC__StreamMetadata structures are recorded field by field (tests/abi_decode.py metadata_to_dict).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import libflac_ref as R  # noqa: E402
from tests import abi_decode, cases  # noqa: E402


def main():
    out = {}
    for sname in cases.METADATA_STREAMS:
        data = cases.metadata_input(sname)
        for setup, ops in sorted(cases.METADATA_SETUPS.items()):
            res = abi_decode.read_metadata(R.lib(), data, ops)
            out['%s/%s' % (sname, setup)] = res
            print('%-10s %-16s ok=%d types=%s' % (sname, setup, res['ok'], [b['type'] for b in res['blocks']]))
    with open(os.path.join(cases.GOLDEN, 'metadata_vectors.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True, separators=(',', ':'))


if __name__ == '__main__':
    main()
