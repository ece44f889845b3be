"""``python -m pyflac_amd`` -- WAV <-> FLAC file conversion on the GPU.

Same command line as the reference's ``pyflac`` tool (pyflac/__main__.py:19-56): the input is recognised by its magic
(``RIFF`` -> encode with FileEncoder, ``fLaC`` -> decode with FileDecoder), the output defaults to the input name with
the other extension.  Like the reference, ``-v`` is a store-false flag: verification is ON by default and passing ``-v``
turns it OFF (pyflac/__main__.py:30).
"""
import argparse
import pathlib
import sys

from . import FileDecoder, FileEncoder


def parse(argv=None):
    ap = argparse.ArgumentParser(prog='pyflac_amd', description='FLAC encoder/decoder on the MI355X',
                                 epilog='Convert WAV files to FLAC and vice versa')
    ap.add_argument('input_file', type=pathlib.Path, help='Input file to encode/decode')
    ap.add_argument('-o', '--output-file', type=pathlib.Path, help='Output file')
    ap.add_argument('-c', '--compression-level', type=int, choices=range(0, 9), default=5,
                    help='0 is the fastest compression, 5 is the default, 8 is the highest compression')
    ap.add_argument('-b', '--block-size', type=int, default=0, help='The block size')
    ap.add_argument('-v', '--verify', action='store_false', default=True, help='Verify the compressed data')
    return ap.parse_args(argv)


def main(argv=None):
    args = parse(argv)
    with open(args.input_file, 'rb') as fh:
        magic = fh.read(4)
    src = pathlib.Path(args.input_file)
    if magic.upper() == b'RIFF':
        dst = args.output_file if args.output_file is not None else src.with_suffix('.flac')
        FileEncoder(input_file=src, output_file=dst, blocksize=args.block_size, compression_level=args.compression_level,
                    verify=args.verify).process()
    elif magic.upper() == b'FLAC':
        dst = args.output_file if args.output_file is not None else src.with_suffix('.wav')
        FileDecoder(src, dst).process()
    else:
        raise ValueError('Please provide either a WAV or a FLAC file')
    return 0


if __name__ == '__main__':
    sys.exit(main())
