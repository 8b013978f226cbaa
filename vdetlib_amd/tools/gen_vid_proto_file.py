#!/usr/bin/env python3
"""CLI: make a .vid protocol file from a directory of frames (reference tools/gen_vid_proto_file.py:10-24).
Idempotent like the reference: exits early when the output exists."""
import argparse
import os
import sys

from ..utils.protocol import vid_proto_from_dir, proto_dump


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('vid_name')
    parser.add_argument('root_dir')
    parser.add_argument('out_file')
    args = parser.parse_args(argv)
    if os.path.isfile(args.out_file):
        print("{} already exists.".format(args.out_file))
        return 0
    vid = vid_proto_from_dir(args.root_dir, args.vid_name)
    save_dir = os.path.dirname(args.out_file)
    if save_dir and not os.path.isdir(save_dir):
        os.makedirs(save_dir)
    proto_dump(vid, args.out_file)
    return 0


if __name__ == '__main__':
    sys.exit(main())
