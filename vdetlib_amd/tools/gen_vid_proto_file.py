#!/usr/bin/env python3
"""``.vid`` protocol generator: a directory of frame images -> one video-protocol JSON file.

Contract (reference tools/gen_vid_proto_file.py:10-24): positional arguments
``vid_name root_dir out_file``; when ``out_file`` is already there the tool says so and exits with status 0
without touching it (the reference pipeline's resume rule: a finished step is never redone); otherwise the
frames of ``root_dir`` become a vid_proto (utils/protocol.py:243-257: natural sort, 1-based frame ids) that
is written with ``proto_dump`` into a freshly created parent directory if need be.
"""
import argparse
import os
import sys

from ..utils.protocol import proto_dump, vid_proto_from_dir


def generate(vid_name, root_dir, out_file, log=None):
    """Returns True when the file was written, False when an existing file made the call a no-op."""
    log = sys.stdout if log is None else log        # (looked up per call: stdout may have been redirected since import)
    if os.path.isfile(out_file):
        log.write("{} already exists.\n".format(out_file))
        return False
    vid = vid_proto_from_dir(root_dir, vid_name)
    parent = os.path.dirname(out_file)
    if parent:
        os.makedirs(parent, exist_ok=True)
    proto_dump(vid, out_file)
    return True


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for name in ("vid_name", "root_dir", "out_file"):
        ap.add_argument(name)
    ns = ap.parse_args(argv)
    generate(ns.vid_name, ns.root_dir, ns.out_file)
    return 0


if __name__ == "__main__":
    sys.exit(main())
