#!/usr/bin/env python3
"""python compat/run_ref.py <reference script> [args...]  -- run a script of the reference (from its root directory) with
compat/ in front of everything on sys.path."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    script = os.path.abspath(sys.argv[1])
    ref_root = os.path.dirname(script)
    sys.path[:] = [HERE, os.path.dirname(HERE), ref_root] + [p for p in sys.path if p not in (HERE, ref_root)]
    import _overlay

    _overlay.install()
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
