"""Imported by Python at start-up when compat/ is on PYTHONPATH: installs the import hook of _overlay.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
try:
    import _overlay

    _overlay.install()
finally:
    sys.path.pop(0)
