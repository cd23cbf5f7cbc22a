#!/usr/bin/env python
"""pytest with model attributes overridden after construction (see tools/ab_attr.py).   usage: tools/pytest_attr.py name=value ... -- <pytest args>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
cut = args.index('--')
sets = dict(a.split('=', 1) for a in args[:cut])
import strajnet_amd
cls = strajnet_amd.STrajNet
init = cls.__init__


def patched(self, *a, **k):
    init(self, *a, **k)
    for n, v in sets.items():
        setattr(self, n, eval(v))


cls.__init__ = patched
import pytest
sys.exit(pytest.main(args[cut + 1:]))
