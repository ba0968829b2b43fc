#!/usr/bin/env python3
"""Tuning aid (needs a -DT1D_AB_FLAGS=1 build in T1D_LIB_PATH): bench.py's headline launch with parts of the minute
switched off through the timing-only batch flags: 0x100 no risk index, 0x800 no integration (results are meaningless).
usage: ab_flags.py HEXFLAGS [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from simglucose_amd import batch_env  # noqa: E402

flags = int(sys.argv[1], 16)
_init = batch_env.BatchedT1DSimEnv.__init__


def init(self, *a, **k):
    _init(self, *a, **k)
    self._flags0 |= flags
    self._b.flags = self._flags0


batch_env.BatchedT1DSimEnv.__init__ = init
bench.main(["--no-cpu-baseline", "--no-accuracy", "--steps", "600", "--warmup", "200"] + sys.argv[2:])
