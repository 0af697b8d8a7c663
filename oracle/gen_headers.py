#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- prints oracle/gen/<model>_gen.h, the plant functions the C checker is compiled
with, from the checker's own statement of the plants (oracle/models_numpy.py, the one the golden vectors were
produced with), using the printing functions of tools/gen_models.py.

Usage:  python oracle/gen_headers.py [car acrobot quad_ea quad_quat]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_models  # noqa: E402
from oracle import models_numpy  # noqa: E402

if __name__ == "__main__":
    gen_models.main(sys.argv, plants=models_numpy, out_dir=os.path.join(ROOT, "oracle", "gen"))
