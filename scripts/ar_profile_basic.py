#!/usr/bin/env python3
"""BaSIC scanline codec, a few encode+decode passes -- meant to run under rocprofv3 --kernel-trace. usage: ar_profile_basic.py <batch> <level>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import scripts.ar_bench as ab
B, lvl = (args + ["64", "0"])[:2]
codec = ab.prep(ab.basic_codec())
codec.set_complex_level(int(lvl))
ab.run(f"BaSIC scanline level {lvl}", codec, int(B), steps=2)
