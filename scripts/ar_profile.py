#!/usr/bin/env python3
"""One topo-group AR codec configuration, a few encode+decode passes -- meant to run under rocprofv3 --kernel-trace --stats.
usage: ar_profile.py <method> <channel_groups> <batch>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, args = sys.argv[:1], sys.argv[1:]
import scripts.ar_bench as ab
method, G, B = (args + ["checkerboard", "1", "256"])[:3]
codec = ab.prep(ab.topogroup_ar_codec(method, channel_groups=int(G)))
ab.run(f"topogroup {method}", codec, int(B), steps=3)
