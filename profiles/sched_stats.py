"""Prints the wavefront scheduler's per-operation execution counts and lane occupancy for the
bench workload (a counting launch).  Usage: python profiles/sched_stats.py [spp]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

rt = load_package()
from rtiow_amd import scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = 1920, 1080
s = rt.Scene()
scenes.cover_scene(s, W, H)
ctx = rt.Context(0)
ctx.upload_scene(s)
ctx.set_target(W, H)
c = ctx.count(0, spp, 50)
st = ctx.sched_stats()
print(c)
for k, (ex, lanes) in st.items():
    print(f"{k:10s} executions={ex:12d} lanes={lanes:14d} avg_lanes={lanes / max(ex, 1):6.2f}")
