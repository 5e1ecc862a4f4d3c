#!/usr/bin/env python3
"""round 4: the 10,792-atom exact-mode replica alone: bench sub-object exact_10792_replica"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
print(json.dumps(bench.exact_replica(torch, pkg, wl)))
