import importlib, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
cfg = bench.CONFIGS[2]
s = bench.build_workload(wl, cfg["reps"], solver=cfg["solver"])
r = bench.md_leg(pkg, s, steps=12, device_neigh=True)
r.pop("what"); print(json.dumps(r))
