#!/bin/bash
# round 3 lab: the MD-shaped legs of bench.py alone (uploaded list / device list / ballistic sorbates)
timeout -k 10 900 python - > gpurun_out/r3_mdleg.log 2>&1 <<'PY'
import importlib, json, sys
sys.path.insert(0, ".")
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
cfg = bench.CONFIGS[2]
s = bench.build_workload(wl, cfg["reps"], solver=cfg["solver"])
for name, kw in (("md_leg_device_neigh", dict(device_neigh=True)),
                 ("md_leg_ballistic", dict(steps=200, device_neigh=True, motion="ballistic", temperature=300.0)),
                 ("md_leg_ballistic_77K", dict(steps=200, device_neigh=True, motion="ballistic", temperature=77.0))):
    r = bench.md_leg(pkg, s, **kw)
    r.pop("what")
    print(name, json.dumps(r), flush=True)
PY
grep -v amdgpu.ids gpurun_out/r3_mdleg.log | tail -8
