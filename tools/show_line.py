#!/usr/bin/env python3
"""Print the interesting fields of a bench.py JSON line found in a log file."""
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d = json.loads(ln); c = d["config"]
        print("natoms %d ms/step %.3f value %.0f solve %.3f per-iter %.4f sweeps %d colors %s lj %.2f force %.2f list %.2f static %.2f frac %.3f eng_pol %.9f" % (
            c["natoms"], d["ms_per_step"], d["value"], c["ms_solve"], c["ms_per_dipole_iteration"], c["sweeps"], c["colors"], c["ms_ljcoul"],
            c["ms_force"], c["ms_list"], c["ms_static"], d["roofline"]["frac"], c["eng_pol"]))
