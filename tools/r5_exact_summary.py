#!/usr/bin/env python3
"""gpurun_out/<tag>_x10k_prof, <tag>_x10k_pmc*, <tag>_c0_prof (tools/r5_a.sh) ->
  profiles/r05_exact_10792_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the exact_10792_replica leg (tools/r4_x10k.py)
  profiles/r05_config0_kernel_stats.csv       the same for config 0 (tools/r4_c0.py)
  profiles/r05_exact_10792_traffic_pmc.txt    FETCH_SIZE / WRITE_SIZE / TCC counters of k_gs_blk<256>, per launch
  profiles/exact_traffic_pmc.json             what bench.py puts into the exact-mode sub-objects (roofline-shaped fields)
VERDICT r4 item 3: the exact-mode claim (0.65 of the HBM roof at 10,792 atoms) backed by the evidence the list mode has."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r5a"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def stats(prof):
    f = glob.glob(os.path.join(G, prof, "*", "*kernel_stats.csv"))[0]
    return f, list(csv.DictReader(open(f)))


f10, rows10 = stats(f"{tag}_x10k_prof")
shutil.copy(f10, os.path.join(P, "r05_exact_10792_kernel_stats.csv"))
f0, rows0 = stats(f"{tag}_c0_prof")
shutil.copy(f0, os.path.join(P, "r05_config0_kernel_stats.csv"))
blk10 = next(r for r in rows10 if "k_gs_blk<256>" in r["Name"])
blk0 = next(r for r in rows0 if "k_gs_blk<256>" in r["Name"])
acc = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(G, f"{tag}_x10k_pmc*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "k_gs_blk<256>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
n, B = 10792, 256
npol = 7832   # polarizable atoms of the 2 x 2 x 2 replica (979 per MOF5+H2 cell): the push skips the rows of the others (alpha = 0:
              # "its field is never read", k_gs_blk), so their tensor rows are never loaded
npol0 = 979
npitch = (n + 63) // 64 * 64
nb = (n + B - 1) // B
R = 3 * B
# one launch = one block of 256 atoms: the block's 256 tensor COLUMNS of every row (the push: n rows x 6 components x 256
# doubles), its G (lower triangle: half of R^2) and N (R^2), the small vectors
alg_launch = 48.0 * npol * B + 1.5 * R * R * 8.0 + 24.0 * (n + 3 * B)
us = float(blk10["AverageNs"]) / 1e3
fetch = mean.get("FETCH_SIZE", 0.0) * 1024.0   # rocprofv3 reports KB
write = mean.get("WRITE_SIZE", 0.0) * 1024.0
# the stream is 8-byte-per-lane coalesced loads (one double per lane, 512 B per wave instruction): a full 128-byte line per
# request like the 16-byte-per-lane stream the guide calibrates, so FETCH_SIZE (64 B per request) is doubled as the guide says
traffic = 2.0 * fetch + write
out = {
    "natoms": n, "kernel": "k_gs_blk<256> (exact mode: d = G cb - N d' of one 256-atom block + the push of the previous block's change)",
    "launches_per_iteration": nb, "us_per_launch": us, "launches_profiled": int(blk10["Calls"]), "share_of_gpu_time": float(blk10["Percentage"]) / 100.0,
    "algorithmic_bytes_per_launch": alg_launch, "achieved_gb_s": alg_launch / (us * 1e-6) / 1e9, "peak_gb_s": 8000.0,
    "frac": alg_launch / (us * 1e-6) / 8.0e12,
    "traffic_bytes_per_launch": traffic, "fetch_size_kb_raw": mean.get("FETCH_SIZE"), "write_size_kb": mean.get("WRITE_SIZE"),
    "tcc_hit_rate": (mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])) if "TCC_HIT_sum" in mean else None,
    "counters_mean": mean, "launches_sampled": len(acc.get("FETCH_SIZE", [])),
    "config0_1349": {"us_per_launch": float(blk0["AverageNs"]) / 1e3, "launches_profiled": int(blk0["Calls"]), "share_of_gpu_time": float(blk0["Percentage"]) / 100.0,
                     "algorithmic_bytes_per_launch": 48.0 * npol0 * B + 1.5 * R * R * 8.0 + 24.0 * (1349 + 3 * B)},
    "source": "profiles/r05_exact_10792_kernel_stats.csv, profiles/r05_exact_10792_traffic_pmc.txt (tools/r5_a.sh: rocprofv3 --kernel-trace --stats, then separate --pmc passes, of tools/r4_x10k.py)",
}
out["config0_1349"]["frac"] = out["config0_1349"]["algorithmic_bytes_per_launch"] / (out["config0_1349"]["us_per_launch"] * 1e-6) / 8.0e12
json.dump(out, open(os.path.join(P, "exact_traffic_pmc.json"), "w"), indent=1)
with open(os.path.join(P, "r05_exact_10792_traffic_pmc.txt"), "w") as fh:
    fh.write(f"# k_gs_blk<256>, MOF5+H2 replicate 2 2 2 = {n} atoms, EXACT mode, {nb} launches per iteration (tools/r5_a.sh, tools/r4_x10k.py); means per launch\n")
    for k, v in sorted(mean.items()):
        fh.write(f"{k:28s} n={len(acc[k]):5d} mean={v:16.1f}\n")
    fh.write(f"# kernel stats (profiles/r05_exact_10792_kernel_stats.csv): {blk10['Calls']} launches, {us:.1f} us average, {blk10['Percentage']} % of GPU time\n")
    fh.write(f"# algorithmic bytes per launch: 48 B x {npol} POLARIZABLE rows x {B} columns (rows with alpha = 0 are skipped) + G (lower triangle) + N of the block + vectors = {alg_launch / 1e6:.1f} MB"
             f" -> {alg_launch / (us * 1e-6) / 1e12:.2f} TB/s = {out['frac']:.3f} of the 8 TB/s roof\n")
    fh.write(f"# traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB = {traffic / 1e6:.1f} MB per launch ({traffic / alg_launch:.2f} x the algorithmic bytes)\n")
    if out["tcc_hit_rate"] is not None:
        fh.write(f"# L2 hit rate {out['tcc_hit_rate']:.3f} (a stream: every line is used once)\n")
    fh.write(f"# config 0 (1,349 atoms, profiles/r05_config0_kernel_stats.csv): {blk0['Calls']} launches, {float(blk0['AverageNs']) / 1e3:.1f} us average "
             f"({out['config0_1349']['algorithmic_bytes_per_launch'] / 1e6:.1f} MB per launch = {out['config0_1349']['frac']:.3f} of the roof: a launch there is mostly its ~4.4 us floor)\n")
print(json.dumps(out, indent=1))
