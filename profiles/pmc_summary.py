#!/usr/bin/env python3
"""Summarises the three PMC passes of profiles/collect_pmc.sh into profiles/<tag>_pmc_summary.json and
profiles/traffic_latest.json (read by bench.py for roofline.traffic).

Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 64 B per memory-side read request.  Calibration on this repo's own access pattern
(tools/probe_bench under the same counter): random 4-byte loads -> 64 B per load as counted (one 64-B
request each, no correction); float4 stream -> exactly half the bytes (128-B requests, x2 correction).
correct_pass is >98 % random probes, so its FETCH_SIZE is used as counted; streaming kernels get x2."""
import collections
import csv
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def load(path):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def find(d, sub, ctr):
    return [v for (k, c), vs in d.items() if sub in k and c == ctr for v in vs]


probe = load(os.path.join(out_dir, "probe", "p_counter_collection.csv"))
fetch = load(os.path.join(out_dir, "fetch", "f_counter_collection.csv"))
write = load(os.path.join(out_dir, "write", "w_counter_collection.csv"))

cal_rand = max(find(probe, "rand_probe<4>", "FETCH_SIZE")) * 1024 / (8192 * 256 * 64)
cal_stream = max(find(probe, "stream_read", "FETCH_SIZE")) * 1024 / (16 * 2 ** 30)

line = None
for ln in open(os.path.join(out_dir, "fetch.log")):
    if ln.startswith("{"):
        line = json.loads(ln)
bases = line["config"]["bases_per_gpu"] if line else None

STREAMING = ("l1_", "ln_", "final_count", "compact", "threshold", "lens_", "popcount")
kern = {}
for (k, c), vs in fetch.items():
    kern.setdefault(k, {})["fetch_kib"] = vs
for (k, c), vs in write.items():
    kern.setdefault(k, {})["write_kib"] = vs
summary = {}
for k, d in kern.items():
    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    f = d.get("fetch_kib", [0.0])
    w = d.get("write_kib", [0.0])
    corr = 2.0 if any(s in short for s in STREAMING) else 1.0
    summary[short] = {"launches": max(len(f), len(w)), "fetch_bytes_per_launch_counted": sum(f) / len(f) * 1024,
                      "fetch_correction": corr, "write_bytes_per_launch": sum(w) / len(w) * 1024,
                      "hbm_bytes_per_launch": sum(f) / len(f) * 1024 * corr + sum(w) / len(w) * 1024}
res = {"tag": tag, "calibration": {"random_4B_load_bytes_as_counted": cal_rand,
                                   "stream_fraction_counted": cal_stream}, "bases_per_launch": bases, "kernels": summary}
with open(os.path.join(here, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(res, f, indent=1)
# correct_pass = the scan passes of the step: the reverse pass is one launch (one_kernel<64, K> / correct_kernel<64, M>), the
# forward pass in lane form is several (lane_kernel / lane_walk_kernel and its helpers: unit tables + packed copy, sync
# points, link, the replay of the fixes, the few reads handed back to the group kernel); bytes of all of them / passes
PASS_MAIN = ("correct_kernel<", "one_kernel<", "lane_kernel<", "lane_walk_kernel<")
PASS_AUX = ("lane_units_kernel", "lane_pack_kernel", "lane_sync_kernel", "lane_link_kernel", "lane_apply_kernel", "lane_apply_walk_kernel",
            "succ_build_kernel")
ck = [v for k, v in summary.items() if k.startswith(PASS_MAIN)]
aux = [v for k, v in summary.items() if k.startswith(PASS_AUX)]
if ck and bases:
    # passes = launches of the main kernels, not counting the list-mode re-runs (a handful of reads: <..., true> instances)
    n_launch = sum(v["launches"] for k, v in summary.items() if k.startswith(PASS_MAIN) and ", true>" not in k)
    per_launch = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ck + aux) / n_launch
    with open(os.path.join(here, "traffic_latest.json"), "w") as f:
        json.dump({"kernel": "correct_pass", "bases_per_launch": bases, "hbm_bytes_per_launch": per_launch,
                   "launches_averaged": n_launch,
                   "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}, f)
print(json.dumps(res["calibration"]), ck)
