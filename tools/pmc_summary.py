#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/collect_profiles.sh (csv output) to per-kernel medians with the gfx950
corrections of MI355X_MICROARCH.md (HBM section): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE reports
half the bytes of wide streaming reads, WRITE_SIZE is exact; both counters are in KiB.  SQ counters are summed over the
shader engines / XCDs of a dispatch.  Usage: python tools/pmc_summary.py <collect dir> <tag>"""
import csv
import glob
import json
import sys
from collections import defaultdict


def collect(d):
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> sum
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"]][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return per


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out = {"note": __doc__.split("Usage")[0].strip(), "runs": {}}
    for name in ("pmc_rr_fetch", "pmc_rr_write", "pmc_e_fetch", "pmc_e_write", "pmc_e_sq", "pmc_e2_fetch", "pmc_e2_write", "pmc_e2_sq"):
        per = collect(f"{root}/{name}")
        out["runs"][name] = {k[:110]: {c: med(list(v.values())) for c, v in cs.items()} for k, cs in per.items()
                             if not k.startswith("__amd") and "at::" not in k}
    # HBM bytes per launch where both passes exist
    for pre, label in (("pmc_rr", "rerank_N20000"), ("pmc_e", "featgemm_20kx20kx768_fp16"),
                       ("pmc_e2", "featgemm_two_tensors_20kx20kx768_fp16")):
        f, w = out["runs"].get(pre + "_fetch", {}), out["runs"].get(pre + "_write", {})
        out[label + "_hbm_bytes_per_launch"] = {
            k: int((2 * f[k].get("FETCH_SIZE", 0) + w.get(k, {}).get("WRITE_SIZE", 0)) * 1024) for k in f}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
