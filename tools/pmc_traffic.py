#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes over tools/gemm_bench.py (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into profiles/r01_gemm_pmc_traffic.json: HBM bytes per launch of each persistent
GEMM class, bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE reads 1/2 on gfx950, WRITE_SIZE exact).
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv
import glob
import json
import sys
from collections import defaultdict

NAMES = {"1": ("qkv_bias_f16", 2304, 768), "3": ("fc_bias_quickgelu", 3072, 768),
         "10": ("split_qkv_bias_f32", 2304, 768), "12": ("split_fc_bias_quickgelu", 3072, 768)}


def collect(d, counter):
    per = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            per[row["Kernel_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return per


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/gemm_bench.py "
                   "(M=65536, random data, launch order qkv/out/fc1/fc2 per round); bytes = (2*FETCH_SIZE + WRITE_SIZE)"
                   "*1024 per MI355X_MICROARCH.md (FETCH_SIZE reads 1/2 on gfx950, WRITE_SIZE exact); medians over "
                   "the launches of a class", "classes": {}}

    def med(v):
        v = sorted(v)
        return v[len(v) // 2]

    for kname in fetch:
        if "gemm_f16_big_kernel" not in kname:
            continue
        epi = kname.split("<")[1].split(",")[0].strip()
        fv = [v for _, v in sorted(fetch[kname])]
        wv = [v for _, v in sorted(write.get(kname, []))]
        if epi in ("2", "11"):   # two shapes share the residual epilogue: out-proj (K=768) and FC2 (K=3072) alternate
            # gemm_bench launches `reps` of one shape back to back; split by value clusters (FC2 fetches ~4x more)
            cut = (min(fv) + max(fv)) / 2
            nm2 = "bias_residual" if epi == "2" else "split_bias_residual"
            groups = {f"{nm2}:768:768": ([v for v in fv if v < cut], None),
                      f"{nm2}:768:3072": ([v for v in fv if v >= cut], None)}
            wmed = med(wv) if wv else None
            for key, (vals, _) in groups.items():
                if vals:
                    f_kb = med(vals)
                    out["classes"][key] = {"kernel": f"gemm_f16_big_kernel<{epi},0>", "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": wmed,
                                           "hbm_bytes_per_launch": None if wmed is None else int((2 * f_kb + wmed) * 1024)}
        elif epi in NAMES:
            nm, n, k = NAMES[epi]
            f_kb, w_kb = med(fv), (med(wv) if wv else None)
            out["classes"][f"{nm}:{n}:{k}"] = {"kernel": f"gemm_f16_big_kernel<{epi},0>", "FETCH_SIZE_KB": f_kb,
                                               "WRITE_SIZE_KB": w_kb,
                                               "hbm_bytes_per_launch": None if w_kb is None else int((2 * f_kb + w_kb) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["classes"], indent=1))


if __name__ == "__main__":
    main()
