#!/usr/bin/env python3
"""Reduce ONE rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_WAIT_ANY; no trace options beside it) over tools/gemm_bench.py into profiles/<tag>_gemm_pmc_mfma.json: matrix-pipe
utilisation of every persistent GEMM class.

  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
      SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (check: 16 cycles per v_mfma_f32_16x16x32_f16 times the
      number of such instructions the launch executes = the counter, to 0.1 %); GRBM_GUI_ACTIVE is reported summed over
      the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
  clock_ghz      = GRBM_GUI_ACTIVE / 8 / kernel duration (End - Start timestamps of the dispatch)
  tflops_at_counter_clock = executed FLOPs / duration under the profiler (profiled passes run 2-5 % slower than plain ones)
Usage: python tools/pmc_mfma.py <pmc dir> <out.json> [M]"""
import csv
import glob
import json
import sys
from collections import defaultdict

NAMES = {"1": ("qkv_bias_f16", 2304, 768, 1), "3": ("fc_bias_quickgelu", 3072, 768, 1),
         "10": ("split_qkv_bias_f32", 2304, 768, 3), "12": ("split_fc_bias_quickgelu", 3072, 768, 3)}
RES = {"2": ("bias_residual", 1), "11": ("split_bias_residual", 3)}


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    d, outp = sys.argv[1], sys.argv[2]
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    per = defaultdict(lambda: defaultdict(dict))   # kernel -> dispatch -> {counter: value, "dur": ns}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "gemm_f16_big_kernel" not in k:
                continue
            e = per[k][row["Dispatch_Id"]]
            e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            e["dur"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    out = {"note": __doc__.split("Usage")[0].strip(), "M": M, "classes": {}}

    def entry(kernel, disp, n, k, mult):
        busy = med([x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for x in disp])
        act = med([x.get("GRBM_GUI_ACTIVE", 0.0) for x in disp]) / 8.0
        dur = med([x["dur"] for x in disp])
        flop = 2.0 * M * n * k * mult
        e = {"kernel": kernel, "launches": len(disp), "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_per_xcd": act,
             "duration_us": round(dur / 1e3, 1), "mfma_busy_frac": round(busy / (1024.0 * act), 4) if act else None,
             "clock_ghz": round(act / dur, 3) if dur else None,
             "expected_busy_cycles_16_per_mfma": flop / 16384.0 * 16.0,
             "executed_tflops_under_profiler": round(flop / dur / 1e3, 1) if dur else None}
        for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            vals = [x[c] for x in disp if c in x]
            if vals:
                e[c] = med(vals)
        if "SQ_WAVE_CYCLES" in e:
            for c, nm in (("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_WAIT_ANY", "parked_frac"), ("SQ_ACTIVE_INST_ANY", "issuing_frac")):
                if c in e:
                    e[nm] = round(e[c] / e["SQ_WAVE_CYCLES"], 4)
        return e

    for kname, disps in per.items():
        epi = kname.split("<")[1].split(",")[0].strip()
        disp = list(disps.values())
        if epi in NAMES:
            nm, n, k, mult = NAMES[epi]
            out["classes"][f"{nm}:{n}:{k}"] = entry(kname[:60], disp, n, k, mult)
        elif epi in RES:   # out-proj (K = 768) and FC2 (K = 3072) share the residual epilogue: split by busy cycles (4x apart)
            nm, mult = RES[epi]
            b = [x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for x in disp]
            cut = (min(b) + max(b)) / 2
            lo = [x for x in disp if x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) < cut]
            hi = [x for x in disp if x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) >= cut]
            if lo:
                out["classes"][f"{nm}:768:768"] = entry(kname[:60], lo, 768, 768, mult)
            if hi:
                out["classes"][f"{nm}:768:3072"] = entry(kname[:60], hi, 768, 3072, mult)
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps({k: {a: v[a] for a in ("mfma_busy_frac", "clock_ghz", "duration_us")} for k, v in out["classes"].items()}, indent=1))


if __name__ == "__main__":
    main()
