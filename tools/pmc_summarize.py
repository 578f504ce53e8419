#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/pmc_collect.sh) per kernel: counter sums and per-dispatch averages."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("kp2d::", "")
    return name.split("(")[0][:44]


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                tot[k][c] += float(row["Counter_Value"])
                calls[k][c] += 1
    counters = sorted({c for k in tot for c in tot[k]})
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
        n = max(calls[k].values())
        print(f"== {k}  ({n} dispatches in the profiled run)")
        for c in counters:
            if c in tot[k]:
                print(f"   {c:28s} total {tot[k][c]:16.0f}   per dispatch {tot[k][c] / calls[k][c]:14.1f}")
        t = tot[k]
        if "SQ_WAVE_CYCLES" in t and t["SQ_WAVE_CYCLES"]:
            wc = t["SQ_WAVE_CYCLES"]
            print(f"   -> wait_any {t.get('SQ_WAIT_ANY', 0) / wc:.2%}  wait_inst_any {t.get('SQ_WAIT_INST_ANY', 0) / wc:.2%}"
                  f"  active_inst_any {t.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2%} of wave-cycles")
        if "SQ_BUSY_CYCLES" in t and "SQ_VALU_MFMA_BUSY_CYCLES" in t and t["SQ_BUSY_CYCLES"]:
            print(f"   -> MFMA busy / SQ busy = {t['SQ_VALU_MFMA_BUSY_CYCLES'] / t['SQ_BUSY_CYCLES']:.3f}")
        if "SQ_LDS_IDX_ACTIVE" in t and t["SQ_LDS_IDX_ACTIVE"]:
            print(f"   -> LDS bank-conflict cycles / LDS active = {t.get('SQ_LDS_BANK_CONFLICT', 0) / t['SQ_LDS_IDX_ACTIVE']:.3f}")
        if "FETCH_SIZE" in t or "WRITE_SIZE" in t:
            # gfx950: FETCH_SIZE (KB) under-reports wide streaming reads by 2x -> corrected = 2 * FETCH_SIZE
            rd = t.get("FETCH_SIZE", 0) * 1024 * 2 / max(1, calls[k].get("FETCH_SIZE", 1))
            wr = t.get("WRITE_SIZE", 0) * 1024 / max(1, calls[k].get("WRITE_SIZE", 1))
            print(f"   -> HBM-side traffic per dispatch: read {rd / 1e6:.2f} MB (FETCH_SIZE x2 corrected), write {wr / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
