#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/pmc_collect.sh) per kernel: counter sums and per-dispatch averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


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
        if "SQ_VALU_MFMA_BUSY_CYCLES" in t and t.get("GRBM_GUI_ACTIVE"):
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs (256 CUs x 4);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs -> / 8 = the dispatches' wall cycles (MI355X_MICROARCH.md)
            n_mfma = calls[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
            n_grbm = calls[k]["GRBM_GUI_ACTIVE"]
            busy = (t["SQ_VALU_MFMA_BUSY_CYCLES"] / n_mfma) / 1024.0
            wall = (t["GRBM_GUI_ACTIVE"] / n_grbm) / 8.0
            print(f"   -> mfma_busy_frac = {busy / wall:.3f}  (MFMA busy cycles per SIMD {busy:.0f} / wall cycles {wall:.0f} per dispatch)")
        if t.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in t:
            # SQ_INSTS_VALU includes the MFMAs
            print(f"   -> VALU instructions per MFMA = {(t['SQ_INSTS_VALU'] / calls[k]['SQ_INSTS_VALU']) / (t['SQ_INSTS_MFMA'] / calls[k]['SQ_INSTS_MFMA']) - 1.0:.2f}")
        if "SQ_LDS_IDX_ACTIVE" in t and t["SQ_LDS_IDX_ACTIVE"]:
            print(f"   -> LDS bank-conflict cycles / LDS active = {t.get('SQ_LDS_BANK_CONFLICT', 0) / t['SQ_LDS_IDX_ACTIVE']:.3f}")
        if "FETCH_SIZE" in t or "WRITE_SIZE" in t:
            # gfx950: FETCH_SIZE (KiB) reports half of a WIDE (16 B per lane) streaming read -> x2 for kernels that
            # load that way; kernels with 4-byte-per-lane loads (conv1a's taps) are uncalibrated: raw value shown
            from pmc_traffic import fetch_factor
            ff = fetch_factor(k)
            rd = t.get("FETCH_SIZE", 0) * 1024 * ff / max(1, calls[k].get("FETCH_SIZE", 1))
            wr = t.get("WRITE_SIZE", 0) * 1024 / max(1, calls[k].get("WRITE_SIZE", 1))
            note = "FETCH_SIZE x2: 16-byte-per-lane loads" if ff == 2 else "FETCH_SIZE raw: narrow loads, x2 correction not applicable / uncalibrated"
            print(f"   -> HBM-side traffic per dispatch: read {rd / 1e6:.2f} MB ({note}), write {wr / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
