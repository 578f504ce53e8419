#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_collect.sh into profiles/<name>.json:
HBM-side bytes per launch of each kernel family, with the gfx950 correction the microarch guide prescribes
(FETCH_SIZE reports half of a wide streaming read -> x2; WRITE_SIZE is exact; both are in KiB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def family(name):
    n = name.replace("void ", "").replace("kp2d::", "").split("(")[0]
    if n.startswith("conv3x3_f32_kernel"):
        args = n[n.index("<") + 1:n.index(">")].split(",")
        taps, prec = args[2].strip(), args[3].strip()
        return ("conv3x3" if taps == "9" else "conv1x1") + ("_f16x3" if prec == "1" else "_f32")
    if n.startswith("head3x3_kernel"):
        return "conv3x3_head"
    # conv3x3_f16.hip / conv3x3_wsm.hip: every tile form of the split-fp16 3x3 convolution is ONE family, the one bench.py's
    # roofline object and its algorithmic_bytes_per_launch cover (round 3 kept the warp-specialised conv1b kernel under its
    # own key, so `traffic` covered 24 of the 25 launches)
    if n.startswith("conv3x3_f16x3"):
        return "conv3x3_f16x3"
    return n.split("<")[0].replace("_kernel", "")


# Kernels whose global loads are 16 bytes per lane (buffer_load_dwordx4 / float4): the only access width the guide's
# "FETCH_SIZE reports exactly half" calibration covers.  Everything else (conv1a's 4-byte tap loads, torch's elementwise
# kernels, copies) keeps the raw counter: "other access widths are uncalibrated" (MI355X_MICROARCH.md, HBM).
WIDE_LOAD_KERNELS = ("conv3x3_f16x3", "conv3x3_f32", "conv1x1", "conv3x3_head", "head3x3", "netvlad_partial", "netvlad_finish",
                     "attention", "channel_layernorm", "dwconv3x3", "seg_argmax4", "post", "lg_")


def fetch_factor(kernel_name):
    n = kernel_name.replace("void ", "").replace("kp2d::", "")
    return 2 if any(n.startswith(w) or family(n).startswith(w) for w in WIDE_LOAD_KERNELS) else 1


def main():
    d, out = sys.argv[1], sys.argv[2]
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(d, "tcc_*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            k = family(row["Kernel_Name"])
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    res = {}
    for k in tot:
        if "FETCH_SIZE" in tot[k] and "WRITE_SIZE" in tot[k]:
            ff = fetch_factor(k)
            rd = tot[k]["FETCH_SIZE"] * 1024 * ff / cnt[k]["FETCH_SIZE"]
            wr = tot[k]["WRITE_SIZE"] * 1024 / cnt[k]["WRITE_SIZE"]
            res[k] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
                      "fetch_size_factor": ff, "launches_profiled": cnt[k]["FETCH_SIZE"]}
    meta = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_collect.sh on tools/layer_profile.py "
                      "(KP2DTiny-S 240x320, 64 frames); KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read correction) only for kernels "
                      "with 16-byte-per-lane loads (fetch_size_factor 2), raw otherwise (uncalibrated access widths)",
            "kernels": res}
    json.dump(meta, open(out, "w"), indent=1)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
