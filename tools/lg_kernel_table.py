#!/usr/bin/env python3
"""Per-kernel table of the LightGlue matcher from a rocprofv3 kernel trace (rocpd sqlite output).

    rocprofv3 --kernel-trace -d DIR -o NAME -- python3 tools/bench_lightglue.py --pairs 1
    python3 tools/lg_kernel_table.py DIR/NAME_results.db

Prints calls per matcher forward, the average duration and the per-forward total of every lg_* / attention kernel.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on "
                           f"d.kernel_id=s.id group by 1 order by 3*2 desc"))
    fwd = max((c for n, c, _ in rows if "lg_filter" in n), default=1)      # one filter launch per forward
    total = 0.0
    print(f"{'kernel':58s} {'per fwd':>7s} {'avg us':>8s} {'us/fwd':>8s}")
    for name, c, avg in sorted(rows, key=lambda r: -r[1] * r[2]):
        if not any(k in name for k in ("lg_", "attention")):
            continue
        short = name.split("kp2d")[-1][:58]
        print(f"{short:58s} {c / fwd:7.1f} {avg / 1e3:8.1f} {c * avg / fwd / 1e3:8.1f}")
        total += c * avg / fwd / 1e3
    print(f"{'sum of matcher kernels per forward':58s} {'':7s} {'':8s} {total:8.1f}")


if __name__ == "__main__":
    main()
