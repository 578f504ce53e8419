#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 kernel trace in rocpd (sqlite) form: calls, average and total duration.

    rocprofv3 --kernel-trace -d DIR -o NAME -- python3 <program>
    python3 tools/kernel_table.py DIR/NAME_results.db [calls-per-iteration divisor]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s "
                           f"on d.kernel_id=s.id group by 1 order by 4 desc"))
    tot = sum(r[3] for r in rows)
    print(f"{'kernel':70s} {'calls':>8s} {'avg us':>8s} {'total us':>10s} {'%':>6s}")
    for name, c, avg, s in rows:
        short = name.replace("_ZN4kp2d", "")[:70]
        print(f"{short:70s} {c / div:8.1f} {avg / 1e3:8.1f} {s / div / 1e3:10.1f} {100 * s / tot:6.1f}")


if __name__ == "__main__":
    main()
