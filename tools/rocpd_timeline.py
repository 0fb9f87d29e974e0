#!/usr/bin/env python3
"""Kernel timeline of one replayed step from a rocprofv3 rocpd database (the default output format): every launch between two
consecutive step_begin kernels with start, duration, queue and grid.  usage: rocpd_timeline.py results.db [steps from the end]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1]).cursor()
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    name_col = [r[1] for r in c.execute(f"pragma table_info({ks})") if r[1] in ("kernel_name", "display_name")][0]
    rows = list(c.execute(f"select s.{name_col}, d.start, d.end, d.queue_id, d.grid_size_x, d.grid_size_y, d.grid_size_z "
                          f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    idx = [i for i, r in enumerate(rows) if "step_begin" in r[0]]
    a, b = idx[-back - 1], idx[-back]
    t0 = rows[a][1]
    queues = {}
    busy = 0
    last_end = t0
    for n, s, e, q, gx, gy, gz in rows[a:b]:
        qn = queues.setdefault(q, len(queues))
        short = n.split("(")[0].replace("_ZN3dua", "").replace("void dua::", "").replace("dua::", "")[:44]
        print(f"{(s - t0) / 1e3:9.1f}us dur {(e - s) / 1e3:7.1f}us q{qn} grid {gx:>7},{gy},{gz}  {short}")
        if e > last_end:
            busy += e - max(s, last_end)
            last_end = e
    span = rows[b][1] - t0
    print(f"step span {span / 1e3:.1f} us, some kernel running for {busy / 1e3:.1f} us of it")


if __name__ == "__main__":
    main()
