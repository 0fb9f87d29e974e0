#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 ``*_results.db`` (rocpd SQLite output): name, launches, total, average, share.

    python tools/rocpd_stats.py gpurun_out/.../x_results.db [--by-grid PATTERN] [--top N]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--by-grid", default=None, help="also list launches whose kernel name contains this, per grid")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} launches")
    print(f"{'kernel':100s} {'calls':>6s} {'total ms':>9s} {'avg us':>8s} {'%':>5s}")
    for r in rows[:a.top]:
        print(f"{r[0][:100]:100s} {r[1]:6d} {r[2] / 1e6:9.2f} {r[3] / 1e3:8.1f} {100 * r[2] / tot:5.1f}")
    if a.by_grid:
        q = ("select name, grid_x, grid_y, grid_z, count(*), avg(end-start), vgpr_count, lds_size from kernels where name like ? "
             "group by name, grid_x, grid_y, grid_z order by 6 desc")
        for r in cur.execute(q, (f"%{a.by_grid}%",)):
            print(f"{r[0][:70]:70s} grid {r[1:4]} calls {r[4]} avg {r[5] / 1e3:.1f} us vgpr {r[6]} lds {r[7]}")


if __name__ == "__main__":
    main()
