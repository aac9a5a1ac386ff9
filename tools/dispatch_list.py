"""Dump the kernel dispatches of the LAST bench step of a rocprofv3 kernel trace as compact text
(name, microseconds, grid, workgroup), for reading a step's timeline without shipping the .db."""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    marker = sys.argv[3] if len(sys.argv) > 3 else "frustum_variance"
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, duration/1000.0, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size "
                       "from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    lo = idx[-2] if len(idx) > 1 else 0
    hi = idx[-1]
    with open(out, "w") as f:
        for r in rows[lo:hi]:
            f.write("%9.1f us  grid(%d,%d,%d) wg %d vgpr %d lds %d  %s\n" % (r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[0][:120]))
    print("wrote", out, hi - lo, "dispatches")


if __name__ == "__main__":
    main()
