"""Summarise a rocprofv3 --pmc run (rocpd SQLite) into a small JSON: per kernel, mean counter value per
dispatch.  Run on the GPU box right after collection (the .db files are too large to ship back).

    python tools/pmc_summary.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_fetch.json [--by-grid]

--by-grid: one entry per (kernel, grid, LDS bytes) -- the layers that share a template instantiation apart; the mean
duration of the dispatches (ns) rides along as the pseudo counter "duration_ns".
"""
import json
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    by_grid = "--by-grid" in sys.argv[3:]
    con = sqlite3.connect(db)
    cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    if by_grid:
        name_col = ("%s || ' grid=' || grid_size_x || 'x' || grid_size_y || 'x' || grid_size_z || ' lds=' || lds_block_size"
                    % name_col)
    rows = con.execute("select %s, counter_name, avg(value), count(*), min(value), max(value), sum(value) "
                       "from counters_collection group by %s, counter_name" % (name_col, name_col)).fetchall()
    res = {}
    for name, ctr, avg, cnt, mn, mx, tot in rows:
        res.setdefault(name, {})[ctr] = {"mean_per_dispatch": avg, "dispatches": cnt, "min": mn, "max": mx, "sum": tot}
    if by_grid and "duration" in cols:
        for name, avg, cnt in con.execute("select %s, avg(duration), count(*) from counters_collection group by %s"
                                          % (name_col, name_col)).fetchall():
            res[name]["duration_ns"] = {"mean_per_dispatch": avg, "dispatches": cnt}
    json.dump({"columns": cols, "kernels": res}, open(out, "w"), indent=0)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
