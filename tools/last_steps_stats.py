"""Per-kernel time of the LAST n steps of a rocprofv3 kernel trace (steady state: library autotuning of the first
steps excluded).  A step boundary is every `per_step`-th dispatch of the kernel whose name contains `marker`.

    python tools/last_steps_stats.py results.db out.md --marker "edge_bwd_reduce_kernel<64" --per-step 2 --steps 2
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("out")
    ap.add_argument("--marker", required=True)
    ap.add_argument("--per-step", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--title", default="")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    rows = con.execute("select name, start, duration/1000.0 from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if a.marker in r[0]]
    ends = marks[a.per_step - 1::a.per_step]                 # index of the last marker dispatch of every step
    if len(ends) < a.steps + 1:
        raise SystemExit("not enough steps in the trace")
    lo, hi = ends[-a.steps - 1] + 1, ends[-1] + 1
    agg = {}
    for name, _, us in rows[lo:hi]:
        t = agg.setdefault(name, [0, 0.0])
        t[0] += 1
        t[1] += us
    total = sum(v[1] for v in agg.values())
    wall = (rows[hi - 1][1] - rows[lo][1]) / 1e3 / a.steps
    with open(a.out, "w") as f:
        f.write("# %s\n\nLast %d steps of the trace (marker `%s`): %.1f ms of kernel time per step in %d dispatches, "
                "%.1f ms between step boundaries.\n\n| kernel | calls/step | avg us | ms/step | %% |\n|---|---|---|---|---|\n"
                % (a.title, a.steps, a.marker, total / 1e3 / a.steps, (hi - lo) // a.steps, wall))
        for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
            f.write("| `%s` | %.1f | %.1f | %.2f | %.1f |\n" % (name[:110], n / a.steps, us / n, us / 1e3 / a.steps, 100 * us / total))
    print(open(a.out).read()[:3000])


if __name__ == "__main__":
    main()
