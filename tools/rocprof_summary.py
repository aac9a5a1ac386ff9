"""Turn a rocprofv3 (--kernel-trace --stats) rocpd SQLite database into a committed text summary.

    python tools/rocprof_summary.py gpurun_out/prof/r1_results.db profiles/r01_xxx.md --steps 9 [--title ...]

rocprofv3 7.x writes one .db per run; the `top_kernels` view holds per-kernel call counts, total and
average durations (microseconds).  `--steps` is the number of bench steps the profiled command executed
(warm-up + calibration + timed), used to print per-step figures.
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("out")
    ap.add_argument("--steps", type=float, default=1.0)
    ap.add_argument("--title", default="rocprofv3 kernel-trace summary")
    ap.add_argument("--command", default="")
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    con = sqlite3.connect(args.db)
    rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    total = sum(r[2] for r in rows)
    with open(args.out, "w") as f:
        f.write("# %s\n\n" % args.title)
        if args.command:
            f.write("Command: `%s`\n\n" % args.command)
        f.write("Source: rocprofv3 `top_kernels` view (durations in microseconds); %g bench steps in the run.\n\n"
                % args.steps)
        f.write("Total kernel time: %.1f us = %.1f us per step over %d distinct kernels.\n\n"
                % (total, total / args.steps, len(rows)))
        f.write("| kernel | calls | calls/step | avg us | total us | us/step | % |\n|---|---|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows[:args.top]:
            short = name if len(name) <= 110 else name[:107] + "..."
            f.write("| `%s` | %d | %.1f | %.2f | %.1f | %.1f | %.2f |\n"
                    % (short.replace("|", "/"), calls, calls / args.steps, avg, tot, tot / args.steps, pct))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
