"""Per-queue busy time and overlap of the LAST step of a rocprofv3 kernel trace (rocpd SQLite): how much of the step's
kernel time ran while another hardware queue was busy too.

    python tools/stream_overlap.py <results.db> <out.md> --marker conv3d_k3_pair_kernel
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("out")
    ap.add_argument("--marker", default="conv3d_k3_pair_kernel")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    cols = [d[0] for d in con.execute("select * from kernels limit 1").description]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = con.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
    idx = [i for i, r in enumerate(rows) if a.marker in r[0]]
    lo, hi = (idx[-2], idx[-1]) if len(idx) > 1 else (0, len(rows))
    step = rows[lo:hi]
    t0, t1 = step[0][1], max(r[2] for r in step)
    per_q = {}
    events = []
    for name, s, e, q in step:
        per_q.setdefault(q, [0, 0.0])
        per_q[q][0] += 1
        per_q[q][1] += (e - s) / 1e3
        events.append((s, 1))
        events.append((e, -1))
    events.sort()
    depth, last, busy1, busy2 = 0, events[0][0], 0.0, 0.0
    for t, d in events:
        if depth >= 1:
            busy1 += (t - last) / 1e3
        if depth >= 2:
            busy2 += (t - last) / 1e3
        depth += d
        last = t
    total = sum(v[1] for v in per_q.values())
    with open(a.out, "w") as f:
        f.write("# Queues of the last step of the trace (marker `%s`)\n\n" % a.marker)
        f.write("%d dispatches between the last two markers, %.1f us from the first kernel's start to the last one's end.\n\n"
                % (len(step), (t1 - t0) / 1e3))
        f.write("| %s | dispatches | kernel time us |\n|---|---|---|\n" % (qcol or "queue"))
        for q, (n, us) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
            f.write("| %s | %d | %.1f |\n" % (q, n, us))
        f.write("\nSum of kernel durations %.1f us; time with at least one kernel running %.1f us; with at least two running "
                "%.1f us (%.0f %% of the sum ran beside another kernel).\n" % (total, busy1, busy2, 100.0 * 2 * busy2 / max(total, 1e-9)))
    print(open(a.out).read())


if __name__ == "__main__":
    main()
