"""Timeline of the LAST bench step of a rocprofv3 kernel trace (works for hipGraph replays too): start
offset, duration and queue of every dispatch, plus how much of the step had no kernel running at all
(launch gaps / dependency bubbles) and how much had two or more (stream overlap)."""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    marker = sys.argv[3] if len(sys.argv) > 3 else "frustum_variance"
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = con.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    lo, hi = (idx[-2], idx[-1]) if len(idx) > 1 else (0, len(rows))
    step = rows[lo:hi]
    t0 = step[0][1]
    t1 = rows[hi][1] if hi < len(rows) else max(r[2] for r in step)
    events = []
    for r in step:
        events.append((r[1], 1))
        events.append((min(r[2], t1), -1))
    events.sort()
    depth, last, idle, multi = 0, t0, 0, 0
    for t, d in events:
        if depth == 0:
            idle += t - last
        elif depth >= 2:
            multi += t - last
        depth += d
        last = t
    idle += max(0, t1 - last)
    queues = sorted(set(r[3] for r in step))
    with open(out, "w") as f:
        f.write("step window %.1f us, %d dispatches on %d queues; no kernel running %.1f us, >= 2 running %.1f us, "
                "sum of kernel durations %.1f us\n" % ((t1 - t0) / 1e3, len(step), len(queues), idle / 1e3, multi / 1e3,
                                                     sum(r[2] - r[1] for r in step) / 1e3))
        prev_end = {}
        for r in step:
            q = r[3]
            gap = (r[1] - prev_end[q]) / 1e3 if q in prev_end else 0.0
            prev_end[q] = r[2]
            f.write("%9.1f +%7.1f us (gap on queue %5.1f) q%-3s %s\n" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap,
                                                                      queues.index(q), r[0][:100]))
    print(open(out).readline().strip())


if __name__ == "__main__":
    main()
