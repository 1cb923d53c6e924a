"""Fold a rocprofv3 --kernel-trace CSV into a timeline summary: busy time per queue, union busy time, idle gaps, and
the kernels that run while no other queue is busy (the ones on the critical path of a multi-stream step).

    python tools/trace_timeline.py <..._kernel_trace.csv> [--skip-frac 0.3] [--top 30]

--skip-frac drops the leading part of the trace (model construction / warm-up)."""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--skip-frac", type=float, default=0.4)
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + int((t1 - t0) * a.skip_frac)
    rows = [r for r in rows if r[0] >= cut]
    span = rows[-1][1] - rows[0][0]
    per_q = collections.Counter()
    for s, e, q, n in rows:
        per_q[q] += e - s
    # sweep: union busy + time each kernel spends as the ONLY running kernel
    events = []
    for i, (s, e, q, n) in enumerate(rows):
        events.append((s, 1, i)); events.append((e, 0, i))
    events.sort()
    active = set()
    last = events[0][0]
    union = 0
    solo = collections.Counter()
    for t, kind, i in events:
        if active:
            union += t - last
            if len(active) == 1:
                solo[rows[next(iter(active))][3]] += t - last
        last = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    print(f"span {span / 1e6:.2f} ms  union-busy {union / 1e6:.2f} ms ({100 * union / span:.1f} %)  idle {(span - union) / 1e6:.2f} ms")
    for q, v in per_q.most_common():
        print(f"  queue {q}: busy {v / 1e6:.2f} ms ({100 * v / span:.1f} %)")
    print("kernels by time spent running ALONE (critical path candidates):")
    tot = sum(solo.values())
    for n, v in solo.most_common(a.top):
        print(f"  {v / 1e6:9.3f} ms {100 * v / tot:5.1f} %  {n[:110]}")


if __name__ == "__main__":
    main()
