"""Idle time of the GPU inside the training step, from a rocprofv3 --kernel-trace CSV: the dispatches of the last steps are
cut at the optimizer kernel, and per step the union of the busy intervals (all queues) is compared with the step's span.

usage: python tools/trace_gaps.py <kernel_trace.csv>
"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = []
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    ev.sort()
    ends = [i for i, e in enumerate(ev) if "rmsprop_kernel" in e[2]]
    if len(ends) < 4:
        print("not enough steps in the trace")
        return
    print("steps in trace: %d" % len(ends))
    for a, b in zip(ends[-4:-1], ends[-3:]):
        step = ev[a + 1:b + 1]
        t0, t1 = ev[a][1], ev[b][1]
        span = (t1 - t0) / 1e3
        # union of busy intervals
        busy, cur_s, cur_e = 0, None, None
        for s, e, _, _ in step:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        total = sum(e - s for s, e, _, _ in step)
        queues = sorted({q for _, _, _, q in step})
        # gaps (no kernel running) longer than 2 us
        gaps = []
        cur_e, prev = ev[a][1], ev[a][2]
        for s, e, n, _ in step:
            if s > cur_e:
                gaps.append((s - cur_e, cur_e, prev, n))
            if e >= cur_e:
                cur_e, prev = e, n
        print("step span %.1f us, busy (union) %.1f us = %.1f %%, idle %.1f us in %d gaps; sum of kernel durations %.1f us; %d dispatches on queues %s"
              % (span, busy / 1e3, 100.0 * busy / 1e3 / span, span - busy / 1e3, len(gaps), total / 1e3, len(step), queues))
        hist = [0, 0, 0, 0]
        for g, _, _, _ in gaps:
            hist[0 if g < 1000 else 1 if g < 3000 else 2 if g < 10000 else 3] += g
        print("   idle by gap length: <1us %.1f us, 1-3us %.1f us, 3-10us %.1f us, >10us %.1f us" % tuple(h / 1e3 for h in hist))
        for g, at, prev, nxt in sorted(gaps, reverse=True)[:12]:
            print("   gap %7.1f us at +%8.1f us: after %-44s before %s" % (g / 1e3, (at - t0) / 1e3, short(prev), short(nxt)))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("hnconv::", "").replace("void ", "")
    return n.split("(")[0][:44]


if __name__ == "__main__":
    main()
