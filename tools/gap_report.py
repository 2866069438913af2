"""Where is the GPU idle during a bench step?  Reads a rocprofv3 --kernel-trace CSV and prints the idle gaps between kernels.

usage: python tools/gap_report.py <..._kernel_trace.csv> [min gap us = 30]

Kernels of all streams are merged into one busy timeline (union of [start, end)); every interval with no kernel running that is
longer than the threshold is listed with the kernel that ended before it and the one that started after it, plus totals: span from
the first to the last kernel, busy time, idle time in gaps above / below the threshold.  (The decode loop's launch-to-launch gaps are
a few microseconds each and show up in the 'below' total.)"""
import csv
import sys


def main():
    path = sys.argv[1]
    thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 30e3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy_end, last = rows[0][1], rows[0][2]
    busy = 0
    cur_start = rows[0][0]
    big, small = [], 0
    for s, e, n in rows[1:]:
        if s > busy_end:
            busy += busy_end - cur_start
            g = s - busy_end
            if g >= thr:
                big.append((busy_end - t0, g, last, n))
            else:
                small += g
            cur_start = s
        if e > busy_end:
            busy_end, last = e, n
    busy += busy_end - cur_start
    print(f"{len(rows)} kernels, span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle in gaps < {thr / 1e3:.0f} us {small / 1e6:.2f} ms, "
          f"in {len(big)} larger gaps {sum(g for _, g, _, _ in big) / 1e6:.2f} ms")
    for at, g, a, b in big:
        print(f"  +{at / 1e6:9.2f} ms  idle {g / 1e3:8.1f} us   after {a:<60} before {b}")


if __name__ == "__main__":
    main()
