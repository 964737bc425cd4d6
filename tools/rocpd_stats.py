"""Kernel statistics from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`, ROCm 7):
writes the same columns as the CSV `--stats` summary.  Usage: python tools/rocpd_stats.py results.db out.csv [--step-kernel adam_kernel]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    rows = list(c.execute('select name, start, end from kernels order by start'))
    agg = {}
    for name, s, e in rows:
        d = agg.setdefault(name, [0, 0, 1 << 62, 0])
        dur = e - s
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    tot = sum(d[1] for d in agg.values())
    with open(out, 'w', newline='') as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for name, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, d[0], d[1], round(d[1] / d[0], 1), round(100.0 * d[1] / tot, 2), d[2], d[3]])
    steps = sum(1 for r in rows if 'adam_kernel' in r[0])
    print('kernels %d, total %.2f ms, optimizer steps %d, %.3f ms of kernel time per step' % (len(rows), tot / 1e6, steps, tot / 1e6 / max(1, steps)))


if __name__ == '__main__':
    main()
