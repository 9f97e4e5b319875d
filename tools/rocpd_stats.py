#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table
(calls, total / average / min / max duration, share), like `--stats` CSV output.

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [--top 40] [--skip-first-ms X]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--name-width", type=int, default=90)
    ap.add_argument("--tail-ms", type=float, default=0.0,
                    help="only dispatches that start within the last X ms of the trace (e.g. the timed steps)")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from kernels").fetchall()
    if args.tail_ms > 0:
        t_last = max(e for _, _, e in rows)
        rows = [r for r in rows if r[1] >= t_last - args.tail_ms * 1e6]
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# {len(rows)} dispatches, {len(agg)} kernels, total kernel time {total / 1e6:.3f} ms")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  name")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"{a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:10.2f} {a[3] / 1e3:10.2f} "
              f"{100 * a[1] / total:6.2f}  {name[:args.name_width]}")


if __name__ == "__main__":
    main()
