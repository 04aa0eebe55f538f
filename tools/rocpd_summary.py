#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel dispatch count / avg / min / max
duration (ns) and, when present, PMC counter values per kernel.  Writes a text table to stdout.
Usage: python tools/rocpd_summary.py [--hist] results.db [more.db ...]
--hist adds, per kernel, the percentiles of the dispatch durations and a coarse histogram."""
import sqlite3
import sys


def summarise_by_grid(path):
    """Per (kernel, grid size): dispatch count and durations; PMC values per dispatch, and -- for counters that are
    reported per hardware instance (no _sum suffix: one row per XCD / channel) -- the spread over the instances."""
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    print(f"== {path}")
    q = f"""select s.kernel_name, d.grid_size_x, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.workgroup_size_x)
            from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name, d.grid_size_x order by sum(d.end-d.start) desc"""
    print(f"{'calls':>6} {'avg_ns':>12} {'min_ns':>12} {'grid':>11} {'wg':>5}  kernel")
    for r in cur.execute(q):
        print(f"{r[2]:6d} {r[3]:12.0f} {r[4]:12d} {r[1]:11d} {r[5]:5d}  {r[0][:90]}")
    pm = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")]
    if not (pm and pi):
        return
    q = f"""select s.kernel_name, d.grid_size_x, p.name, e.event_id, e.value
            from {pm[0]} e join {pi[0]} p on e.pmc_id = p.id
            join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id"""
    acc = {}
    try:
        for name, grid, pname, ev, val in cur.execute(q):
            acc.setdefault((name, grid, pname), {}).setdefault(ev, []).append(val)
    except sqlite3.Error as e:
        print("   (pmc query failed:", e, ")")
        return
    print("-- PMC per dispatch (mean over the dispatches); instance spread for per-instance counters")
    for (name, grid, pname), evs in sorted(acc.items()):
        if grid < 1024:
            continue
        totals = [sum(v) for v in evs.values()]
        line = f"   grid {grid:10d} {pname:44s} {sum(totals) / len(totals):18.1f}"
        inst = max(len(v) for v in evs.values())
        if inst > 1:
            v = sorted(list(evs.values())[-1])
            mean = sum(v) / len(v)
            line += f"   [{inst} instances: min {v[0]:.0f} median {v[len(v) // 2]:.0f} max {v[-1]:.0f}, max/mean {v[-1] / mean if mean else 0:.2f}]"
        print(line + f"   {name[:40]}")


def summarise(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    print(f"== {path}")
    q = f"""select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                   s.arch_vgpr_count, s.sgpr_count, max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x)
            from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by sum(d.end-d.start) desc"""
    print(f"{'calls':>6} {'avg_ns':>10} {'min_ns':>10} {'max_ns':>10} {'vgpr':>5} {'sgpr':>5} {'lds':>7} {'grid':>9} {'wg':>4}  kernel")
    for r in cur.execute(q):
        name = r[0]
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"{r[1]:6d} {r[2]:10.0f} {r[3]:10d} {r[4]:10d} {r[5]:5d} {r[6]:5d} {r[7]:7d} {r[8]:9d} {r[9]:4d}  {short}")
    pm = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")]
    if pm and pi:
        try:
            q = f"""select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value)
                    from {pm[0]} e join {pi[0]} p on e.pmc_id = p.id
                    join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id
                    group by s.kernel_name, p.name order by s.kernel_name, p.name"""
            rows = list(cur.execute(q))
            if rows:
                print("-- PMC (per-dispatch average)")
                for r in rows:
                    nm = r[0] if len(r[0]) < 70 else r[0][:67] + "..."
                    print(f"   {r[1]:28s} {r[3]:16.1f}   n={r[2]:<5d} {nm}")
        except sqlite3.Error as e:
            print("   (pmc query failed:", e, ")")
            for t in pm + pi:
                print("   ", t, [c[1] for c in cur.execute(f"pragma table_info('{t}')")])


def histogram(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    names = [r[0] for r in cur.execute(f"select distinct s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id")]
    for name in names:
        rows = list(cur.execute(f"select d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id where s.kernel_name = ? order by d.start", (name,)))
        if len(rows) < 20:
            continue
        dur = sorted(e - s for s, e in rows)
        gaps = sorted(rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1))
        starts = sorted(rows[i + 1][0] - rows[i][0] for i in range(len(rows) - 1))
        q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]  # noqa: E731
        print(f"-- {name[:100]}")
        print(f"   n={len(dur)}  duration ns: min {dur[0]} p5 {q(dur, .05)} p25 {q(dur, .25)} p50 {q(dur, .5)} p75 {q(dur, .75)} p95 {q(dur, .95)} max {dur[-1]}  mean {sum(dur) / len(dur):.0f}")
        print(f"   start-to-start ns (back-to-back cadence): p5 {q(starts, .05)} p25 {q(starts, .25)} p50 {q(starts, .5)} p75 {q(starts, .75)} p95 {q(starts, .95)}")
        print(f"   end-to-next-start gap ns: p5 {q(gaps, .05)} p50 {q(gaps, .5)} p95 {q(gaps, .95)}")
        lo, hi = dur[0], q(dur, .99)
        nb = 16
        w = max(1, (hi - lo + nb) // nb)
        cnt = [0] * (nb + 1)
        for d in dur:
            cnt[min(nb, (d - lo) // w)] += 1
        for b, c in enumerate(cnt):
            if c:
                print(f"   {lo + b * w:7d}..{lo + (b + 1) * w:7d} ns {c:7d} {'#' * max(1, 60 * c // len(dur))}")


args = sys.argv[1:]
hist = "--hist" in args
bygrid = "--by-grid" in args
for p in [a for a in args if a not in ("--hist", "--by-grid")]:
    if bygrid:
        summarise_by_grid(p)
        continue
    summarise(p)
    if hist:
        histogram(p)
