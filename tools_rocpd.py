#!/usr/bin/env python
"""Summarise rocprofv3 rocpd databases (kernel stats + PMC counters per kernel) as text."""
import re, sqlite3, sys

def short(n):
    n = re.sub(r"effocr::\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:95]

def stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':95s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for n, cnt, s, a, mn, mx in rows:
        print(f"{short(n):95s} {cnt:6d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")

def pmc(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {namecol}, counter_name, count(*), avg(value), sum(value) from counters_collection group by {namecol}, counter_name").fetchall()
    by = {}
    for n, cn, cnt, avg, sm in rows:
        by.setdefault(n, {})[cn] = (cnt, avg)
    ctrs = sorted({cn for v in by.values() for cn in v})
    print("per-dispatch averages")
    print(f"{'kernel':70s} " + " ".join(f"{c[:18]:>18s}" for c in ctrs))
    for n, v in sorted(by.items(), key=lambda kv: -max(x[1] for x in kv[1].values())):
        print(f"{short(n)[:70]:70s} " + " ".join(f"{v.get(c,(0,0))[1]:18.4g}" for c in ctrs))

if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    (stats if mode == "stats" else pmc)(db)
