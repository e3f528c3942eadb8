#!/usr/bin/env python3
"""Per-kernel sums/averages of PMC counters from a rocprofv3 rocpd database (--pmc ... --kernel-trace).
Usage: tools/rocpd_pmc.py db [kernel_name_filter]
       tools/rocpd_pmc.py db --last N kernel     sums over the last N dispatches of a kernel
       tools/rocpd_pmc.py db --each kernel       one line per dispatch of a kernel, in launch order"""
import sqlite3
import sys


def last_dispatches(path, kernel, n):
    """per counter: sum over the last n dispatches of `kernel` (e.g. the timed launches of a bench run)"""
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if "kernel_dispatch" in t)
    ks = next(t for t in tables if "kernel_symbol" in t)
    pe = next(t for t in tables if "rocpd_pmc_event" in t)
    pi = next(t for t in tables if "rocpd_info_pmc" in t)
    icols = [r[1] for r in db.execute(f"pragma table_info('{pi}')")]
    namecol = "symbol" if "symbol" in icols else "name"
    ids = [r[0] for r in db.execute(f"""select d.dispatch_id from {kd} d join {ks} s on d.kernel_id = s.id
                                        where s.display_name like '{kernel}%' order by d.dispatch_id""")][-n:]
    q = f"""select i.{namecol}, sum(p.value) from {pe} p join {pi} i on p.pmc_id = i.id join {kd} d on p.event_id = d.event_id
            where d.dispatch_id in ({",".join(str(i) for i in ids)}) group by i.{namecol}"""
    print("# %s: sums over the last %d dispatches of %s" % (path, len(ids), kernel))
    for name, v in db.execute(q):
        print("%-24s %18.0f   per dispatch %16.0f" % (name, v, v / max(1, len(ids))))


def each_dispatch(path, kernel):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if "kernel_dispatch" in t)
    ks = next(t for t in tables if "kernel_symbol" in t)
    pe = next(t for t in tables if "rocpd_pmc_event" in t)
    pi = next(t for t in tables if "rocpd_info_pmc" in t)
    icols = [r[1] for r in db.execute(f"pragma table_info('{pi}')")]
    namecol = "symbol" if "symbol" in icols else "name"
    q = f"""select d.dispatch_id, i.{namecol}, sum(p.value), max(d.end - d.start), max(d.grid_size_x) from {pe} p join {pi} i on p.pmc_id = i.id
            join {kd} d on p.event_id = d.event_id join {ks} s on d.kernel_id = s.id where s.display_name like '{kernel}%'
            group by d.dispatch_id, i.{namecol} order by d.dispatch_id, i.{namecol}"""
    rows = {}
    for did, name, v, ns, gx in db.execute(q):
        rows.setdefault(did, {"ms": ns / 1e6, "grid": gx})[name] = v
    names = sorted({k for r in rows.values() for k in r if k not in ("ms", "grid")})
    print("# %s: every dispatch of %s in launch order" % (path, kernel))
    print("%4s %10s %9s  " % ("#", "grid_x", "ms") + "  ".join("%22s" % n for n in names))
    for n, (did, r) in enumerate(sorted(rows.items())):
        print("%4d %10d %9.3f  " % (n, r["grid"], r["ms"]) + "  ".join("%22.0f" % r.get(k, 0) for k in names))


def main(path, flt=None):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if "kernel_dispatch" in t)
    ks = next(t for t in tables if "kernel_symbol" in t)
    pe = next(t for t in tables if "rocpd_pmc_event" in t)
    pi = next(t for t in tables if "rocpd_info_pmc" in t)
    cols = [r[1] for r in db.execute(f"pragma table_info('{pe}')")]
    icols = [r[1] for r in db.execute(f"pragma table_info('{pi}')")]
    namecol = "symbol" if "symbol" in icols else "name"
    q = f"""select s.display_name, i.{namecol}, count(*), sum(p.value), avg(p.value), avg(d.end - d.start)
            from {pe} p join {pi} i on p.pmc_id = i.id join {kd} d on p.event_id = d.event_id join {ks} s on d.kernel_id = s.id
            group by s.display_name, i.{namecol} order by s.display_name"""
    print("# PMC per kernel from %s" % path)
    print("%-28s %-24s %7s %18s %18s %10s" % ("kernel", "counter", "calls", "sum", "avg/dispatch", "avg_ms"))
    for r in db.execute(q):
        name = (r[0] or "?").split("(")[0]
        if flt and flt not in name:
            continue
        print("%-28s %-24s %7d %18.0f %18.1f %10.4f" % (name[:28], r[1], r[2], r[3], r[4], r[5] / 1e6))


if __name__ == "__main__" and len(sys.argv) == 5 and sys.argv[2] == "--last":
    last_dispatches(sys.argv[1], sys.argv[4], int(sys.argv[3]))
elif __name__ == "__main__" and len(sys.argv) == 4 and sys.argv[2] == "--each":
    each_dispatch(sys.argv[1], sys.argv[3])
elif __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
