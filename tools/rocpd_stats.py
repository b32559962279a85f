"""Per-kernel duration summary of a rocprofv3 .db (rocpd sqlite) result: python tools/rocpd_stats.py <file.db> [name filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
name = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
    "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name, kd, ks, name)).fetchall()
tot = sum(r[2] for r in rows)
print("%-70s %7s %12s %10s %9s %9s %6s" % ("kernel", "calls", "total us", "avg us", "min us", "max us", "%"))
for r in rows:
    if flt in r[0]:
        print("%-70s %7d %12.1f %10.2f %9.2f %9.2f %6.2f" % (r[0][:70], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
