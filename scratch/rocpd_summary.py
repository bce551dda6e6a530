"""Summarise a rocprofv3 rocpd sqlite database: per-kernel totals (ms per forward) and the busy/span ratio."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); nf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, count(*), sum(end-start)/1e6, min(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows); n = sum(r[1] for r in rows)
print(f"kernel time {tot/nf:.3f} ms/fwd, launches {n/nf:.1f}/fwd")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    nm = re.sub(r"\(anonymous namespace\)::|ovm::|void ", "", r[0])
    print(f"{r[2]/nf:8.3f} ms/fwd {r[1]/nf:7.1f} calls  avg {r[2]/r[1]*1e3:7.1f} us  min {r[3]:6.1f}  {nm[:70]}")
