#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of --kernel-trace): per-step time of the top kernels, split by
grid size so the launches of one template on different shapes stay apart.  usage: prof_db.py <results.db> <steps> [top]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.display_name, d.grid_size_x, count(*), sum(d.end-d.start), s.arch_vgpr_count, s.accum_vgpr_count "
                      f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.display_name, d.grid_size_x order by 4 desc"))
tot = sum(r[3] for r in rows)
for n, g, cnt, ns, v, a in rows[:top]:
    print(f"{n[:80]:80s} grid {g:>8d} {cnt:>6d} {ns/steps/1e3:9.1f}us/step {ns/cnt/1e3:8.1f}us {ns/tot*100:5.1f}%  vgpr {v}+{a}")
print(f"{tot/steps/1e6:.3f} ms/step GPU busy")
