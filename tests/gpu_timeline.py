"""Device timeline of a rocprofv3 --kernel-trace run (rocpd SQLite): wall time between the first and the last kernel,
time with at least one kernel running, idle gaps, and the busy time per kernel family.  Diagnostic for the
64-mixture search (tests/perf_batch64.py): where the time outside the spot network's launches goes.
Usage: python3 tests/gpu_timeline.py <results.db> [skip_fraction]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0     # drop this leading fraction of the run (setup / warm-up)
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
sc = "start" if "start" in cols else [c for c in cols if "start" in c][0]
ec = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = con.execute(f'select name, "{sc}", "{ec}" from kernels order by "{sc}"').fetchall()
t_first, t_last = rows[0][1], max(r[2] for r in rows)
t0 = t_first + skip * (t_last - t_first)
rows = [r for r in rows if r[1] >= t0]


def family(n):
    n = re.sub(r"\(.*$", "", n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
    n = re.sub(r"<.*$", "", n)
    if n.startswith("at::") or "elementwise" in n or "vectorized" in n or "reduce" in n or "index" in n:
        return "torch:" + n.split("::")[-1][:40]
    return n[:48]


busy, gaps, fam = 0, [], defaultdict(lambda: [0, 0])
cur_s, cur_e = rows[0][1], rows[0][2]
for n, s, e in rows:
    f = fam[family(n)]
    f[0] += 1
    f[1] += e - s
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = max(r[2] for r in rows) - rows[0][1]
print(f"kernels {len(rows)}  wall {wall * 1e-9:.3f} s  busy {busy * 1e-9:.3f} s  idle {(wall - busy) * 1e-9:.3f} s")
for lim in (10e3, 100e3, 1e6, 10e6):
    g = [x for x in gaps if x >= lim]
    print(f"  gaps >= {lim * 1e-3:7.0f} us: {len(g):6d}, {sum(g) * 1e-9:.3f} s")
print("busy time per kernel family:")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {k:50s} {c:7d} launches {t * 1e-9:8.3f} s")
