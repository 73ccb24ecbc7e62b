"""Longest HIP API calls of a rocprofv3 --hip-trace run (rocpd database): python tools/hip_api_top.py <file.db>.
Used with tools/first_call.py to see what the first GPU call of a process spends its milliseconds on (the lazy
code-object load inside the first hipLaunchKernel)."""
import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
cols=[r[1] for r in c.execute("pragma table_info(regions)")]
print(cols)
rows=c.execute("select name, (end-start)/1e6 as ms, start from regions order by ms desc limit 25").fetchall()
t0=min(r[2] for r in c.execute("select name,0,start from regions").fetchall())
for n,ms,st in rows: print("%-40s %8.3f ms at %+9.3f ms"%(n[:40],ms,(st-t0)/1e6))
