import sqlite3,re,sys
db=sqlite3.connect(sys.argv[1]); c=db.cursor()
pat=sys.argv[2] if len(sys.argv)>2 else '.'
rows=c.execute("select name, count(*), avg(end-start), min(end-start), max(grid_x) from kernels group by name").fetchall()
for r in sorted(rows,key=lambda r:-r[2]*r[1]):
    m=re.search(r'(k_\w+)(<[^>]*>)?',r[0]); nm=m.group(0) if m else r[0][:40]
    if re.search(pat,nm): print("%-40s %5d %9.1f %9.1f grid %d"%(nm,r[1],r[2]/1e3,r[3]/1e3,r[4]))
