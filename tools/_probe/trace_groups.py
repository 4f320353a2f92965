import sqlite3, glob, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0]); cur = db.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
rows = [r for r in rows if sys.argv[2] in r[0]]
n = int(sys.argv[3])
for i in range(0, len(rows), n):
    grp = rows[i:i+n]
    print(grp[0][0][5:75], [round((e-s)/1e3,1) for _, s, e in grp][1:])
