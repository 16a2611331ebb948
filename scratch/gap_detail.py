import sqlite3, sys, re
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
rows=cur.execute("select name,start,end from kernels order by start").fetchall()
def short(n): return re.sub(r'\(.*','',n.replace('(anonymous namespace)::','').replace('void ',''))[:90]
cnt=0
for i in range(1,len(rows)):
    g=rows[i][1]-rows[i-1][2]
    if g>3e6 and i>len(rows)//3:
        cnt+=1
        if cnt>3: break
        print('---- gap %.2f ms'%(g/1e6))
        for j in range(max(0,i-6), min(len(rows), i+6)):
            print('  %s%8.1f us  +%8.1f  %s'%('>>' if j==i else '  ', (rows[j][2]-rows[j][1])/1e3, (rows[j][1]-rows[j-1][2])/1e3, short(rows[j][0])))
