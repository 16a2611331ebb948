import sqlite3, sys, re, collections
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
rows=cur.execute("select name,start,end from kernels order by start").fetchall()
# timed region: find the last convert_depth_kernel occurrences -> frames; use last N frames
frames=[r[1] for r in rows if 'convert_depth_kernel' in r[0]]
nf=int(sys.argv[2]) if len(sys.argv)>2 else 100
t0=frames[-nf]; t1=rows[-1][2]
sel=[r for r in rows if r[1]>=t0]
# drop the trailing roofline microbench: stop at the end of the kernel preceding the 50x repeated raster launches
# (approximation: stop at last adam_kernel end + 5 ms)
last_adam=max(r[2] for r in sel if 'adam_kernel' in r[0])
sel=[r for r in sel if r[1]<=last_adam]
t1=last_adam
busy=sum(r[2]-r[1] for r in sel)
wall=t1-t0
print('frames %d wall %.3f ms (%.3f ms/frame) kernel-busy %.3f ms (%.1f%%) dispatches %d (%.1f/frame)'%(nf,wall/1e6,wall/1e6/nf,busy/1e6,100*busy/wall,len(sel),len(sel)/nf))
gaps=collections.defaultdict(lambda:[0,0])
prev=None
big=[]
for r in sel:
    if prev is not None:
        g=r[1]-prev[2]
        if g>0:
            key=re.sub(r'\(.*','',prev[0].replace('(anonymous namespace)::','').replace('void ',''))[:40]+' -> '+re.sub(r'\(.*','',r[0].replace('(anonymous namespace)::','').replace('void ',''))[:40]
            gaps[key][0]+=g; gaps[key][1]+=1
    prev=r
tot=sum(v[0] for v in gaps.values())
print('total gap %.3f ms'%(tot/1e6))
for k,v in sorted(gaps.items(), key=lambda kv:-kv[1][0])[:25]:
    print('%8.3f ms  n=%5d avg %7.1f us  %s'%(v[0]/1e6,v[1],v[0]/v[1]/1e3,k))
