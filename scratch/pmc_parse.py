import sqlite3, sys, re, json, collections
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
pat=sys.argv[2] if len(sys.argv)>2 else 'raster'
rows=cur.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for k,c,v in rows:
    if re.search(pat,k): agg[re.sub(r'\(.*','',k.replace('(anonymous namespace)::','').replace('void ',''))[:60]][c].append(v)
for k,d in agg.items():
    print(k)
    for c,vs in sorted(d.items()):
        print('   %-28s n=%4d avg %.4g'%(c,len(vs),sum(vs)/len(vs)))
