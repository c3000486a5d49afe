import re,collections,sys
rows=[l.split('|') for l in open(sys.argv[1]) if l.startswith('| ') and l[2].isdigit()]
agg=collections.OrderedDict()
for r in rows:
    k,st,du,gap,wg,name=[x.strip() for x in r[1:7]]
    name=re.sub(r'\(.*','',name.strip('`')).replace('tsd::','').replace('(anonymous namespace)::','')
    name=name.replace('void ','').replace('at::native::','at:')[:50]
    a=agg.setdefault(name,[0,0.0]); a[0]+=1; a[1]+=float(du)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 22]:
    print(f"{v[1]:8.1f} us  x{v[0]:3d}  avg {v[1]/v[0]:6.1f}  {k}")
print(sum(v[1] for v in agg.values()))
