import csv, collections, glob, sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=collections.defaultdict(list)
for r in rows:
    name=r['Kernel_Name']
    name=name[name.find('trebi::')+7:][:28] if 'trebi::' in name else name[:28]
    key=(name, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),r['Grid_Size_Y'],r['Grid_Size_Z'],r['Workgroup_Size_X'])
    d[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=sum(sum(v) for v in d.values())
print('total kernel time ms', tot/1e6, 'n', len(rows))
# last 3000 kernels: busy vs span
tail=rows[-3000:]
span=int(tail[-1]['End_Timestamp'])-int(tail[0]['Start_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in tail)
print('tail span ms',span/1e6,'busy ms',busy/1e6)
gaps=[int(b['Start_Timestamp'])-int(a['End_Timestamp']) for a,b in zip(tail[:-1],tail[1:])]
gaps.sort()
print('gap median us',gaps[len(gaps)//2]/1e3,'p90',gaps[int(len(gaps)*.9)]/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv)>2 else 25]:
    v2=sorted(v)
    print(f"{sum(v)/tot*100:5.1f}% n={len(v):5d} avg={sum(v)/len(v)/1e3:7.2f}us med={v2[len(v2)//2]/1e3:7.2f} min={v2[0]/1e3:6.2f}", k)
