import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# rows: one per (dispatch, counter)
d = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    name = r['Kernel_Name']
    name = name[name.find('trebi::') + 7:][:26] if 'trebi::' in name else name[:26]
    key = (name, r['Grid_Size'], r['Workgroup_Size'], r.get('LDS_Block_Size', ''))
    d[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES':
        cnt[key] += 1
names = sorted({r['Counter_Name'] for r in rows})
print('kernel grid wg lds | dispatches | per-wave:', ' '.join(names))
for key, c in sorted(d.items(), key=lambda kv: -kv[1].get('SQ_WAVES', 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    w = max(c.get('SQ_WAVES', 1), 1)
    print(key, cnt[key], ' '.join(f'{c[n] / w:9.1f}' for n in names))
