"""Per-launch-shape view of one or more rocprofv3 --pmc csv runs (same program, different counter sets):
groups dispatches by (kernel template arguments, grid, workgroup size, LDS bytes) and prints, per group, the average duration and
every counter per wave.   usage: analyze_layers.py DIR [DIR ...] [--min-us N]"""
import collections, csv, glob, re, sys
args = sys.argv[1:]
min_us = 0.0
if '--min-us' in args:
    i = args.index('--min-us')
    min_us = float(args[i + 1])
    del args[i:i + 2]
dirs = args
G = collections.defaultdict(lambda: {'n': 0, 'ns': 0.0, 'c': collections.defaultdict(float), 'seen': set(), 'vgpr': 0})
for d in dirs:
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        m = re.search(r'(conv_gemm_kernel|narrow_conv_kernel|resident_chain_kernel|wgrad_kernel|gnb_kernel)(<[^>]*>)?', k)
        if not m:
            continue
        key = (m.group(1) + (m.group(2) or ''), int(r['Grid_Size']), int(r['Workgroup_Size']), int(r['LDS_Block_Size']))
        g = G[key]
        g['vgpr'] = r.get('VGPR_Count', 0)
        tag = (d, r['Dispatch_Id'])
        if tag not in g['seen']:
            g['seen'].add(tag)
            if d == dirs[0]:
                g['n'] += 1
                g['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        g['c'][(d, r['Counter_Name'])] += float(r['Counter_Value'])
rows = sorted(G.items(), key=lambda kv: -kv[1]['ns'])
tot = sum(g['ns'] for _, g in rows)
for key, g in rows:
    us = g['ns'] / max(g['n'], 1) / 1e3
    if us < min_us:
        continue
    name, grid, wg, lds = key
    nd = {d: len([1 for t in g['seen'] if t[0] == d]) for d in dirs}
    waves = {d: g['c'].get((d, 'SQ_WAVES'), 0.0) / max(nd[d], 1) for d in dirs}
    print(f'{name} grid {grid // wg} x {wg} lds {lds} vgpr {g["vgpr"]}: n={g["n"]} avg {us:.1f} us ({100 * g["ns"] / tot:.1f} % of time) '
          f'cycles/launch {us * 2400:.0f}')
    for (d, cn), v in sorted(g['c'].items()):
        if cn == 'SQ_WAVES':
            continue
        per_launch = v / max(nd[d], 1)
        w = waves[d] or (grid / 64)
        print(f'      {cn:32s} {per_launch:14.0f} per launch  {per_launch / w:10.1f} per wave')
