"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch).
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md 'HBM')."""
import csv, collections, glob, json, sys
def load(d, name):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != name: continue
        k = r['Kernel_Name']
        k = 'conv_gemm_kernel' if 'conv_gemm_kernel' in k else 'narrow_conv_kernel' if 'narrow_conv_kernel' in k else 'resident_chain_kernel' if 'resident_chain_kernel' in k else k[:40]
        out[k].append(float(r['Counter_Value']))
    return out
fe, wr = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
res = {}
for k in fe:
    n = len(fe[k])
    f_kb = 2.0 * sum(fe[k]) / n
    w_kb = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1)
    res[k] = dict(dispatches=n, fetch_bytes_per_launch=f_kb * 1024, write_bytes_per_launch=w_kb * 1024,
                  hbm_bytes_per_launch=(f_kb + w_kb) * 1024)
top = dict(sorted(res.items(), key=lambda kv: -kv[1]['dispatches'] * kv[1]['hbm_bytes_per_launch'])[:8])
print(json.dumps(top, indent=1))
