"""tools/layer_profile.py table (tsv) -> JSON rows {layer, tile, launches, total_us, avg_us, gflop_per_launch, tflops}.
usage: layers_to_json.py in.tsv out.json [label]"""
import json, re, sys
rows = []
for l in open(sys.argv[1]).read().splitlines()[1:]:
    n, tot, avg, tf, name = l.split('\t')
    m = re.match(r'(.*?) ((?:MI\d.*)|(?:S\d+ .*)|(?:narrow N\d+.*))$', name) if 'chain[' not in name else None
    layer, tile = (m.group(1), m.group(2)) if m else (name, '')
    rows.append({'layer': layer, 'tile': tile, 'launches': int(n), 'total_us': float(tot), 'avg_us': float(avg),
                 'gflop_per_launch': float(tf) * float(avg) * 1e-3, 'tflops': float(tf)})
rows.sort(key=lambda r: -r['total_us'])
json.dump({'label': sys.argv[3] if len(sys.argv) > 3 else '', 'total_ms': sum(r['total_us'] for r in rows) / 1e3,
           'note': 'one profiled planning call (plain launches, every GEMM launch bracketed by HIP events); algorithmic FLOPs',
           'rows': rows}, open(sys.argv[2], 'w'), indent=0)
