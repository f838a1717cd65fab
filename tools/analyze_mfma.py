"""MFMA-pipe utilisation per kernel family from a rocprofv3 --pmc run (csv output):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
              --output-format csv -d DIR -- python3 <program>
    python tools/analyze_mfma.py DIR [label]  -> JSON on stdout

Two readings per family (sums over its dispatches):
  mfma_issue_frac = SQ_INSTS_VALU_MFMA_F32 x 32 cycles / (GRBM_GUI_ACTIVE_per_XCD x 1024 SIMDs): the share of the chip's
                    matrix-pipe cycles that held a v_mfma_f32_16x16x4_f32 (8 passes = 32 cycles each) while the kernel ran.
                    GRBM_GUI_ACTIVE is reported summed over the 8 XCDs, hence / 8.
  mfma_busy_over_sq_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (both as reported: summed over SQs; on gfx950
                    SQ_VALU_MFMA_BUSY_CYCLES came out as exactly 32 x SQ_INSTS_VALU_MFMA_F32)."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
fam = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    for key in ('conv_gemm_kernel', 'narrow_conv_kernel', 'resident_chain_kernel', 'wgrad_kernel', 'gnb_kernel', 'wreduce_kernel'):
        if key in k:
            k = key
            break
    else:
        continue
    fam[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in disp[k]:
        dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    disp[k].add(r['Dispatch_Id'])
out = {'source': d, 'label': sys.argv[2] if len(sys.argv) > 2 else ''}
for k, c in fam.items():
    gui = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    out[k] = {'dispatches': len(disp[k]), 'counters': dict(c),
              'mfma_issue_frac': (c.get('SQ_INSTS_VALU_MFMA_F32', 0.0) * 32.0 / (gui * 1024.0)) if gui else None,
              'mfma_busy_over_sq_busy': (c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / c['SQ_BUSY_CYCLES']) if c.get('SQ_BUSY_CYCLES') else None,
              # the same with the dispatch's own timestamps as the clock (2.4 GHz, 1024 SIMDs): robust for short kernels,
              # whose GRBM window also counts the profiler's start / stop
              'mfma_issue_frac_by_timestamps': c.get('SQ_INSTS_VALU_MFMA_F32', 0.0) * 32.0 / (dur[k] * 2.4 * 1024.0) if dur[k] else None,
              'kernel_time_ms': dur[k] / 1e6,
              'mfma_insts_per_wave': c.get('SQ_INSTS_VALU_MFMA_F32', 0.0) / max(c.get('SQ_WAVES', 1.0), 1.0)}
print(json.dumps(out, indent=1))
