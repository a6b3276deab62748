import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'trace' in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
m = {k: sum(v) / len(v) for k, v in agg.items()}
us = sum(dur) / max(len(dur), 1)
print(f"[{tag}] kernel avg {us:.1f} us (profiled), VGPR {r.get('VGPR_Count')}")
for k in sorted(m):
    print(f"   {k:30s} {m[k]:.4g}")
g = m.get
if g('SQ_ACTIVE_INST_VALU') and g('SQ_THREAD_CYCLES_VALU'):
    print(f"   lane utilisation   = {g('SQ_THREAD_CYCLES_VALU') / (g('SQ_ACTIVE_INST_VALU') * 64) * 100:.1f} %")
if g('SQ_WAVE_CYCLES') and g('SQ_WAVES'):
    cyc = us * 1e-6 * 2.4e9
    print(f"   avg waves per SIMD = {g('SQ_WAVE_CYCLES') * 4 / cyc / 1024:.2f}  (wave lifetime {g('SQ_WAVE_CYCLES') * 4 / g('SQ_WAVES') / 2400:.1f} us)")
    print(f"   VALU issue share   = {g('SQ_INSTS_VALU') * 2 / (cyc * 1024) * 100:.1f} % of SIMD cycles (2 cyc / wave64 VALU)")
    print(f"   VALU per wave {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f}, SALU {g('SQ_INSTS_SALU') / g('SQ_WAVES'):.0f}, LDS {g('SQ_INSTS_LDS') / g('SQ_WAVES'):.0f}")
if g('SQ_WAIT_ANY'):
    tot = g('SQ_WAIT_ANY') + g('SQ_WAIT_INST_ANY') + g('SQ_ACTIVE_INST_ANY')
    print(f"   wave time: waitcnt {g('SQ_WAIT_ANY') / tot * 100:.0f} %, issue stall {g('SQ_WAIT_INST_ANY') / tot * 100:.0f} %, issuing {g('SQ_ACTIVE_INST_ANY') / tot * 100:.0f} %")
if g('TCC_HIT_sum') is not None:
    print(f"   L2 hit rate {g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1) * 100:.1f} %")
