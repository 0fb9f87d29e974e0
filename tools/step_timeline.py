#!/usr/bin/env python3
"""Print the kernel timeline of one replayed step from a rocprofv3 --kernel-trace CSV directory.
usage: step_timeline.py <dir> [index of the step_begin launch to start from; default 100 = inside the default bench's timed
replays (the run ends with the 1000-step loops and the fp32 pass), or the 12th from the end of a short run]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'step_begin' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else (100 if len(idx) > 140 else len(idx) - 12)
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]['Start_Timestamp'])
agg = collections.OrderedDict()
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0]
    name = name.replace('_ZN3dua', '').split('ILi')[0].split('IDF16_')[0][:34]
    print(f"{(s-t0)/1e3:9.1f}us dur {(e-s)/1e3:8.1f}us grid {r['Grid_Size_X']:>8},{r['Grid_Size_Y']},{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']:>3} {name}")
    agg[name] = agg.get(name, 0) + (e - s) / 1e3
print(f"step span {(int(rows[b]['Start_Timestamp'])-t0)/1e3:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"   {v:8.1f} us  {k}")
