#!/usr/bin/env python3
"""Per-kernel time of ONE replayed training step (BASELINE config 4) from a rocprofv3 --kernel-trace CSV directory: the launches
between two consecutive q_sample launches (the step's first kernel), grouped by kernel, with the share that is not this
library's (torch / hipBLASLt / runtime copies) listed separately.
usage: train_timeline.py <dir> [steps from the end; default 3] [--list]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'q_sample' in r['Kernel_Name']]
a, b = idx[-back - 1], idx[-back]
t0 = int(rows[a]['Start_Timestamp'])


def short(n):
    n = n.split('(')[0]
    if 'dua' in n:
        return n.replace('_ZN3dua', '').replace('void dua::', '').replace('dua::', '').split('ILi')[0].split('IDF16_')[0].split('If')[0][:40]
    if n.startswith('Cijk'):
        return 'hipBLASLt ' + n[:24]
    if 'multi_tensor_apply' in n:
        return 'torch multi_tensor ' + ('Adam' if 'Adam' in n else 'unscale' if 'amp_foreach' in n else 'other')
    for key in ('FillFunctor', 'MulFunctor', 'CUDAFunctor_add', 'direct_copy', 'float16_copy', 'sigmoid', 'reduce_kernel', 'CatArray',
                'copyBuffer', 'fillBuffer', 'amp_update_scale'):
        if key in n:
            return 'torch/' + key
    return 'torch/' + n[:60]


agg, cnt = collections.OrderedDict(), collections.Counter()
busy, last_end = 0, t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    k = short(r['Kernel_Name'])
    if '--list' in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f}us dur {(e - s) / 1e3:8.1f}us grid {r['Grid_Size_X']:>8},{r['Grid_Size_Y']},{r['Grid_Size_Z']} {k}")
    agg[k] = agg.get(k, 0) + (e - s) / 1e3
    cnt[k] += 1
    if e > last_end:
        busy += e - max(s, last_end)
        last_end = e
span = (int(rows[b]['Start_Timestamp']) - t0) / 1e3
print(f"step span {span:.1f} us, {b - a} launches, device busy {busy / 1e3:.1f} us")
ours = sum(v for k, v in agg.items() if not (k.startswith('torch') or k.startswith('hipBLASLt')))
print(f"this library's kernels {ours:.1f} us; torch / hipBLASLt / runtime {sum(agg.values()) - ours:.1f} us "
      f"in {sum(c for k, c in cnt.items() if k.startswith('torch') or k.startswith('hipBLASLt'))} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"   {v:9.1f} us {cnt[k]:4d} x  {k}")
