#!/usr/bin/env python3
"""Is the conv kernel power-limited?  Runs one layer shape back to back for a few seconds and samples the board power,
clocks and temperature (rocm-smi / amd-smi, whichever answers) from a side thread.
usage: python tools/power_probe.py [shape index of tools/bench_conv.SHAPES] [seconds]"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_unet_amos_amd import ops
from tools.bench_conv import SHAPES

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
S, cin, cout, fused = SHAPES[idx]
dev = "cuda"
dt = torch.float16
x = torch.randn(1, S, S, S, cin, device=dev).to(dt)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) / (27 * cin) ** 0.5
wp, bp = ops.pack_conv3_weights(w, torch.zeros(cout, device=dev), dt)
y = torch.empty(1, S, S, S, cout, device=dev, dtype=dt)
stats = ops.stats_buffer(1, cout, dev)
samples = []
stop = False


def sample():
    cmds = [["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showmaxpower"], ["amd-smi", "metric", "-p", "-c"]]
    while not stop:
        for c in cmds:
            try:
                out = subprocess.run(c, capture_output=True, text=True, timeout=5).stdout
                if out.strip():
                    samples.append((time.time(), out))
                    break
            except Exception as e:       # noqa: BLE001
                samples.append((time.time(), f"{c[0]}: {e}"))
        time.sleep(0.3)


th = threading.Thread(target=sample)
th.start()
time.sleep(1.0)
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(200):
        ops.conv3d_k3(x, cin, 0, wp, bp, cout, y, 0, stats)
    torch.cuda.synchronize()
    n += 200
t1 = time.time()
time.sleep(1.0)
stop = True
th.join()
fl = 2.0 * cin * cout * 27 * S ** 3
print(f"{S}^3 {cin}->{cout}: {n} launches in {t1 - t0:.2f} s = {(t1 - t0) / n * 1e6:.1f} us each, {fl * n / (t1 - t0) / 1e12:.0f} TF/s; loop from {t0:.1f} to {t1:.1f}")
for ts, out in samples:
    keep = [l.strip() for l in out.splitlines() if any(k in l.lower() for k in ("power", "sclk", "mclk", "temperature (sensor junction)", "gfx", "socket"))]
    print(f"t={ts - t0:+.1f}s | " + " | ".join(keep)[:600])
