#!/usr/bin/env python3
"""Board power and clocks while the demo step (or the GEMM micro-benchmark's kernels) runs back to back: is the step at the board's power cap?
Reads only (rocm-smi); changes no setting.  Phases: idle, the demo step for ~6 s, a bandwidth-bound loop (the instance norm alone) for ~3 s."""
import glob, os, re, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import Generator, mean_variance_norm, synthetic, synthetic_state_dict

def read_hwmon():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input"):
            f = os.path.join(h, name)
            if os.path.exists(f):
                try: out[name] = int(open(f).read().strip())
                except (OSError, ValueError): pass
        if out: break
    return out

def smi_values():
    """{'power_w': ..., 'cap_w': ..., 'sclk_mhz': ...} of the first GPU rocm-smi lists (the one this container sees)."""
    try:
        t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:      # noqa: BLE001
        return {"error": str(e)}
    out = {}
    for l in t.splitlines():
        m = re.search(r"GPU\[0\].*Power \(W\):\s*([\d.]+)", l)
        if m:
            out["cap_w" if "Max" in l else "power_w"] = float(m.group(1))
        m = re.search(r"GPU\[0\].*sclk clock level.*\((\d+)Mhz\)", l)
        if m: out["sclk_mhz"] = float(m.group(1))
    return out

def smi():
    return str(smi_values())

samples, stop = [], False
def sampler():
    while not stop:
        h = smi_values(); h["t"] = time.perf_counter(); samples.append(h); time.sleep(0.05)

dev = torch.device("cuda:0")
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
src = torch.from_numpy(synthetic.pose_windows(1777, 585, 22)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(4242, 585, 22)).to(dev)
mean, std = synthetic.cnt_norm(7)
for _ in range(3): model.characterize_pair(src, cha, mean, std)
torch.cuda.synchronize()
print("hwmon at rest:", read_hwmon()); print("rocm-smi at rest:", smi())
th = threading.Thread(target=sampler, daemon=True); th.start()
marks = [("idle", time.perf_counter())]
time.sleep(1.0)
marks.append(("demo step", time.perf_counter()))
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(20): model.characterize_pair(src, cha, mean, std)
    torch.cuda.synchronize(); n += 20
step_ms = (time.perf_counter() - t0) / n * 1e3
mid = smi()
marks.append(("instance norm only (HBM-bound)", time.perf_counter()))
x = torch.randn((1170, 90, 256), device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    for _ in range(200): mean_variance_norm(x.permute(0, 2, 1))
    torch.cuda.synchronize()
marks.append(("end", time.perf_counter()))
stop = True; th.join()
print(f"demo step sustained: {step_ms:.3f} ms = {585 / step_ms:.1f} k frames/s"); print("rocm-smi during the demo loop:", mid)
for (name, a), (_, b) in zip(marks, marks[1:]):
    ss = [s for s in samples if a + 0.3 <= s["t"] <= b]
    if not ss: continue
    def avg(k): 
        v = [s[k] for s in ss if k in s]; return sum(v) / len(v) if v else float("nan")
    pw = [s["power_w"] for s in ss if "power_w" in s]
    print(f"{name:32s}: {len(ss):3d} samples  power {avg('power_w'):7.1f} W (max {max(pw) if pw else float('nan'):.0f}; cap {avg('cap_w'):.0f} W)  sclk {avg('sclk_mhz'):7.0f} MHz")
