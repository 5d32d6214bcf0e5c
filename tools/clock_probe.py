"""Shader clock and board power while the headline step runs in a loop (sysfs hwmon of the first AMD GPU, polled every
10 ms from a thread), next to the step time: do the boxes of the pool differ in what the card sustains under this load?
Usage: python tools/clock_probe.py [seconds]"""
import glob, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine, FgbPlan

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
N, G = 1_000_000_000, 1 << 20
eng = Engine(0)
p, k, v = eng.alloc(N * 4), eng.alloc(N * 4), eng.alloc(N * 4)
eng.gen_columns(0x4861726B4442, 0, N, G, False, p, k, v)
plan = FgbPlan(eng, N, G, timing=1)
sums, counts = eng.alloc(G * 4), eng.alloc(G * 8)


def pci_dir():
    """sysfs directory of THE device this process computes on (a box shows the other cards of its host, too)."""
    import ctypes
    hip = ctypes.CDLL(None)                     # the HIP runtime harkdb_amd._ffi loaded with RTLD_GLOBAL
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, 0) != 0:
        return None
    return "/sys/bus/pci/devices/" + buf.value.decode().lower()


PCI = pci_dir()
print("device:", PCI, flush=True)


def hw(pattern):
    f = sorted(glob.glob((PCI or "/sys/class/drm/card0/device") + "/hwmon/hwmon*/" + pattern))
    return f[0] if f else None


f_clk, f_pow, f_tmp = hw("freq1_input"), hw("power1_average") or hw("power1_input"), hw("temp2_input")
print("sensors:", f_clk, f_pow, f_tmp, flush=True)


def show_static(tag):
    for name in ("current_compute_partition", "current_memory_partition", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "pp_dpm_sclk", "mem_info_vram_total", "pcie_bw", "gpu_busy_percent", "mem_busy_percent"):
        try:
            txt = open(os.path.join(PCI, name)).read().strip().replace("\n", " | ")
            print(f"  [{tag}] {name}: {txt}", flush=True)
        except Exception as e:
            print(f"  [{tag}] {name}: unreadable ({type(e).__name__})", flush=True)
    f2 = hw("freq2_input")
    if f2:
        print(f"  [{tag}] freq2 ({open(f2.replace('_input', '_label')).read().strip()}): {float(open(f2).read()) / 1e6:.0f} MHz", flush=True)


show_static("idle")
samples, stop = [], False


def poll():
    while not stop:
        row = [time.perf_counter()]
        for f in (f_clk, f_pow, f_tmp):
            try:
                row.append(float(open(f).read()))
            except Exception:
                row.append(float("nan"))
        samples.append(row)
        time.sleep(0.01)


def step():
    plan.reset(); plan.run(p, ">", 0.5, k, v, N); plan.finish(sums, counts, check=False)


for _ in range(3):
    step()
eng.sync()
idle = [float(open(f).read()) if f else float("nan") for f in (f_clk, f_pow, f_tmp)]
th = threading.Thread(target=poll); th.start()
t0 = time.perf_counter(); steps = 0
while time.perf_counter() - t0 < secs:
    for _ in range(10):
        step()
    eng.sync(); steps += 10
dt = time.perf_counter() - t0
for _ in range(20):
    step()
show_static("under load")
eng.sync()
stop = True; th.join()
a = np.array(samples)
ms, cnt = plan.timing()
print(f"{steps} steps, {dt / steps * 1e3:.3f} ms/step wall; kernels per step: " + ", ".join(f"{kk} {ms[kk] / max(cnt[kk], 1):.3f} ms" for kk in ms))
print(f"before the loop: sclk {idle[0] / 1e6:.0f} MHz, power {idle[1] / 1e6:.0f} W, temp {idle[2] / 1e3:.0f} C")
half = a[len(a) // 2:]
print(f"under load (second half, {len(half)} samples): sclk mean {np.nanmean(half[:, 1]) / 1e6:.0f} min {np.nanmin(half[:, 1]) / 1e6:.0f} max {np.nanmax(half[:, 1]) / 1e6:.0f} MHz; "
      f"power mean {np.nanmean(half[:, 2]) / 1e6:.0f} max {np.nanmax(half[:, 2]) / 1e6:.0f} W; temp {np.nanmean(half[:, 3]) / 1e3:.0f} C")
