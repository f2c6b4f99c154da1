"""Round 3: is the sample kernel's clock set by a power limit?  Runs each variant of the steady-state loop for a few
seconds while a thread samples the GPU's power / clock / temperature (sysfs hwmon + pp_dpm_sclk, else rocm-smi)."""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basebandboard_amd as bbb

N = 1_000_000_000


def find_sources():
    src = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(card + "/hwmon/hwmon*")
        if not hw:
            continue
        h = hw[0]
        for name in ("power1_average", "power1_input", "freq1_input", "temp1_input", "temp2_input", "power1_cap"):
            p = os.path.join(h, name)
            if os.path.exists(p):
                src.setdefault(card, {})[name] = p
        for name in ("pp_dpm_sclk", "gpu_busy_percent", "current_link_speed"):
            p = os.path.join(card, name)
            if os.path.exists(p):
                src.setdefault(card, {})[name] = p
    return src


def read(p):
    try:
        return open(p).read().strip()
    except Exception as e:
        return f"ERR {e}"


SRC = find_sources()
print("sources:", {k: sorted(v) for k, v in SRC.items()}, flush=True)
try:
    print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showmaxpower"], capture_output=True, text=True, timeout=60).stdout[-3000:], flush=True)
except Exception as e:
    print("rocm-smi failed:", e, flush=True)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            row = {}
            for card, d in SRC.items():
                for k in ("power1_average", "power1_input", "freq1_input", "temp1_input"):
                    if k in d:
                        v = read(d[k])
                        if v.lstrip("-").isdigit():
                            row[k] = row.get(k, 0) if False else int(v)
                break            # first card with hwmon = this box's GPU (one GPU visible)
            self.rows.append(row)
            time.sleep(0.02)


def run(tag, body, seconds=4.0):
    s = Sampler(); s.start()
    torch.cuda.synchronize(); t0 = time.perf_counter(); steps = 0
    while time.perf_counter() - t0 < seconds:
        body(steps); steps += 10
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s.stop = True; s.join()
    rows = s.rows[len(s.rows) // 4:]           # drop the ramp
    def avg(k):
        v = [r[k] for r in rows if k in r]
        return sum(v) / len(v) if v else float("nan")
    pk = "power1_average" if any("power1_average" in r for r in rows) else "power1_input"
    print(f"{tag:40s} {dt / steps * 1e3:7.4f} ms/step  power {avg(pk) / 1e6:7.1f} W  sclk {avg('freq1_input') / 1e6:7.1f} MHz  temp {avg('temp1_input') / 1e3:5.1f} C  ({len(rows)} samples)", flush=True)


buf = torch.empty(N, dtype=torch.int8, device="cuda")
for staged, prefetch in ((True, True), (True, False), (False, True), (False, False)):
    u = bbb.LUTOPT.shipped(256); u.set_staged(staged); g = bbb.CLTGRNG(u)
    def body(s0, g=g, prefetch=prefetch):
        for s in range(s0, s0 + 10):
            g.generate(N, first_step=16 + s * N, out=buf)
            if prefetch:
                g.prefetch(N, first_step=16 + (s + 1) * N)
    body(0); torch.cuda.synchronize()
    for rep in range(2):
        run(f"staged={staged} prefetch={prefetch} #{rep}", body)
# a memory-only load for comparison: plain fills
fb = torch.empty(N, dtype=torch.int8, device="cuda")
def fills(s0):
    for _ in range(10):
        fb.fill_(1)
run("torch fill_ 1e9 B", fills)
# idle
s = Sampler(); s.start(); time.sleep(2.0); s.stop = True; s.join()
pk = "power1_average" if any("power1_average" in r for r in s.rows) else "power1_input"
v = [r[pk] for r in s.rows if pk in r]
print("idle power W:", (sum(v) / len(v) / 1e6) if v else None, flush=True)
