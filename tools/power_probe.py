"""Samples the GPU's power and shader clock (hwmon of the amdgpu device, readable without root) while a bench
workload runs: is a launch-bound loop also power-bound?   python tools/power_probe.py [bench args...]
(development aid; run with gpurun)"""
import glob, os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def rd(p):
    try:
        return open(p).read().strip()
    except Exception:
        return None
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
print("hwmon dirs:", hw)
for h in hw:
    for f in sorted(os.listdir(h)):
        if f.startswith(("power", "freq", "temp1_input", "name")):
            print(" ", h, f, rd(os.path.join(h, f)))
args = sys.argv[1:] or ["--steps", "4000", "--warmup", "50", "--no-cpu"]
p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, text=True)
samples = []
t0 = time.time()
while p.poll() is None:
    row = {"t": round(time.time() - t0, 2)}
    for h in hw:
        for f in ("power1_average", "power1_input", "freq1_input", "freq2_input"):
            v = rd(os.path.join(h, f))
            if v is not None:
                row[f] = int(v)
    samples.append(row)
    time.sleep(0.1)
out = p.stdout.read()
for s in samples[::5]:
    print(s)
try:
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print("ms_per_step", d["ms_per_step"], d["kernels_ms"])
except Exception as e:
    print("bench output:", out[-500:], e)
subprocess.call(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"])
