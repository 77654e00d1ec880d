"""Sample the GPU's power draw and clocks (rocm-smi) while a bench.py workload runs, in one process tree (GPU diagnostic).

    python tools/power_trace.py [bench.py args...]  ->  samples on stdout, then the bench line

The child `python bench.py ...` runs the workload; this parent never touches the GPU (it only shells out to rocm-smi every 0.5 s):
what clock does the chip grant the guided step, and at what power?  DESIGN.md section 3 reads the 1.74 GHz under the dominant
kernel from in-kernel s_memtime / s_memrealtime stamps; this is the same fact from the driver's side."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:], stdout=subprocess.PIPE, text=True)
t0 = time.time()
rows = []
while child.poll() is None:
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d.get("card0") or next(iter(d.values()))
        keep = {k: v for k, v in card.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "fclk", "gpu use"))}
        rows.append((round(time.time() - t0, 2), keep))
    except Exception as e:  # noqa: BLE001
        rows.append((round(time.time() - t0, 2), {"error": str(e)[:100]}))
    time.sleep(0.5)
line = child.stdout.read()
for t, k in rows:
    print(t, json.dumps(k))
print(line.strip().splitlines()[-1][:600] if line.strip() else "(no bench line)")
sys.exit(child.returncode)
