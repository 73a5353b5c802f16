"""Engine clock / package power while the dAC matvec runs back to back (rocm-smi sampled from a side thread): what
clock the fp64-MFMA roofline is actually attainable at under sustained load."""
import os, sys, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mpskit_jl_amd as mk
be = mk.Backend(0)
D, d, W = 1024, 2, 5
H = mk.heisenberg_XXX(0.5, be=be)
r = lambda *s: mk.DTensor(torch.rand(*s, dtype=torch.float64, device=be.device).flatten(), s)
GL, GR, x, y = r(W, D, D), r(W, D, D), r(D, d, D), be.empty(D, d, D)
h = mk.MPO_ddAC(be, H[0], GL, GR)
stop = False
samples = []
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        keep = [l.split(":", 1)[1].strip() for l in out.splitlines() if "sclk" in l or "Power (W)" in l]
        samples.append((time.time(), keep))
        time.sleep(0.5)
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 10.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        h(x, out=y)
    e1.record(); torch.cuda.synchronize()
    n += 1
    print(f"t={time.time()-t0:5.1f}s  {e0.elapsed_time(e1)/200:.4f} ms/matvec", flush=True)
stop = True; th.join()
for t, k in samples:
    print(f"t={t-t0:5.1f}s", k)
