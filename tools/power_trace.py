#!/usr/bin/env python3
"""Power + clock trace of the GPU while one kernel family runs back to back (evidence for "power-bound").

usage: power_trace.py <gemm|xprod_f4|xprod_i8|i8t<n>|i8tn<n>|gram|bare_f4|bare_i8|bare_f64|idle> [seconds] [operand pattern of the bare loops: geno|zero|rand]

A sampler thread reads the card's hwmon / gpu_metrics files in sysfs every ~20 ms (falls back to `rocm-smi --showpower --showclocks --json`
every ~0.5 s when sysfs is not readable) while the main thread keeps the GPU busy with ONE kernel family:
  gemm      k_gemm<8,8,3>   : C2 shape 1M x 50k x 32, 'N' + 'T' (fp64 MFMA)
  xprod_f4  k_crossprod_gang: 500k x 100k at reduced rows (200k SNPs x 50k individuals keeps a launch near 100 ms), FP4 MFMA
  xprod_i8  the same on the int8 MFMA
  i8t<n> / i8tn<n> / gram   the int8 streaming kernels of a CG step ('T' on k_gemm_i8, 'N' of a one-copy object on k_gemm_i8_tn), or the whole step
  hbm_plain / hbm_tn   the data movement of those two kernels alone (tools/hbm_pattern_probe: no arithmetic)
  bare_*    tools/mfma_power_probe: a bare stream of that MFMA instruction from registers, no memory traffic (child process) -- the rate the board sustains
Prints: the samples (time, power W, sclk MHz, temperature) thinned to 10 per second, and the summary over the busy window
(mean / p10 / p90 of power and clock, kernel time per launch)."""
import ctypes, glob, json, os, subprocess, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miraculix_amd as mx
from bench import synth_genotypes_device

target = sys.argv[1]
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
dev = torch.device("cuda", 0)
L = mx.load_shared_library()


def find_hwmon():
    """hwmon directory of the HIP device 0 (matched by PCI bus id), or None"""
    try:
        bus = torch.cuda.get_device_properties(0).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(0), "pci_domain_id", 0)
        devn = torch.cuda.get_device_properties(0).pci_device_id
        want = f"{dom:04x}:{bus:02x}:{devn:02x}.0"
    except Exception:
        want = None
    cands = []
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        devdir = os.path.realpath(os.path.join(card, "device"))
        hw = glob.glob(os.path.join(devdir, "hwmon", "hwmon*"))
        if hw:
            cands.append((os.path.basename(devdir), hw[0]))
    for name, hw in cands:
        if want and name.lower() == want.lower():
            return hw, name
    return (cands[0][1], cands[0][0] + " (first card: bus id not matched)") if len(cands) == 1 else (None, f"no match for {want} among {[c[0] for c in cands]}")


def read_num(path):
    try:
        return float(open(path).read().split()[0])
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.hw, self.card = find_hwmon()
        self.rows, self.stop = [], False
        self.mode = "sysfs" if self.hw and any(read_num(os.path.join(self.hw, f)) is not None for f in ("power1_input", "power1_average")) else "rocm-smi"

    def sample_sysfs(self):
        p = read_num(os.path.join(self.hw, "power1_input"))
        if p is None:
            p = read_num(os.path.join(self.hw, "power1_average"))
        f = read_num(os.path.join(self.hw, "freq1_input"))
        t = read_num(os.path.join(self.hw, "temp1_input"))
        return (p * 1e-6 if p is not None else None, f * 1e-6 if f is not None else None, t * 1e-3 if t is not None else None)

    def sample_smi(self):
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
            j = json.loads(r.stdout)
            c = next(iter(j.values()))
            p = next((float(v) for k, v in c.items() if "Power" in k and "(W)" in k), None)
            s = next((float(str(v).strip("()Mhz ")) for k, v in c.items() if k.startswith("sclk clock speed")), None)
            t = next((float(v) for k, v in c.items() if "Temperature" in k and "junction" in k.lower()), None)
            return p, s, t
        except Exception:
            return None, None, None

    def run(self):
        t0 = time.perf_counter()
        while not self.stop:
            s = self.sample_sysfs() if self.mode == "sysfs" else self.sample_smi()
            self.rows.append((time.perf_counter() - t0,) + s)
            time.sleep(0.02 if self.mode == "sysfs" else 0.05)


def stage():
    if target == "gemm":
        snps, indiv, n = 1_000_000, 50_000, 32
        plink = synth_genotypes_device(torch, snps, indiv, 42, dev)
        f = mx.read_plink.calc_freq(plink, snps, indiv)
        dg = mx.dgemm_compressed
        dg.set_options(use_gpu=True, not_center=True, verbose=0)
        obj = dg.init_compressed(plink, None, snps, indiv, f, n)
        del plink
        g = torch.Generator(device=dev); g.manual_seed(1)
        BN = torch.randn((n, snps), dtype=torch.float64, device=dev, generator=g).t()
        BT = torch.randn((n, indiv), dtype=torch.float64, device=dev, generator=g).t()
        CN = torch.zeros((n, indiv), dtype=torch.float64, device=dev).t()
        CT = torch.zeros((n, snps), dtype=torch.float64, device=dev).t()

        def step():
            dg.dgemm_compressed_main(False, obj, BN, snps, indiv, out=CN)
            dg.dgemm_compressed_main(True, obj, BT, snps, indiv, out=CT)
        return step, 2, f"k_gemm 1M x 50k x 32 'N' + 'T' (3.2 TFLOP per launch)"
    if target.startswith("i8tn") or target.startswith("i8t") or target == "gram":
        # i8tn<n>: the 'N' product of a one-copy object on k_gemm_i8_tn (n = 1: one digit tile per pass, n = 4 .. 6: two); i8t<n>: the 'T' product on the plain
        # int8 kernel; gram: one CG step (mxa_gram_matvec, n = 1) = both.  Shape: TN_SNPS x TN_INDIV (default: the config-5 shard)
        snps, indiv = int(os.environ.get("TN_SNPS", 250_000)), int(os.environ.get("TN_INDIV", 100_000))
        n = 1 if target == "gram" else int(target[4:] or 1) if target.startswith("i8tn") else int(target[3:] or 1)
        trans = target.startswith("i8t") and not target.startswith("i8tn")
        plink = synth_genotypes_device(torch, snps, indiv, 42, dev)
        f = mx.read_plink.calc_freq(plink, snps, indiv)
        dg = mx.dgemm_compressed
        dg.set_options(use_gpu=True, not_center=True, verbose=0)
        obj = dg.init_compressed(plink, None, snps, indiv, f, n)
        del plink
        g = torch.Generator(device=dev); g.manual_seed(1)
        kk, mm = (indiv, snps) if trans else (snps, indiv)
        if target == "gram":
            kk = mm = indiv
        BN = torch.randn((n, kk), dtype=torch.float64, device=dev, generator=g).t()
        CN = torch.zeros((n, mm), dtype=torch.float64, device=dev).t()

        def step():
            if target == "gram":
                dg.gram_matvec(obj, BN, snps, indiv, out=CN, sync=False)
            else:
                dg.dgemm_compressed_main(trans, obj, BN, snps, indiv, out=CN)
        what = "one CG step mxa_gram_matvec ('T' on k_gemm_i8 + 'N' on k_gemm_i8_tn)" if target == "gram" else ("k_gemm_i8 'T'" if trans else "k_gemm_i8_tn 'N'")
        return step, (2 if target == "gram" else 1), f"{what} {snps} x {indiv} x {n} (one packed copy)"
    if target.startswith("xprod"):
        snps, indiv = int(os.environ.get("XP_SNPS", 200_000)), int(os.environ.get("XP_INDIV", 50_000))
        if target == "xprod_i8":
            os.environ["MXA_XPROD_ENGINE"] = "i8"
        X = synth_genotypes_device(torch, indiv, snps, 46, dev, p_along="cols")
        M = torch.empty((indiv, indiv), dtype=torch.float64, device=dev)

        def step():
            mx.crossproduct.snp_crossprod(X, snps, indiv, is_snpmajor=False, is_plink_format=True, out=M)
        return step, 1, f"k_crossprod_gang ({'int8' if target == 'xprod_i8' else 'FP4'} MFMA) {snps} SNPs x {indiv} indiv"
    if target == "idle":
        return (lambda: time.sleep(0.05)), 0, "idle"
    raise SystemExit(__doc__)


def power_cap(hw):
    return {k: (read_num(os.path.join(hw, k)) or 0) * 1e-6 for k in ("power1_cap", "power1_cap_max", "power1_cap_default") if hw and os.path.exists(os.path.join(hw, k))}


if target.startswith("bare_") or target.startswith("hbm_"):
    # hbm_plain / hbm_tn: the DATA MOVEMENT of the two int8 kernels of a CG step alone (tools/hbm_pattern_probe, patterns 2 and 4: LDS DMA of the packed tiles +
    # their digit traffic, no arithmetic), ~1 ms launches back to back for `seconds`
    hbm = target.startswith("hbm_")
    exe = os.path.join(ROOT, "tools", "hbm_pattern_probe" if hbm else "mfma_power_probe")
    if hbm and not os.path.exists(exe):
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-o", exe, exe + ".hip"])
    pat = "-" if hbm else sys.argv[3] if len(sys.argv) > 3 else "geno"
    argv = [exe, "250000", "100000", str(int(seconds * 1050)), "2" if target == "hbm_plain" else "4"] if hbm else [exe, target[5:], str(seconds), "1", pat]
    smp = Sampler()
    smp.start()
    time.sleep(1.0)
    t0 = time.perf_counter()
    r = subprocess.run(argv, capture_output=True, text=True, timeout=seconds + 120)
    t1 = time.perf_counter()
    time.sleep(1.0)
    smp.stop = True
    smp.join()
    b0, b1 = 1.0 + 1.0, 1.0 + (t1 - t0) - 0.3          # (the child needs a moment to come up)
    busy = [x for x in smp.rows if b0 <= x[0] <= b1]

    def st(col):
        v = sorted(x[col] for x in busy if x[col] is not None)
        return None if not v else {"mean": round(sum(v) / len(v), 1), "p10": round(v[len(v) // 10], 1), "p50": round(v[len(v) // 2], 1), "p90": round(v[(9 * len(v)) // 10], 1), "max": round(v[-1], 1), "samples": len(v)}
    print(f"# power_trace {target} ({pat} operands): sampler {smp.mode} ({smp.card}); power cap {power_cap(smp.hw)}")
    print(r.stdout + r.stderr[-500:])
    last = -1.0
    for x in smp.rows:
        if x[0] - last >= 0.25:
            last = x[0]
            print(f"t={x[0]:6.2f}s power={x[1] if x[1] is None else round(x[1], 1)} W sclk={x[2] if x[2] is None else round(x[2])} MHz")
    mean = [ln for ln in r.stdout.splitlines() if ln.startswith("mean over") or " best " in ln]
    print(json.dumps({"target": target, "operands": pat, "sampler": smp.mode, "power_cap_W": power_cap(smp.hw), "busy_window_s": [b0, round(b1, 2)], "power_W": st(1), "sclk_MHz": st(2),
                      "rate": mean[0] if mean else None}))
    sys.exit(r.returncode)

step, launches_per_step, what = stage()
step(); torch.cuda.synchronize()
smp = Sampler()
smp.start()
time.sleep(1.0)                                   # one second of idle samples first
t_busy0 = time.perf_counter()
L.mxa_profile_reset()
nsteps = 0
kernel_ms = []
while time.perf_counter() - t_busy0 < seconds:
    step()
    nsteps += 1
    if nsteps % 4 == 0:
        torch.cuda.synchronize()
        la, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        L.mxa_profile_get(ctypes.byref(la), ctypes.byref(ms))
        if la.value:
            kernel_ms.append((time.perf_counter() - t_busy0, ms.value / la.value))
        L.mxa_profile_reset()
torch.cuda.synchronize()
t_busy1 = time.perf_counter()
time.sleep(1.0)
smp.stop = True
smp.join()
b0 = 1.0 + 0.5                                    # skip the ramp: the busy window starts half a second in
b1 = 1.0 + (t_busy1 - t_busy0)
busy = [r for r in smp.rows if b0 <= r[0] <= b1]


def stats(col):
    v = sorted(r[col] for r in busy if r[col] is not None)
    if not v:
        return None
    return {"mean": round(sum(v) / len(v), 1), "p10": round(v[len(v) // 10], 1), "p50": round(v[len(v) // 2], 1), "p90": round(v[(9 * len(v)) // 10], 1), "max": round(v[-1], 1), "samples": len(v)}


print(f"# power_trace {target}: {what}; sampler {smp.mode} ({smp.card}); busy {t_busy1 - t_busy0:.1f} s, {nsteps} steps")
last = -1.0
for r in smp.rows:
    if r[0] - last >= 0.1:
        last = r[0]
        print(f"t={r[0]:6.2f}s power={r[1] if r[1] is None else round(r[1], 1)} W sclk={r[2] if r[2] is None else round(r[2])} MHz temp={r[3]}")
print("# kernel ms per launch over time:", " ".join(f"{t:.1f}s:{m:.2f}" for t, m in kernel_ms[:: max(1, len(kernel_ms) // 12)]))
print(json.dumps({"target": target, "what": what, "sampler": smp.mode, "power_cap_W": power_cap(smp.hw), "busy_window_s": [b0, round(b1, 2)], "power_W": stats(1), "sclk_MHz": stats(2), "temp_C": stats(3),
                  "kernel_ms_per_launch_mean": round(sum(m for _, m in kernel_ms) / max(1, len(kernel_ms)), 3)}))
