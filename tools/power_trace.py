#!/usr/bin/env python3
"""Socket power / shader clock trace of a sampling loop (round 5; the review asked for an amd-smi corroboration of
`profiles/r04_clock_probe.md`, whose 1.77-GHz-under-full-load figure came from s_memtime alone).

    python tools/power_trace.py [--hz 20] [--seconds 6] [--out gpurun_out/r05_c5_power_clock.md] WORKLOAD[:flags] ...

WORKLOAD: c5 (1024 x 64-atom graphs), dNNN (NNN x 64-atom graphs), c2, ens8, gNNN (tools/ab_step.py's names).  For
every workload a CHILD process runs the sampling loop for `seconds` (this process never initialises the GPU); a
sampler thread here polls the SMU metrics table through the amdsmi Python binding (amdsmi_get_gpu_metrics_info:
current_gfxclk and the per-XCD current_gfxclks, socket power, hotspot temperature, throttle status) and, where the
binding is missing, the hwmon files under /sys/class/drm.  Output: one markdown table per workload (idle lead-in,
loaded plateau: min / median / max of clock and power; the ms/step the child measured) + the raw samples as CSV
next to it."""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(workload, seconds, flags):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from tsdiff_amd import engine, synth
    if "fused0" in flags:
        engine.OPTIONS.fused_encoder = False
    if "fused1" in flags:
        engine.OPTIONS.fused_encoder = "force"
    from bench import SamplingRun, make_models, to_dev
    from tsdiff_amd.sampler import EnsembleSampler
    dev = torch.device("cuda:0")
    cfg = synth.DEFAULT_MODEL_CONFIG
    M = int(workload[3:]) if workload.startswith("ens") else 1
    models = make_models(cfg, range(M), dev)
    if workload == "c5" or workload[0] == "d":
        G = 1024 if workload == "c5" else int(workload[1:])
        b = synth.dense_stress_batch(G, n=64, seed=1000)
    else:
        G = int(workload[1:]) if workload[0] == "g" else 100
        b = synth.wb97xd3_like_batch(G, seed=1000)
    g = to_dev(b, dev)
    N = g["pos"].shape[0]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    dense = workload == "c5" or workload[0] == "d"
    pos = g["pos"].clone() if dense else torch.randn(N, 3, device=dev, generator=gen) * 1.5
    run = SamplingRun(EnsembleSampler(models), g, G, pos, True, 1234)
    run.run(3)
    torch.cuda.synchronize()
    steps = 8 if dense else 200
    print(f"LOOP_START {time.time():.6f}", flush=True)
    t0 = time.perf_counter()
    ts = []
    while time.perf_counter() - t0 < seconds:
        dt, p = run.timed(steps)
        ts.append(dt / steps * 1e3)
    print(f"LOOP_END {time.time():.6f}", flush=True)
    assert torch.isfinite(p).all()
    print(f"RESULT {min(ts):.4f} {float(np.median(ts)):.4f} {len(ts) * steps} {N}", flush=True)


class Sampler:
    """polls the first GPU's metrics at `hz`; .rows = [(t, gfxclk_MHz, [per-XCD MHz], power_W, temp_C, throttle)]"""

    def __init__(self, hz):
        self.dt, self.rows, self.stop, self.src = 1.0 / hz, [], False, None
        self.h = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.smi = amdsmi
            self.h = amdsmi.amdsmi_get_processor_handles()[0]
            self.src = "amdsmi_get_gpu_metrics_info"
        except Exception as e:  # noqa: BLE001
            self.smi, self.err = None, repr(e)
            self.hw = self._find_hwmon()
            self.src = f"sysfs hwmon ({self.hw})" if self.hw else None

    @staticmethod
    def _find_hwmon():
        import glob
        for p in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if os.path.exists(os.path.join(p, "freq1_input")):
                return p
        return None

    def one(self):
        t = time.time()
        if self.smi is not None:
            m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
            def num(v):
                return None if v in (None, "N/A") or (isinstance(v, int) and v >= 0xffff) else v
            clks = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, int) and 0 < c < 0xffff]
            clk = num(m.get("current_gfxclk")) or (sum(clks) / len(clks) if clks else None)
            pw = num(m.get("current_socket_power")) or num(m.get("average_socket_power"))
            return (t, clk, clks, pw, num(m.get("temperature_hotspot")), m.get("throttle_status"),
                    num(m.get("ppt_residency_acc")), num(m.get("accumulation_counter")), num(m.get("socket_thm_residency_acc")),
                    num(m.get("energy_accumulator")))
        if self.hw:
            def rd(n, sc):
                try:
                    return int(open(os.path.join(self.hw, n)).read()) / sc
                except Exception:  # noqa: BLE001
                    return None
            return (t, rd("freq1_input", 1e6), [], rd("power1_average", 1e6) or rd("power1_input", 1e6),
                    rd("temp2_input", 1e3), None, None, None, None, None)
        return (t, None, [], None, None, None, None, None, None, None)

    def loop(self):
        while not self.stop:
            t0 = time.time()
            try:
                self.rows.append(self.one())
            except Exception as e:  # noqa: BLE001
                self.rows.append((t0, None, [], None, None, repr(e), None, None, None, None))
            time.sleep(max(0.0, self.dt - (time.time() - t0)))


def med(v):
    v = sorted(x for x in v if x is not None)
    return v[len(v) // 2] if v else None


def summarize(rows, t0, t1):
    idle = [r for r in rows if r[0] < t0 - 0.2]
    load = [r for r in rows if t0 + 0.5 <= r[0] <= t1 - 0.1]  # (plateau: skip the ramp)
    def stat(rs, i):
        v = [r[i] for r in rs if r[i] is not None]
        return (min(v), med(v), max(v)) if v else (None, None, None)
    xcd = [min(r[2]) for r in load if r[2]], [max(r[2]) for r in load if r[2]]
    # residency accumulators of the SMU over the loaded window: the fraction of its sampling ticks spent at the package
    # power limit (ppt) / the thermal limit
    ppt = thm = None
    acc = [r for r in load if r[6] is not None and r[7] is not None]
    if len(acc) >= 2 and acc[-1][7] > acc[0][7]:
        ppt = (acc[-1][6] - acc[0][6]) / (acc[-1][7] - acc[0][7])
        if acc[0][8] is not None:
            thm = (acc[-1][8] - acc[0][8]) / (acc[-1][7] - acc[0][7])
    return {"ppt_frac": ppt, "thm_frac": thm,"idle_clk": stat(idle, 1), "idle_pw": stat(idle, 3), "load_clk": stat(load, 1), "load_pw": stat(load, 3),
            "load_temp": stat(load, 4), "n_load": len(load), "xcd_min_med": med(xcd[0]), "xcd_max_med": med(xcd[1]),
            "throttle": sorted({str(r[5]) for r in load})}


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(args[1], float(args[2]), args[3].split(","))
        return
    hz, seconds, out, wl = 20.0, 6.0, os.path.join(ROOT, "gpurun_out", "r05_c5_power_clock.md"), []
    i = 0
    while i < len(args):
        if args[i] == "--hz":
            hz = float(args[i + 1]); i += 2
        elif args[i] == "--seconds":
            seconds = float(args[i + 1]); i += 2
        elif args[i] == "--out":
            out = args[i + 1]; i += 2
        else:
            wl.append(args[i]); i += 1
    wl = wl or ["c5", "d128"]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    s = Sampler(hz)
    th = threading.Thread(target=s.loop, daemon=True)
    th.start()
    lines = ["# Socket power and shader clock during sampling loops (tools/power_trace.py)", "",
             f"Source: {s.src or 'NONE AVAILABLE: ' + getattr(s, 'err', '')}; {hz:g} Hz; every workload is a child process "
             f"that runs `dynamic_sampling` calls back to back for {seconds:g} s.  Loaded statistics skip the first 0.5 s.",
             "", "| workload | ms/step (min / median) | idle clock MHz | loaded clock MHz (min / median / max) | per-XCD clock "
             "(median of min / of max) | loaded power W (min / median / max) | hotspot C | at power limit (ppt residency) | at thermal limit | throttle | samples |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    csv = ["workload,t_rel_s,gfxclk_mhz,xcd_min_mhz,xcd_max_mhz,power_w,temp_c,throttle"]
    for spec in wl:
        name, *flags = spec.split(":")
        time.sleep(1.5)  # idle lead-in
        n0 = len(s.rows)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, str(seconds), ",".join(flags) or "-"],
                           capture_output=True, text=True)
        o = p.stdout
        try:
            t0 = float([l for l in o.splitlines() if l.startswith("LOOP_START")][0].split()[1])
            t1 = float([l for l in o.splitlines() if l.startswith("LOOP_END")][0].split()[1])
            res = [l for l in o.splitlines() if l.startswith("RESULT")][0].split()[1:]
        except Exception:  # noqa: BLE001
            lines.append(f"| {spec} | FAILED | | | | | | | | | |")
            print(spec, "FAILED", p.stdout[-1500:], p.stderr[-3000:])
            continue
        rows = s.rows[n0:]
        st = summarize(rows, t0, t1)
        def f3(t, fmt="{:.0f}"):
            return " / ".join("-" if x is None else fmt.format(x) for x in t)
        lines.append(f"| {spec} | {res[0]} / {res[1]} | {f3(st['idle_clk'][1:2])} | {f3(st['load_clk'])} | "
                     f"{st['xcd_min_med']} / {st['xcd_max_med']} | {f3(st['load_pw'])} | {f3(st['load_temp'][1:2])} | "
                     f"{'-' if st['ppt_frac'] is None else format(st['ppt_frac'], '.2f')} | "
                     f"{'-' if st['thm_frac'] is None else format(st['thm_frac'], '.2f')} | "
                     f"{', '.join(st['throttle'])[:60]} | {st['n_load']} |")
        for r in rows:
            csv.append(f"{spec},{r[0] - t0:.3f},{r[1]},{min(r[2]) if r[2] else ''},{max(r[2]) if r[2] else ''},{r[3]},{r[4]},"
                       f"{str(r[5])[:40]}")
        print(lines[-1], flush=True)
    s.stop = True
    open(out, "w").write("\n".join(lines) + "\n")
    open(out.replace(".md", ".csv"), "w").write("\n".join(csv) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
