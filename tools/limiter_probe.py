"""WHICH limiter holds the card below its 2.4 GHz peak under the prover's kernels?  

Round 4 inferred "a sustained power budget" from hwmon power / clock samples.  This probe reads what the power-management firmware
itself accounts: the throttler residency accumulators of `gpu_metrics` (accumulation_counter, ppt / socket-thermal / VR-thermal /
HBM-thermal / PROCHOT residency: kgd_pp_interface.h gpu_metrics_v1_6+; AMD SMI's amdsmi_get_violation_status adds the per-XCC
"gfx clock below the host limit because of power | thermal" counters) and the throttle-status words, sampled next to hwmon while

  phase `sched`   the headline schedule runs (ProofSchedule.run_once in this process, HBM-resident inputs),
  phase `mad`     tools/bin/energy_probe mad 2 <seconds>   (nothing but independent v_mad_i64_i32, 2 waves per SIMD),
  phase `mix`     tools/bin/energy_probe mix 2 <seconds>   (the mixed addition's instruction ratio),

with idle gaps in between.  Per phase: mean / max socket power and gfx clock, hotspot / HBM temperature, and for every residency
counter  delta(counter) / delta(accumulation_counter)  = the share of firmware iterations that limiter was ACTIVE.  The one that is
non-zero names the limiter; whether the synthetic streams, held for a minute, droop to the schedule's power answers whether the budget
is the card's or the kernel's.

usage (GPU box, repository root):  python tools/limiter_probe.py gpurun_out/r5/limiter.json [--sched-seconds 40] [--stream-seconds 60]
The sampler reads sysfs / AMD SMI only; the child streams are separate processes (never an exec from this one).
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/opt/rocm/share/amd_smi")

ACC_KEYS = ("acc_counter", "acc_prochot_thrm", "acc_ppt_pwr", "acc_socket_thrm", "acc_vr_thrm", "acc_hbm_thrm", "acc_gfx_clk_below_host_limit")
ACC_LISTS = ("acc_gfx_clk_below_host_limit_pwr", "acc_gfx_clk_below_host_limit_thm", "acc_gfx_clk_below_host_limit_total", "acc_low_utilization")
METRIC_KEYS = ("current_socket_power", "average_socket_power", "temperature_hotspot", "temperature_mem", "temperature_vrsoc", "throttle_status",
               "indep_throttle_status", "average_gfx_activity", "average_umc_activity", "accumulation_counter", "prochot_residency_acc",
               "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "current_uclk", "gfxclk_lock_status")


def num(v):
    return v if isinstance(v, (int, float)) else None


def flat_num_list(v):
    """AMD SMI hands the per-XCP / per-XCC arrays back nested; keep the numbers"""
    out = []
    for x in (v or []):
        if isinstance(x, (list, tuple)):
            out += [y for y in x if isinstance(y, (int, float))]
        elif isinstance(x, (int, float)):
            out.append(x)
    return out


class Sampler:
    def __init__(self, period=0.025):
        self.period = period
        self.rows = []
        self.marks = []
        self.err = {}
        self.smi = None
        self.handle = None
        self.hw = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            f = {k: os.path.join(d, k) for k in ("power1_average", "power1_input", "freq1_input", "temp2_input", "temp3_input", "power1_cap") if os.path.exists(os.path.join(d, k))}
            if "freq1_input" in f:
                self.hw.append((d, f))
        self._stop = threading.Event()
        self._th = None

    def attach_smi(self, bdf_hint=None):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.smi = amdsmi
            info = []
            for h in hs:
                try:
                    info.append((amdsmi.amdsmi_get_gpu_device_bdf(h), h))
                except Exception as e:
                    info.append((f"?{e}", h))
            self.devices = [b for b, _ in info]
            pick = None
            if bdf_hint:
                for b, h in info:
                    if b.lower().startswith(bdf_hint.lower()):
                        pick = h
            self.handle = pick if pick is not None else (info[0][1] if len(info) == 1 else None)
            self._all = info
            return True
        except Exception as e:
            self.err["amdsmi_init"] = repr(e)
            return False

    @staticmethod
    def _rd(p):
        try:
            with open(p) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            return None

    def sample(self):
        row = {"t": time.perf_counter(), "hw": []}
        for _, f in self.hw:
            row["hw"].append((self._rd(f.get("power1_average", f.get("power1_input", ""))), self._rd(f["freq1_input"]),
                              self._rd(f["temp2_input"]) if "temp2_input" in f else None, self._rd(f["temp3_input"]) if "temp3_input" in f else None))
        if self.smi is not None:
            targets = [self.handle] if self.handle is not None else [h for _, h in self._all]
            row["smi"] = []
            for h in targets:
                ent = {}
                try:
                    v = self.smi.amdsmi_get_violation_status(h)
                    for k in ACC_KEYS:
                        ent[k] = num(v.get(k))
                    for k in ACC_LISTS:
                        ent[k] = flat_num_list(v.get(k))
                except Exception as e:
                    self.err.setdefault("violation_status", repr(e))
                try:
                    m = self.smi.amdsmi_get_gpu_metrics_info(h)
                    for k in METRIC_KEYS:
                        ent[k] = num(m.get(k))
                    g = [x for x in (m.get("current_gfxclks") or []) if isinstance(x, (int, float)) and 0 < x < 60000]
                    ent["gfxclk_mean"] = sum(g) / len(g) if g else None
                    ent["gfxclk_min"] = min(g) if g else None
                    hb = [x for x in (m.get("temperature_hbm") or []) if isinstance(x, (int, float)) and 0 < x < 1000]
                    ent["temperature_hbm_max"] = max(hb) if hb else None
                except Exception as e:
                    self.err.setdefault("gpu_metrics", repr(e))
                row["smi"].append(ent)
        self.rows.append(row)

    def start(self):
        def loop():
            while not self._stop.is_set():
                self.sample()
                time.sleep(self.period)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        self._stop.set()
        self._th.join()

    def mark(self, name):
        self.marks.append((name, time.perf_counter()))


def raw_gpu_metrics():
    """header + hex of every readable gpu_metrics blob (the firmware table the counters above are parsed from)"""
    out = []
    for p in sorted(glob.glob("/sys/class/drm/card*/device/gpu_metrics")):
        try:
            b = open(p, "rb").read()
            out.append({"path": p, "bytes": len(b), "structure_size": int.from_bytes(b[0:2], "little"), "format_revision": b[2], "content_revision": b[3],
                        "hex": b.hex()})
        except OSError as e:
            out.append({"path": p, "error": repr(e)})
    return out


def summarize(s: Sampler, t0, t1, active_card):
    rows = [r for r in s.rows if t0 <= r["t"] <= t1]
    if len(rows) < 2:
        return {"samples": len(rows)}
    out = {"samples": len(rows), "seconds": t1 - t0}
    if s.hw and active_card is not None:
        pw = [r["hw"][active_card][0] / 1e6 for r in rows if r["hw"][active_card][0] is not None]
        fq = [r["hw"][active_card][1] / 1e6 for r in rows if r["hw"][active_card][1] is not None]
        tj = [r["hw"][active_card][2] / 1e3 for r in rows if r["hw"][active_card][2] is not None]
        tm = [r["hw"][active_card][3] / 1e3 for r in rows if r["hw"][active_card][3] is not None]
        if pw:
            half = len(pw) // 2
            out["hwmon"] = {"power_w_mean": sum(pw) / len(pw), "power_w_max": max(pw), "power_w_first_half": sum(pw[:half]) / max(half, 1),
                            "power_w_second_half": sum(pw[half:]) / max(len(pw) - half, 1),
                            "sclk_mhz_mean": sum(fq) / len(fq) if fq else None, "sclk_mhz_min": min(fq) if fq else None,
                            "sclk_mhz_first_half": sum(fq[:half]) / max(half, 1) if fq else None, "sclk_mhz_second_half": sum(fq[half:]) / max(len(fq) - half, 1) if fq else None,
                            "junction_c_max": max(tj) if tj else None, "mem_c_max": max(tm) if tm else None}
    if s.smi is not None and rows[0].get("smi"):
        ndev = len(rows[0]["smi"])
        devs = []
        for d in range(ndev):
            a, b = rows[0]["smi"][d], rows[-1]["smi"][d]
            ent = {}
            dc = None
            if a.get("acc_counter") is not None and b.get("acc_counter") is not None:
                dc = b["acc_counter"] - a["acc_counter"]
            ent["firmware_iterations"] = dc
            for k in ACC_KEYS[1:]:
                if a.get(k) is not None and b.get(k) is not None:
                    ent[k + "_delta"] = b[k] - a[k]
                    ent[k.replace("acc_", "share_")] = (b[k] - a[k]) / dc if dc else None
            for k in ACC_LISTS:
                la, lb = a.get(k) or [], b.get(k) or []
                if la and len(la) == len(lb):
                    dl = [y - x for x, y in zip(la, lb)]
                    ent[k + "_delta_per_xcc"] = dl
                    nz = [v for v in dl if v]
                    ent[k.replace("acc_", "share_") + "_mean"] = (sum(dl) / len(dl) / dc) if (dc and dl) else None
            # the same residencies straight from gpu_metrics (cross-check of the violation-status path)
            for k in ("accumulation_counter", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc"):
                if a.get(k) is not None and b.get(k) is not None:
                    ent["metrics_" + k + "_delta"] = b[k] - a[k]
            for k in ("current_socket_power", "gfxclk_mean", "temperature_hotspot", "temperature_mem", "temperature_hbm_max", "average_gfx_activity", "average_umc_activity", "current_uclk"):
                v = [r["smi"][d].get(k) for r in rows if r["smi"][d].get(k) is not None]
                if v:
                    ent[k] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v)}
            ts = 0
            its = 0
            for r in rows:
                ts |= int(r["smi"][d].get("throttle_status") or 0)
                its |= int(r["smi"][d].get("indep_throttle_status") or 0)
            ent["throttle_status_or"] = ts
            ent["indep_throttle_status_or"] = its
            devs.append(ent)
        out["smi"] = devs
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--sched-seconds", type=float, default=40.0)
    ap.add_argument("--stream-seconds", type=float, default=60.0)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--idle-seconds", type=float, default=6.0)
    ap.add_argument("--phases", default="sched,mad,mix,sched2")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd.prover_schedule import ProofSchedule
    from bench import build_srs

    pr = torch.cuda.get_device_properties(0)
    bdf = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    s = Sampler()
    smi_ok = s.attach_smi(bdf)
    report = {"device": {"name": pr.name, "bdf": bdf}, "amdsmi": smi_ok, "amdsmi_devices": getattr(s, "devices", None),
              "amdsmi_matched_by_bdf": s.handle is not None, "raw_before": raw_gpu_metrics()}
    if smi_ok and s.handle is not None:
        try:
            report["metrics_header"] = s.smi.amdsmi_get_gpu_metrics_header_info(s.handle)
        except Exception as e:
            report["metrics_header"] = repr(e)
        try:
            report["power_cap"] = {k: (v if isinstance(v, (int, float, str)) else str(v)) for k, v in s.smi.amdsmi_get_power_cap_info(s.handle).items()}
        except Exception as e:
            report["power_cap"] = repr(e)
    report["hwmon_caps_w"] = [Sampler._rd(f["power1_cap"]) / 1e6 if "power1_cap" in f and Sampler._rd(f["power1_cap"]) else None for _, f in s.hw]

    ctx = zk.Context(0)
    ctx.use_torch_stream()
    cv = zk.get_curve("bls12_381")
    n = 1 << args.log_n
    ck = zk.CommitterKey(build_srs(ctx, cv, n, 0, n, torch), cv, ctx).precompute()
    sched = ProofSchedule(args.log_n, ctx, ck, cv)
    for _ in range(2):
        sched.run_once()
    torch.cuda.synchronize()
    s.start()
    phases = {}
    time.sleep(args.idle_seconds)
    s.mark("start")
    for ph in args.phases.split(","):
        t_idle0 = time.perf_counter()
        time.sleep(args.idle_seconds)
        phases["idle_before_" + ph] = (t_idle0, time.perf_counter())
        t0 = time.perf_counter()
        extra = {}
        if ph.startswith("sched"):
            k = 0
            while time.perf_counter() - t0 < args.sched_seconds:
                sched.run_once()
                k += 1
            torch.cuda.synchronize()
            extra = {"proofs": k}
        elif ph == "msm":
            # nothing but commitments: 16-job batches, the accumulation is ~85 % of their time
            k = 0
            while time.perf_counter() - t0 < args.sched_seconds:
                ck.commit_batch(sched.coef[:13] + sched.coef[:3])
                k += 16
            extra = {"msms": k}
        elif ph == "ntt":
            # nothing but transforms: the twelve coset ffts of the quotient round as one batch
            k = 0
            srcs = sched.coef[:12]
            outs = [sched.cos[name] for name in list(sched.cos)[:12]]
            while time.perf_counter() - t0 < args.sched_seconds:
                sched.dom_4n.batch(2, srcs, outs=outs)
                k += 12
                if k % 120 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            extra = {"coset_ffts_4n": k}
        else:
            exe = os.path.join(ROOT, "tools", "bin", "energy_probe")
            r = subprocess.run([exe, ph, "2", str(args.stream_seconds)], capture_output=True, text=True)
            extra = {"child_rc": r.returncode, "child_stdout": r.stdout.strip()[-400:]}
        t1 = time.perf_counter()
        phases[ph] = (t0, t1, extra)
        print(f"phase {ph}: {t1 - t0:.1f} s {extra}", flush=True)
    time.sleep(2.0)
    s.stop()
    ck.close()
    # the card of this process among the hwmon directories of the host (other tenants' cards are visible too): by PCI address
    active = None
    for h, (d_, _) in enumerate(s.hw):
        if os.path.basename(os.path.realpath(os.path.join(d_, "..", ".."))).lower().startswith(bdf.lower()):
            active = h
    report["hwmon_matched_by_pci_address"] = active is not None
    report["hwmon_card"] = s.hw[active][0] if active is not None else None
    report["hwmon_cap_w"] = report["hwmon_caps_w"][active] if active is not None else None
    report["errors"] = s.err
    report["phases"] = {}
    for name, v in phases.items():
        t0, t1 = v[0], v[1]
        d = summarize(s, t0, t1, active)
        if len(v) > 2:
            d.update(v[2])
            if "proofs" in v[2]:
                d["proofs_per_s"] = v[2]["proofs"] / (t1 - t0)
                if d.get("hwmon"):
                    d["joules_per_proof"] = d["hwmon"]["power_w_mean"] * (t1 - t0) / max(v[2]["proofs"], 1)
        report["phases"][name] = d
    report["raw_after"] = [{k: v for k, v in e.items() if k != "hex"} for e in raw_gpu_metrics()]
    # the verdict in one field: which residency counters moved during the schedule
    lim = []
    sm = (report["phases"].get("sched") or {}).get("smi") or []
    for e in sm:
        for k, v in e.items():
            if k.startswith("share_") and isinstance(v, (int, float)) and v and v > 0.01:
                lim.append((k, round(v, 4)))
    report["limiters_active_during_sched"] = lim
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    # a short text rendering
    for name, d in report["phases"].items():
        hw = d.get("hwmon") or {}
        line = f"{name:18s} {d.get('seconds', 0):6.1f} s  {hw.get('power_w_mean', 0):7.1f} W (halves {hw.get('power_w_first_half', 0):7.1f} / {hw.get('power_w_second_half', 0):7.1f})  " \
               f"sclk {hw.get('sclk_mhz_mean') or 0:7.1f} MHz (halves {hw.get('sclk_mhz_first_half') or 0:7.1f} / {hw.get('sclk_mhz_second_half') or 0:7.1f})  Tj max {hw.get('junction_c_max')}"
        for e in d.get("smi") or []:
            line += "  | " + "  ".join(f"{k[6:]}={v:.3f}" for k, v in e.items() if k.startswith("share_") and isinstance(v, (int, float)))
            line += f"  thr=0x{e.get('throttle_status_or', 0):x} indep=0x{e.get('indep_throttle_status_or', 0):x}"
        print(line, flush=True)
    print("limiters active during sched:", lim, "errors:", s.err)


if __name__ == "__main__":
    main()
