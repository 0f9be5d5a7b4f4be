"""Socket power / shader clock sampling and the firmware's throttler residencies for bench.py's `power` leg (moved out of bench.py in round 6:
plain sysfs and AMD SMI reads, no GPU call; never inside the timed region of `value`)."""
from __future__ import annotations

import glob
import os
import sys
import threading
import time


class PowerSampler:
    """Socket power and shader clock of the card the process runs on, from the amdgpu hwmon files (plain sysfs reads, no GPU call),
    sampled by a thread while a leg runs.  A box shows the hwmon of every card of its host -- other tenants' cards too: the card is
    the one whose PCI address is `bdf` (the sysfs `device` link of the card names it); only without a match, the one whose power
    moved most (rounds 1-4's heuristic: wrong whenever a neighbour's job starts or stops meanwhile -- profiles/r05/r05_notes.md).  Used by
    the `power` leg only -- never inside the timed region of `value`."""

    WANT = ("power1_average", "power1_input", "freq1_input", "temp2_input")

    def __init__(self, period_s: float = 0.004, bdf: str | None = None):
        import glob
        self.hw = []
        self.matched_bdf = False
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            files = {k: os.path.join(d, k) for k in self.WANT + ("power1_cap",) if os.path.exists(os.path.join(d, k))}
            if "freq1_input" in files and ("power1_average" in files or "power1_input" in files):
                self.hw.append((d, files))
        if bdf:
            mine = [(d, f) for d, f in self.hw if os.path.basename(os.path.realpath(os.path.join(d, "..", ".."))).lower().startswith(bdf.lower())]
            if mine:
                self.hw, self.matched_bdf = mine[:1], True
        self.period = period_s
        self.rows = []
        self._stop = None
        self._th = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            return None

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                row = [time.perf_counter()]
                for _, files in self.hw:
                    pw = self._read(files.get("power1_average", files.get("power1_input", "")))
                    row += [pw, self._read(files["freq1_input"]), self._read(files["temp2_input"]) if "temp2_input" in files else None]
                self.rows.append(row)
                time.sleep(self.period)

        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._th.join()

    def summary(self, t0: float, t1: float):
        if not self.hw or not self.rows:
            return None
        best, span = None, -1.0
        for h in range(len(self.hw)):
            v = [r[1 + 3 * h] for r in self.rows if r[1 + 3 * h] is not None]
            if v and max(v) - min(v) > span:
                best, span = h, max(v) - min(v)
        if best is None:
            return None
        sel = [r for r in self.rows if t0 <= r[0] <= t1]
        pw = sorted(r[1 + 3 * best] / 1e6 for r in sel if r[1 + 3 * best] is not None)
        fq = sorted(r[2 + 3 * best] / 1e6 for r in sel if r[2 + 3 * best] is not None)
        tj = [r[3 + 3 * best] / 1e3 for r in sel if r[3 + 3 * best] is not None]
        if not pw or not fq:
            return None
        cap = self._read(self.hw[best][1].get("power1_cap", ""))
        return {"socket_power_w": {"mean": sum(pw) / len(pw), "median": pw[len(pw) // 2], "max": pw[-1]},
                "sclk_mhz": {"mean": sum(fq) / len(fq), "median": fq[len(fq) // 2], "min": fq[0], "max": fq[-1]},
                "junction_c_max": max(tj) if tj else None, "power_cap_w": cap / 1e6 if cap else None, "samples": len(sel),
                "hwmon": self.hw[best][0], "card_matched_by_pci_address": self.matched_bdf}


class FirmwareThrottlers:
    """The throttler residency accumulators of the card's power-management firmware (gpu_metrics v1.6+: accumulation_counter and the
    PPT / socket-thermal / VR-thermal / HBM-thermal / PROCHOT residencies), read through AMD SMI (`amdsmi_get_violation_status`: sysfs
    underneath, no GPU call) for the card with PCI address `bdf`.  between(a, b) = the share of firmware iterations each limiter was
    ACTIVE between two snapshots: the one that is non-zero NAMES what holds the clock below its peak (profiles/r05/r05_notes.md)."""

    KEYS = {"ppt": "acc_ppt_pwr", "socket_thermal": "acc_socket_thrm", "vr_thermal": "acc_vr_thrm", "hbm_thermal": "acc_hbm_thrm", "prochot": "acc_prochot_thrm"}

    def __init__(self, bdf: str):
        self.smi, self.h, self.error = None, None, None
        try:
            sys.path.insert(0, "/opt/rocm/share/amd_smi")
            import amdsmi
            amdsmi.amdsmi_init()
            for h in amdsmi.amdsmi_get_processor_handles():
                if amdsmi.amdsmi_get_gpu_device_bdf(h).lower().startswith(bdf.lower()):
                    self.smi, self.h = amdsmi, h
            if self.h is None:
                self.error = f"no AMD SMI processor with PCI address {bdf}"
        except Exception as e:
            self.error = repr(e)

    def snapshot(self):
        if self.h is None:
            return None
        try:
            v = self.smi.amdsmi_get_violation_status(self.h)
            return {k: v.get(k) for k in ("acc_counter",) + tuple(self.KEYS.values())}
        except Exception as e:
            self.error = repr(e)
            return None

    def between(self, a, b):
        if not a or not b or not isinstance(a.get("acc_counter"), int) or not isinstance(b.get("acc_counter"), int) or b["acc_counter"] <= a["acc_counter"]:
            return None
        it = b["acc_counter"] - a["acc_counter"]
        shares = {name: (b[k] - a[k]) / it for name, k in self.KEYS.items() if isinstance(a.get(k), int) and isinstance(b.get(k), int)}
        active = {k: v for k, v in shares.items() if v > 0.02}
        return {"firmware_iterations": it, "active_share": shares, "limiter": max(active, key=active.get) if active else None,
                "source": "AMD SMI amdsmi_get_violation_status (gpu_metrics throttler residency accumulators): share of firmware iterations "
                          "each limiter was active during the sampled run"}


