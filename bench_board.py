"""Board sensors of the GPU a bench process computes on: shader clock and board power from the amdgpu hwmon files of THAT device (found by PCI address --
the card numbers under /sys/class/drm do not follow HIP's device order), sampled by a thread while a timed region runs.

Why the bench lines carry it: the 1400 W board limit, not a unit's throughput, sets the clock of this path's MFMA-heavy kernels on random operands
(profiles/r04h_power_cap_probe.txt: convolutions 1.72-1.76 GHz at 1400 W against 2.4 GHz / 1190 W for the same launches on all-zero operands), so a
roofline fraction against the 2.4 GHz peak reads differently once the sustained clock is next to it.  Measurement support only: no product path imports it."""
from __future__ import annotations

import glob
import os
import threading
import time
from typing import Dict, Optional

NOMINAL_SCLK_MHZ = 2400.0        # the clock the dense MFMA peaks of /opt/skills/guides/MI355X_MICROARCH.md are quoted at


def find_sensors(device_index: int = 0) -> Dict[str, object]:
    import torch
    out: Dict[str, object] = {}
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception as e:          # noqa: BLE001
        return {"error": f"no PCI address: {e}"}
    out["pci"] = bus
    hw = glob.glob(f"/sys/bus/pci/devices/{bus}/hwmon/hwmon*")
    if not hw:
        out["error"] = "no hwmon directory for this device"
        return out
    for name in ("power1_average", "power1_input"):
        if os.path.exists(os.path.join(hw[0], name)):
            out["power"] = os.path.join(hw[0], name)
            break
    if os.path.exists(os.path.join(hw[0], "freq1_input")):
        out["sclk"] = os.path.join(hw[0], "freq1_input")
    try:
        out["cap_w"] = int(open(os.path.join(hw[0], "power1_cap")).read()) / 1e6
    except Exception:               # noqa: BLE001
        pass
    return out


class BoardSampler:
    """with BoardSampler(dev_index) as b: ...timed region... ; b.summary() -> dict (or None when the sensors cannot be read)."""

    def __init__(self, device_index: int = 0, period_s: float = 0.05):
        self.sens = find_sensors(device_index)
        self.period = period_s
        self.mhz, self.w = [], []
        self._stop = threading.Event()
        self._th: Optional[threading.Thread] = None

    def _read(self, key, scale, dst):
        p = self.sens.get(key)
        if p:
            try:
                with open(p) as f:
                    dst.append(int(f.read()) / scale)
            except Exception:       # noqa: BLE001
                pass

    def _run(self):
        while not self._stop.is_set():
            self._read("sclk", 1e6, self.mhz)
            self._read("power", 1e6, self.w)
            time.sleep(self.period)

    def __enter__(self):
        if "sclk" in self.sens or "power" in self.sens:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join()
        return False

    def summary(self):
        if not self.mhz and not self.w:
            return None
        out = {"samples": max(len(self.mhz), len(self.w)), "period_s": self.period, "power_cap_w": self.sens.get("cap_w"), "pci": self.sens.get("pci")}
        if self.mhz:
            out.update({"sclk_mhz_mean": sum(self.mhz) / len(self.mhz), "sclk_mhz_min": min(self.mhz), "sclk_mhz_max": max(self.mhz),
                        "sclk_fraction_of_nominal": sum(self.mhz) / len(self.mhz) / NOMINAL_SCLK_MHZ})
        if self.w:
            out.update({"power_w_mean": sum(self.w) / len(self.w), "power_w_max": max(self.w)})
            if self.sens.get("cap_w"):
                out["share_of_samples_within_3pct_of_cap"] = sum(1 for x in self.w if x >= 0.97 * self.sens["cap_w"]) / len(self.w)
        return out
