#!/usr/bin/env python3
"""Socket power / shader clock telemetry for bench.py (measurement infrastructure, not product).

Why: every matrix kernel of the front-end call runs ON the board's power cap (profiles/r04_p_*): the headline is the cap divided by the joules
one stereo pair costs, so joules - not clocks, not launch microseconds - are the unit a kernel change has to be priced in (VERDICT r05 "do
this" 3).  The reference has no counterpart (include/Profiling.h:14-74 is wall clock only).

A poller thread samples (monotonic time, socket watts, shader MHz) at >= 20 Hz through the first backend that answers for the GPU under the
HIP device index given:
  1. sysfs hwmon of the device's PCI function (power1_average / power1_input in microwatts, freq1_input in Hz, power1_cap) - a file read, ~30 us;
  2. librocm_smi64 through ctypes (rsmi_dev_power_get / rsmi_dev_current_socket_power_get, rsmi_dev_gpu_clk_freq_get, rsmi_dev_power_cap_get);
  3. the `rocm-smi --json` command line (slow: ~3 Hz; last resort).
`summarize(samples, t0, t1)` turns a window of samples into {avg_W, max_W, sclk_MHz, n}; `parse_rocm_smi_text` reads the text form the round-4
telemetry was recorded in (tests/test_power_telemetry.py feeds it the committed sample).  Nothing here needs root.
"""
from __future__ import annotations

import ctypes as C
import glob
import json
import os
import re
import subprocess
import threading
import time


def hip_pci_bus_id(device_index: int = 0) -> str | None:
    """'0000:c1:00.0' of a HIP device (hipDeviceGetPCIBusId), lower case; None when HIP is not there."""
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
            return None
        return buf.value.decode().lower()
    except OSError:
        return None


class _Sysfs:
    name = "sysfs-hwmon"

    def __init__(self, bus_id: str):
        base = f"/sys/bus/pci/devices/{bus_id}"
        mons = sorted(glob.glob(base + "/hwmon/hwmon*"))
        if not mons:
            raise RuntimeError("no hwmon under " + base)
        self.mon = mons[0]
        self.fpow = next((p for p in (self.mon + "/power1_average", self.mon + "/power1_input") if self._readable(p)), None)
        if self.fpow is None:
            raise RuntimeError("no readable power1_average / power1_input in " + self.mon)
        self.fclk = self.mon + "/freq1_input" if self._readable(self.mon + "/freq1_input") else None
        self.fdpm = base + "/pp_dpm_sclk" if self._readable(base + "/pp_dpm_sclk") else None
        if self.read()[0] <= 0:
            raise RuntimeError("power reads 0 from " + self.fpow)

    @staticmethod
    def _readable(p):
        try:
            with open(p) as f:
                f.read()
            return True
        except OSError:
            return False

    def cap_w(self):
        for p in (self.mon + "/power1_cap", self.mon + "/power1_cap_max"):
            try:
                return int(open(p).read()) / 1e6
            except (OSError, ValueError):
                pass
        return None

    def read(self):
        w = int(open(self.fpow).read()) / 1e6
        mhz = 0.0
        if self.fclk:
            mhz = int(open(self.fclk).read()) / 1e6
        elif self.fdpm:
            m = re.search(r"(\d+)\s*[Mm][Hh]z\s*\*", open(self.fdpm).read())
            mhz = float(m.group(1)) if m else 0.0
        return w, mhz


class _Rsmi:
    name = "librocm_smi64"

    class _Freq(C.Structure):
        _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]

    def __init__(self, bus_id: str | None):
        self.lib = C.CDLL("librocm_smi64.so")
        if self.lib.rsmi_init(C.c_uint64(0)) != 0:
            raise RuntimeError("rsmi_init failed")
        n = C.c_uint32(0)
        self.lib.rsmi_num_monitor_devices(C.byref(n))
        self.dv = None
        want = None
        if bus_id:
            m = re.match(r"([0-9a-f]+):([0-9a-f]+):([0-9a-f]+)\.([0-9a-f])", bus_id)
            if m:
                dom, bus, dev, fn = (int(x, 16) for x in m.groups())
                want = (dom, bus, dev, fn)
        for i in range(n.value):
            bdf = C.c_uint64(0)
            if self.lib.rsmi_dev_pci_id_get(C.c_uint32(i), C.byref(bdf)) != 0:
                continue
            v = bdf.value   # BDFID = ((DOMAIN & 0xffffffff) << 32) | ((BUS & 0xff) << 8) | ((DEVICE & 0x1f) << 3) | (FUNCTION & 0x7)
            got = ((v >> 32) & 0xffffffff, (v >> 8) & 0xff, (v >> 3) & 0x1f, v & 0x7)
            if want is None or got == want:
                self.dv = C.c_uint32(i)
                break
        if self.dv is None:
            if n.value == 0:
                raise RuntimeError("rsmi: no devices")
            self.dv = C.c_uint32(0)
        if self.read()[0] <= 0:
            raise RuntimeError("rsmi: power reads 0")

    def cap_w(self):
        cap = C.c_uint64(0)
        if self.lib.rsmi_dev_power_cap_get(self.dv, C.c_uint32(0), C.byref(cap)) == 0 and cap.value:
            return cap.value / 1e6
        return None

    def read(self):
        uw = C.c_uint64(0)
        ok = False
        if hasattr(self.lib, "rsmi_dev_power_get"):
            typ = C.c_int(0)
            ok = self.lib.rsmi_dev_power_get(self.dv, C.byref(uw), C.byref(typ)) == 0
        if not ok and hasattr(self.lib, "rsmi_dev_current_socket_power_get"):
            ok = self.lib.rsmi_dev_current_socket_power_get(self.dv, C.byref(uw)) == 0
        if not ok:
            ok = self.lib.rsmi_dev_power_ave_get(self.dv, C.c_uint32(0), C.byref(uw)) == 0
        f = self._Freq()
        mhz = 0.0
        if self.lib.rsmi_dev_gpu_clk_freq_get(self.dv, C.c_int(0), C.byref(f)) == 0 and f.current < 33:
            mhz = f.frequency[f.current] / 1e6
        return (uw.value / 1e6 if ok else 0.0), mhz


def parse_rocm_smi_json(text: str):
    """(watts, MHz) of the first card in `rocm-smi --showpower --showclocks --json` output."""
    j = json.loads(text)
    card = j[sorted(k for k in j if k.startswith("card"))[0]]
    w = mhz = 0.0
    for k, v in card.items():
        kl = k.lower()
        if "power" in kl and "(w)" in kl and w == 0.0:
            try:
                w = float(v)
            except ValueError:
                pass
        if kl.startswith("sclk clock speed"):
            m = re.search(r"(\d+)\s*mhz", str(v).lower())
            mhz = float(m.group(1)) if m else 0.0
    return w, mhz


def parse_rocm_smi_text(text: str):
    """[(watts, MHz), ...] from the plain-text lines scripts/dev/power_poll.sh recorded in rounds 4-5, e.g.
    'GPU[0] : Current Socket Graphics Package Power (W): 1341.0 GPU[0] : sclk clock level: 3: (2031Mhz) ...' - one tuple per line holding both."""
    out = []
    for line in text.splitlines():
        pw = re.search(r"Power \(W\):\s*([0-9.]+)", line)
        ck = re.search(r"sclk[^()]*\((\d+)\s*[Mm][Hh]z\)", line)
        if pw and ck:
            out.append((float(pw.group(1)), float(ck.group(1))))
    return out


class _Cli:
    name = "rocm-smi --json"

    def __init__(self, _bus_id):
        if self.read()[0] <= 0:
            raise RuntimeError("rocm-smi: no power figure")

    def cap_w(self):
        try:
            t = subprocess.run(["rocm-smi", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=20).stdout
            j = json.loads(t)
            card = j[sorted(k for k in j if k.startswith("card"))[0]]
            for k, v in card.items():
                if "max" in k.lower() and "power" in k.lower():
                    return float(v)
        except Exception:
            pass
        return None

    def read(self):
        t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        return parse_rocm_smi_json(t)


def open_backend(device_index: int = 0, verbose: bool = False):
    bus = hip_pci_bus_id(device_index)
    errs = []
    for make in ((lambda: _Sysfs(bus)) if bus else None, lambda: _Rsmi(bus), lambda: _Cli(bus)):
        if make is None:
            continue
        try:
            return make()
        except Exception as e:   # a backend that is not there is not an error: the next one is tried
            errs.append(f"{type(e).__name__}: {e}")
    if verbose:
        print("power telemetry: no backend (" + "; ".join(errs) + ")")
    return None


class PowerPoller:
    """with PowerPoller(0) as p: ...; p.window(t0, t1) -> summary of the samples taken between two time.monotonic() stamps."""

    def __init__(self, device_index: int = 0, hz: float = 50.0):
        self.backend = open_backend(device_index)
        self.hz = hz
        self.samples = []          # (t, watts, MHz)
        self._stop = threading.Event()
        self._th = None
        self.cap_w = self.backend.cap_w() if self.backend else None

    @property
    def ok(self):
        return self.backend is not None

    def _run(self):
        period = 1.0 / self.hz
        nxt = time.monotonic()
        while not self._stop.is_set():
            try:
                w, mhz = self.backend.read()
                self.samples.append((time.monotonic(), w, mhz))
            except Exception:
                pass
            nxt += period
            d = nxt - time.monotonic()
            if d > 0:
                self._stop.wait(d)
            else:
                nxt = time.monotonic()

    def __enter__(self):
        if self.backend:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._th:
            self._th.join(timeout=5)

    def window(self, t0: float, t1: float, settle_s: float = 0.0):
        return summarize(self.samples, t0 + settle_s, t1)


def summarize(samples, t0: float, t1: float):
    """{avg_W, max_W, sclk_MHz, n, rate_hz} over the samples with t0 <= t <= t1 (None when there are none)."""
    sel = [(t, w, c) for (t, w, c) in samples if t0 <= t <= t1 and w > 0]
    if not sel:
        return None
    ws = [w for _, w, _ in sel]
    cs = [c for _, _, c in sel if c > 0]
    span = sel[-1][0] - sel[0][0]
    return {"avg_W": round(sum(ws) / len(ws), 1), "max_W": round(max(ws), 1), "min_W": round(min(ws), 1),
            "sclk_MHz": round(sum(cs) / len(cs), 0) if cs else None, "n": len(sel),
            "rate_hz": round((len(sel) - 1) / span, 1) if span > 0 and len(sel) > 1 else None}


def energy_block(summary, cap_w, units_per_s: float, unit: str = "pair"):
    """The derived figures next to a power summary: joules per unit = avg_W / (units per second); what the cap would allow at that energy."""
    if not summary or not units_per_s:
        return None
    j = summary["avg_W"] / units_per_s
    out = dict(summary)
    out["cap_W"] = cap_w
    out[f"joules_per_{unit}"] = round(j, 5)
    if cap_w:
        out[f"{unit}s_per_s_at_cap"] = round(cap_w / j, 1)
        out["frac_of_cap"] = round(summary["avg_W"] / cap_w, 4)
    return out


if __name__ == "__main__":   # probe: which backend answers on this box, at what rate
    import sys

    dev = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    print("pci bus id:", hip_pci_bus_id(dev))
    b = open_backend(dev, verbose=True)
    print("backend:", b.name if b else None, "cap_W:", b.cap_w() if b else None)
    if b:
        t0 = time.monotonic()
        n = 0
        while time.monotonic() - t0 < 1.0 and n < 2000:
            r = b.read(); n += 1
        print(f"{n} reads in {time.monotonic() - t0:.2f} s; last (W, MHz) = {r}")
    for extra in (_Rsmi, _Cli):
        try:
            e = extra(hip_pci_bus_id(dev))
            print("also available:", e.name, e.read(), "cap", e.cap_w())
        except Exception as ex:
            print("not available:", extra.name, ex)
