"""Shader clock / package power / junction temperature of one device, sampled beside the timed regions."""
import ctypes
import json
import os
import time

# the sampler of this process (rank 0 only), set by bench.py once the device is known; the airfri / extras legs read it
ACTIVE = None


class Telemetry:
    """Shader clock, package power and junction temperature of ONE device, sampled from a side thread while the
    timed regions run (the 50 ms window of rounds 1 - 4 sat inside the power controller's ramp,
    profiles/r03_power_clock_bulk.txt).  Source: the amdgpu hwmon files of the PCI function HIP reports for the
    device (freq1_input = sclk in Hz, power1_input = socket power in uW, temp2_input = junction in mC); a box whose
    sysfs does not show them falls back to `rocm-smi --json`.  Reading costs well under a millisecond and the
    timed loop spends its time inside ctypes calls that release the GIL."""

    def __init__(self, dev_index, period_s=0.02):
        import threading
        self.period = period_s
        self.samples = []  # (t, sclk_mhz, power_w, temp_c)
        self._stop = threading.Event()
        self._thread = None
        self.source = None
        self._files = self._find_hwmon(dev_index)
        if self._files:
            self.source = "sysfs hwmon " + self._files["dir"]
        else:
            import shutil
            if shutil.which("rocm-smi"):
                self.source = "rocm-smi --showpower --showclocks --json (card0)"
                self.period = max(period_s, 0.25)

    @staticmethod
    def _find_hwmon(dev_index):
        import glob
        try:
            # the HIP runtime this process already runs on (torch's, loaded RTLD_GLOBAL by starkperp._lib): never
            # dlopen a second libamdhip64 by name
            bus = None
            try:
                buf = ctypes.create_string_buffer(64)
                if ctypes.CDLL(None).hipDeviceGetPCIBusId(buf, 64, int(dev_index)) == 0:
                    bus = buf.value.decode().lower()
            except (AttributeError, OSError):
                bus = None
            if not bus:
                import torch
                pr = torch.cuda.get_device_properties(int(dev_index))
                bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for d in glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bus):
                if os.path.exists(os.path.join(d, "freq1_input")) and (
                        os.path.exists(os.path.join(d, "power1_input")) or os.path.exists(os.path.join(d, "power1_average"))):
                    return {"dir": d, "bus": bus, "sclk": os.path.join(d, "freq1_input"),
                            "power": os.path.join(d, "power1_input") if os.path.exists(os.path.join(d, "power1_input"))
                            else os.path.join(d, "power1_average"),
                            "temp": os.path.join(d, "temp2_input"), "cap": os.path.join(d, "power1_cap")}
        except Exception:  # noqa: BLE001 - telemetry never breaks the measurement
            return None
        return None

    @staticmethod
    def _read_num(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:  # noqa: BLE001
            return None

    def sample(self):
        t = time.perf_counter()
        if self._files:
            sclk, pw, tc = (self._read_num(self._files[k]) for k in ("sclk", "power", "temp"))
            self.samples.append((t, None if sclk is None else sclk / 1e6, None if pw is None else pw / 1e6,
                                 None if tc is None else tc / 1e3))
        elif self.source:
            import subprocess
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True,
                                     text=True, timeout=10).stdout
                card = next(iter(json.loads(out).values()))
                sclk = pw = None
                for k, v in card.items():
                    if k.lower().startswith("sclk clock speed"):
                        sclk = float("".join(ch for ch in v if ch.isdigit() or ch == "."))
                    if "power (w)" in k.lower():
                        pw = float(v)
                self.samples.append((t, sclk, pw, None))
            except Exception:  # noqa: BLE001
                pass

    def start(self):
        import threading
        if not self.source or self._thread is not None:
            return self

        def run():
            while not self._stop.is_set():
                self.sample()
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def window(self, t0, t1):
        """Median / min / max of the samples taken between the perf_counter times t0 and t1."""
        rows = [s for s in self.samples if t0 <= s[0] <= t1]

        def stat(i):
            v = sorted(x[i] for x in rows if x[i] is not None)
            if not v:
                return None
            return {"median": v[len(v) // 2], "min": v[0], "max": v[-1]}
        sclk, pw, tc = stat(1), stat(2), stat(3)
        return {"samples": len(rows), "seconds": t1 - t0,
                "sclk_mhz_median": sclk and sclk["median"], "sclk_mhz_min": sclk and sclk["min"],
                "sclk_mhz_max": sclk and sclk["max"],
                "power_w_median": pw and pw["median"], "power_w_min": pw and pw["min"], "power_w_max": pw and pw["max"],
                "junction_c_median": tc and tc["median"]}

    def describe(self):
        cap = self._read_num(self._files["cap"]) if self._files else None
        return {"source": self.source, "period_s": self.period, "pci_bus": self._files and self._files["bus"],
                "power_cap_w": None if cap is None else cap / 1e6}
