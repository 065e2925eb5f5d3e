"""Device telemetry beside a measurement: shader / memory / fabric clocks, socket power and the partition modes, read from
the amdgpu driver's sysfs files by a sampler THREAD of the measuring process (no child process, no exec: the GPU boxes
refuse an exec from a process that has initialised the GPU, and rocm-smi is a program).

Not on the hot path and not part of the C ABI: bench.py's `device_calibration` and tools/devclass_probe.py use it to state
what distinguishes a device on which long-lived writer blocks reach 5.0 TB/s from one on which they reach 6.3
(VERDICT round 5, item 4a).  Everything is best effort: a file that does not exist or cannot be read is skipped.
"""
import glob
import os
import re
import struct
import threading
import time


def _read(path, binary=False):
    try:
        with open(path, "rb" if binary else "r") as f:
            return f.read()
    except Exception:
        return None


def list_cards():
    """[(sysfs device directory, PCI address 'dddd:bb:dd.f')] of every amdgpu card this container can see."""
    cards = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"), key=lambda p: int(re.findall(r"card(\d+)", p)[0])):
        if (_read(os.path.join(d, "vendor")) or "").strip() != "0x1002":
            continue
        cards.append((d, os.path.basename(os.path.realpath(d)).lower()))
    return cards


def find_card(device_index=0, pci_address=None):
    """sysfs directory of the card at `pci_address` ('dddd:bb:dd.f', as hipDeviceGetPCIBusId / torch's device properties give
    it) -- the GPU a process computes on is NOT in general card0: a box's container may see the sysfs nodes of cards it cannot
    compute on.  Without an address: the `device_index`-th amdgpu card (only right on a single-card host).  None if absent."""
    cards = list_cards()
    if pci_address:
        want = pci_address.lower()
        for d, addr in cards:
            if addr == want or addr.endswith(want) or want.endswith(addr):
                return d
        return None
    return cards[device_index][0] if device_index < len(cards) else None


def torch_pci_address(device_index=0):
    """PCI address of torch's cuda:`device_index` ('dddd:bb:dd.0'); None when torch does not expose it."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def _hwmon(card):
    h = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*")))
    return h[0] if h else None


def _dpm_current(text):
    """`pp_dpm_*` lists levels, the current one marked '*': -> MHz"""
    if not text:
        return None
    for line in text.splitlines():
        if "*" in line:
            m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
            if m:
                return int(m.group(1))
    return None


def parse_gpu_metrics(blob):
    """gpu_metrics of the MI300 / MI350 family (format 1, content revision 4-8; MI355X reports 1.8, 3,872 bytes): the header, then fixed-width little-endian fields.
    Only the leading fields whose layout is the same in every one of these revisions are decoded:
        u16 temperature_hotspot, temperature_mem, temperature_vrsoc, curr_socket_power (W), average_gfx_activity,
        average_umc_activity
    -> dict, or None when the blob is not of this family."""
    if not blob or len(blob) < 16:
        return None
    size, fmt, rev = struct.unpack_from("<HBB", blob, 0)
    out = {"gpu_metrics_size": size, "gpu_metrics_format": fmt, "gpu_metrics_revision": rev}
    if fmt == 1 and 4 <= rev <= 8:
        t_hot, t_mem, t_vr, power, gfx_act, umc_act = struct.unpack_from("<6H", blob, 4)
        out.update({"metrics_socket_power_W": power, "metrics_gfx_activity": gfx_act, "metrics_umc_activity": umc_act,
                    "metrics_temp_hotspot_C": t_hot, "metrics_temp_mem_C": t_mem})
    return out


def static_state(card):
    """What does not change during a run: partition modes, power cap, clock level tables."""
    if not card:
        return {}
    out = {"sysfs": card}
    for key in ("current_compute_partition", "current_memory_partition", "available_compute_partition"):
        v = _read(os.path.join(card, key))
        if v is not None:
            out[key] = v.strip()
    h = _hwmon(card)
    if h:
        for key in ("power1_cap", "power1_cap_max", "power1_cap_default"):
            v = _read(os.path.join(h, key))
            if v and v.strip().isdigit():
                out[key + "_W"] = int(v) / 1e6
    for key in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        v = _read(os.path.join(card, key))
        if v:
            out[key] = " | ".join(x.strip() for x in v.strip().splitlines())
    v = _read(os.path.join(card, "power_dpm_force_performance_level"))
    if v:
        out["performance_level"] = v.strip()
    return out


def sample(card):
    """One reading of everything that moves."""
    if not card:
        return {}
    s = {}
    h = _hwmon(card)
    if h:
        for name, key, scale in (("freq1_input", "sclk_MHz", 1e-6), ("freq2_input", "mclk_MHz", 1e-6),
                                 ("power1_average", "power_W", 1e-6), ("power1_input", "power_input_W", 1e-6),
                                 ("temp1_input", "temp_edge_C", 1e-3), ("temp2_input", "temp_junction_C", 1e-3),
                                 ("temp3_input", "temp_mem_C", 1e-3)):
            v = _read(os.path.join(h, name))
            if v and v.strip().lstrip("-").isdigit():
                s[key] = int(v) * scale
    for name, key in (("pp_dpm_sclk", "dpm_sclk_MHz"), ("pp_dpm_mclk", "dpm_mclk_MHz"), ("pp_dpm_fclk", "dpm_fclk_MHz"),
                      ("pp_dpm_socclk", "dpm_socclk_MHz")):
        v = _dpm_current(_read(os.path.join(card, name)))
        if v is not None:
            s[key] = v
    v = _read(os.path.join(card, "gpu_busy_percent"))
    if v and v.strip().isdigit():
        s["gpu_busy_percent"] = int(v)
    m = parse_gpu_metrics(_read(os.path.join(card, "gpu_metrics"), binary=True))
    if m:
        s.update({k: v for k, v in m.items() if k.startswith("metrics_")})
    return s


class Sampler:
    """Samples `sample(card)` every `period_s` on a thread between start() and stop(); summary() gives the median, minimum
    and maximum of every field.  The measuring thread spends its time inside ctypes / torch calls that release the GIL."""

    def __init__(self, card, period_s=0.005):
        self.card, self.period, self.rows = card, period_s, []
        self._stop = threading.Event()
        self._t = None

    def start(self):
        self.rows = []
        self._stop.clear()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            s = sample(self.card)
            if s:
                self.rows.append(s)
            self._stop.wait(self.period)

    def stop(self):
        self._stop.set()
        if self._t:
            self._t.join()
        return self.summary()

    def __enter__(self):
        return self.start()

    def __exit__(self, *a):
        self.stop()

    def summary(self):
        out = {"samples": len(self.rows)}
        keys = sorted({k for r in self.rows for k in r})
        for k in keys:
            v = sorted(r[k] for r in self.rows if k in r)
            if v:
                out[k] = {"median": round(v[len(v) // 2], 2), "min": round(v[0], 2), "max": round(v[-1], 2)}
        return out


def timed_sampled(card, fn, seconds):
    """Run fn() back to back for ~`seconds` under the sampler -> (calls, elapsed s, telemetry summary)."""
    with Sampler(card) as smp:
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            fn()
            n += 1
        el = time.perf_counter() - t0
    return n, el, smp.summary()
