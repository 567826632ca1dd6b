"""Clock / power / partition state of the GPU a process runs on, read from sysfs (amdgpu) -- measurement plumbing for
bench.py and tools/: a roofline fraction is read against the box it was taken on (MI355X_MICROARCH.md, "DVFS
give-back": the chip lowers its clock under load and devices differ).  Nothing here touches the HIP runtime beyond
asking which PCI device `device` is; every field is best-effort (None when the box does not expose it)."""
import glob
import os
import re


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _pci_bus_id(device=0):
    """'0000:bb:dd.f' of HIP device `device`, or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device)
        dom, bus, dev = getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        if bus is not None and dev is not None:
            return "%04x:%02x:%02x.0" % (dom or 0, bus, dev)
    except Exception:
        pass
    try:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) == 0:
            return buf.value.decode().lower()
    except Exception:
        pass
    return None


def sysfs_dir(device=0):
    """The amdgpu sysfs directory of HIP device `device` (by PCI address; the only amdgpu card if there is just one)."""
    bdf = _pci_bus_id(device)
    if bdf and os.path.isdir("/sys/bus/pci/devices/%s" % bdf):
        return "/sys/bus/pci/devices/%s" % bdf
    cards = [d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if _read(d + "/vendor") == "0x1002"]
    return os.path.realpath(cards[0]) if len(cards) == 1 else None


def _dpm_current(text):
    """pp_dpm_* lists its levels one per line, the current one marked '*': -> MHz of that level."""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            m = re.search(r"(\d+)\s*Mhz", line, re.I)
            if m:
                return int(m.group(1))
    return None


def read(device=0):
    """One sample: {sclk_mhz, mclk_mhz, fclk_mhz, socclk_mhz, power_w, power_cap_w, temp_c, hbm_temp_c, gpu_busy,
    compute_partition, memory_partition, perf_level, sysfs}."""
    d = sysfs_dir(device)
    out = {"sysfs": d}
    if not d:
        return out
    for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk"),
                      ("socclk_mhz", "pp_dpm_socclk")):
        out[key] = _dpm_current(_read(os.path.join(d, name)))
    out["compute_partition"] = _read(os.path.join(d, "current_compute_partition"))
    out["memory_partition"] = _read(os.path.join(d, "current_memory_partition"))
    out["perf_level"] = _read(os.path.join(d, "power_dpm_force_performance_level"))
    busy = _read(os.path.join(d, "gpu_busy_percent"))
    out["gpu_busy"] = int(busy) if busy and busy.isdigit() else None
    for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
        def num(name, scale):
            v = _read(os.path.join(hw, name))
            try:
                return round(int(v) * scale, 1)
            except (TypeError, ValueError):
                return None
        p = num("power1_average", 1e-6)
        out["power_w"] = p if p is not None else num("power1_input", 1e-6)
        out["power_cap_w"] = num("power1_cap", 1e-6)
        out["temp_c"] = num("temp1_input", 1e-3)          # edge / junction by ASIC
        t2, t3 = num("temp2_input", 1e-3), num("temp3_input", 1e-3)
        out["junction_temp_c"], out["hbm_temp_c"] = t2, t3
        f1 = num("freq1_input", 1e-6)
        if f1 is not None:
            out["freq1_mhz"] = f1
        break
    return out


def summarize(before, after):
    """The pair as the bench line carries it."""
    keys = ("sclk_mhz", "mclk_mhz", "fclk_mhz", "socclk_mhz", "power_w", "power_cap_w", "temp_c", "junction_temp_c",
            "hbm_temp_c", "freq1_mhz")
    rec = {k: [before.get(k), after.get(k)] for k in keys if before.get(k) is not None or after.get(k) is not None}
    for k in ("compute_partition", "memory_partition", "perf_level"):
        rec[k] = after.get(k) or before.get(k)
    rec["source"] = "amdgpu sysfs of this device, [before, after] the timed window" if after.get("sysfs") else \
        "amdgpu sysfs not readable on this box"
    return rec
