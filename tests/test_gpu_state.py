"""Measurement plumbing of round 4: gpu_state (amdgpu sysfs sample; must degrade to None fields without a GPU) and
ammsb_clock_probe (the shader clock an XCD holds, from s_memtime / s_memrealtime of idle probe waves)."""
import numpy as np
import pytest


def _mods():
    import __graft_entry__ as ge
    ge.build()
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import gpu_state
    return gpu_state


def test_gpu_state_degrades_without_a_device():
    gs = _mods()
    s = gs.read(0)
    assert isinstance(s, dict) and "sysfs" in s
    rec = gs.summarize(s, s)
    assert "source" in rec
    for k, v in rec.items():  # [before, after] pairs or plain strings / None: always JSON-able
        assert v is None or isinstance(v, (str, list))
    assert gs._dpm_current("0: 500Mhz\n1: 2100Mhz *\n2: 2400Mhz") == 2100
    assert gs._dpm_current("S: 95Mhz *\n0: 500Mhz") == 95 and gs._dpm_current("") is None


@pytest.mark.gpu
def test_clock_probe_reads_a_plausible_shader_clock():
    import torch
    gs = _mods()
    from mcmc_ammsb_gpu_amd import ops
    ctx = ops.Context(ops.make_params(1000, 32))
    probe = ops.ClockProbe(ctx, 64)
    probe.launch(300)
    r = probe.read()
    assert r["blocks"] == 64 and 300.0 < r["mhz"] < 2600.0, r          # MI355X: 500 .. 2400 MHz levels
    xcds = [x for x in r["mhz_per_xcd"] if x is not None]
    assert len(xcds) == 8 and all(300.0 < x < 2600.0 for x in xcds), r  # 64 blocks reach all eight XCDs
    st = gs.read(torch.cuda.current_device())
    assert st["sysfs"] and st["sclk_mhz"] and st["power_cap_w"]         # the box exposes its clocks and its power cap
    assert st["compute_partition"] and st["memory_partition"]
