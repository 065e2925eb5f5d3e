"""disparity_to_point_cloud_amd/telemetry.py (the sysfs sampler bench.py and tools/devclass_probe.py put beside a measurement) and
d2pc_clock_probe_device.  CPU: the sampler on a fake sysfs tree and on a machine without any card; GPU: the clock probe."""
import os
import struct
import time

import numpy as np
import pytest

from disparity_to_point_cloud_amd import telemetry


def _fake_card(tmp_path, sclk_hz=2_390_000_000, power_uw=1_392_000_000):
    card = tmp_path / "card3" / "device"
    hw = card / "hwmon" / "hwmon7"
    hw.mkdir(parents=True)
    (card / "vendor").write_text("0x1002\n")
    (card / "current_compute_partition").write_text("SPX\n")
    (card / "current_memory_partition").write_text("NPS1\n")
    (card / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 2390Mhz *\n2: 2400Mhz\n")
    (card / "pp_dpm_fclk").write_text("0: 1250Mhz *\n")
    (card / "gpu_busy_percent").write_text("100\n")
    (hw / "freq1_input").write_text(f"{sclk_hz}\n")
    (hw / "freq2_input").write_text("2000000000\n")
    (hw / "power1_input").write_text(f"{power_uw}\n")
    (hw / "power1_cap").write_text("1400000000\n")
    (hw / "temp2_input").write_text("54000\n")
    # gpu_metrics v1.8 header + the six leading u16 fields (hotspot, mem, vrsoc, socket power, gfx activity, umc activity)
    (card / "gpu_metrics").write_bytes(struct.pack("<HBB6H", 3872, 1, 8, 54, 61, 40, 1392, 100, 42) + bytes(64))
    return str(card)


def test_sampler_reads_a_card_and_summarises(tmp_path):
    card = _fake_card(tmp_path)
    one = telemetry.sample(card)
    assert one["sclk_MHz"] == pytest.approx(2390.0) and one["mclk_MHz"] == pytest.approx(2000.0)
    assert one["power_input_W"] == pytest.approx(1392.0) and one["metrics_socket_power_W"] == 1392
    assert one["dpm_sclk_MHz"] == 2390 and one["dpm_fclk_MHz"] == 1250 and one["gpu_busy_percent"] == 100
    assert one["temp_junction_C"] == pytest.approx(54.0) and one["metrics_gfx_activity"] == 100
    st = telemetry.static_state(card)
    assert st["current_compute_partition"] == "SPX" and st["current_memory_partition"] == "NPS1" and st["power1_cap_W"] == 1400.0
    with telemetry.Sampler(card, 0.002) as smp:
        time.sleep(0.05)
    s = smp.summary()
    assert s["samples"] >= 3 and s["sclk_MHz"]["median"] == pytest.approx(2390.0) and s["sclk_MHz"]["min"] <= s["sclk_MHz"]["max"]


def test_no_card_no_telemetry_no_error():
    """a container without the computing card's sysfs node (or this one, without any GPU): everything answers empty"""
    assert telemetry.sample(None) == {} and telemetry.static_state(None) == {}
    with telemetry.Sampler(None, 0.002) as smp:
        time.sleep(0.01)
    assert smp.summary() == {"samples": 0}
    assert telemetry.find_card(pci_address="ffff:ff:1f.0") is None
    assert telemetry.parse_gpu_metrics(b"") is None and telemetry.parse_gpu_metrics(b"\x00" * 8) is None
    # a blob of another family: the header is reported, no field is guessed
    other = telemetry.parse_gpu_metrics(struct.pack("<HBB", 100, 2, 3) + bytes(100))
    assert other == {"gpu_metrics_size": 100, "gpu_metrics_format": 2, "gpu_metrics_revision": 3}


def test_dpm_table_parsing():
    assert telemetry._dpm_current("S: 95Mhz *\n0: 500Mhz\n1: 2400Mhz") == 95
    assert telemetry._dpm_current("0: 500Mhz\n1: 2400Mhz") is None and telemetry._dpm_current(None) is None


@pytest.mark.gpu
def test_clock_probe_reports_a_shader_clock():
    """d2pc_clock_probe_device: eight one-wave blocks sleep for min_us of the 100 MHz counter and report the shader cycles that
    passed -- on an idle device the clock they see lies between the deep-idle and the boost clock; bad arguments are refused."""
    import torch

    import disparity_to_point_cloud_amd as d2pc

    with d2pc.Context(q=d2pc.make_q()) as ctx:
        out = torch.zeros(16, dtype=torch.int64, device="cuda:0")
        s = torch.cuda.current_stream().cuda_stream
        ctx.clock_probe(out.data_ptr(), 3000, s)
        torch.cuda.synchronize()
        v = out.cpu().numpy().reshape(8, 2)
        assert (v[:, 1] >= 300_000).all() and (v[:, 1] < 5_000_000).all(), v          # >= 3 ms of 100-MHz ticks, bounded
        ghz = v[:, 0] / v[:, 1] * 0.1
        assert ((ghz > 0.05) & (ghz < 3.0)).all(), ghz
        for bad_ptr, bad_us in ((None, 100), (out.data_ptr() + 4, 100), (out.data_ptr(), 0), (out.data_ptr(), 3_000_000)):
            with pytest.raises(d2pc.D2pcError) as ei:
                ctx.clock_probe(bad_ptr, bad_us, s)
            assert ei.value.status == 1                                               # D2PC_ERR_INVALID_ARG
