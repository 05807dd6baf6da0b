"""bench_board.BoardSampler on fake hwmon files (no GPU): the summary the bench line carries, and None when nothing can be read."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_board  # noqa: E402


def test_sampler_reads_clock_and_power_files(tmp_path):
    clk, pw = tmp_path / "freq1_input", tmp_path / "power1_input"
    clk.write_text("1950000000\n")
    pw.write_text("1330000000\n")
    s = bench_board.BoardSampler.__new__(bench_board.BoardSampler)
    s.sens = {"pci": "0000:00:00.0", "sclk": str(clk), "power": str(pw), "cap_w": 1400.0}
    s.period, s.mhz, s.w = 0.005, [], []
    import threading
    s._stop, s._th = threading.Event(), None
    with s:
        time.sleep(0.05)
        clk.write_text("1750000000\n")
        pw.write_text("1400000000\n")
        time.sleep(0.05)
    out = s.summary()
    assert out["samples"] >= 4 and out["power_cap_w"] == 1400.0
    assert 1750.0 <= out["sclk_mhz_min"] <= out["sclk_mhz_mean"] <= out["sclk_mhz_max"] == 1950.0
    assert abs(out["sclk_fraction_of_nominal"] - out["sclk_mhz_mean"] / 2400.0) < 1e-12
    assert out["power_w_max"] == 1400.0 and 0.0 < out["share_of_samples_within_3pct_of_cap"] < 1.0


def test_no_sensors_gives_none():
    s = bench_board.BoardSampler.__new__(bench_board.BoardSampler)
    import threading
    s.sens, s.period, s.mhz, s.w, s._stop, s._th = {"error": "no hwmon"}, 0.01, [], [], threading.Event(), None
    with s:
        time.sleep(0.02)
    assert s.summary() is None


def test_find_sensors_without_a_gpu_reports_instead_of_raising():
    out = bench_board.find_sensors(0)
    assert isinstance(out, dict) and ("error" in out or "pci" in out)
