"""CPU: the dependency-free TensorBoard event writer the Runner logs through when tensorboard is not installed
(reference: ObjTracker/run.py:127, jointopt.py:151-153 -- add_scalar per loss key per step under <exp>/board)."""
import glob
import math
import os

from dynhor_amd import tb_events


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC-32C (Castagnoli)
    assert tb_events._crc32c(b"\x00" * 32) == 0x8A9136AA
    assert tb_events._crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tb_events._crc32c(bytes(range(32))) == 0x46DD794E
    assert tb_events._crc32c(b"123456789") == 0xE3069283


def test_event_file_roundtrip(tmp_path):
    w = tb_events.EventFileWriter(str(tmp_path / "board"))
    vals = [("Loss/loss", 1.25, 100), ("Statistics/psnr", 23.5, 100), ("Loss/loss", 0.75, 200), ("lr", 5e-4, 300000)]
    for tag, v, step in vals:
        w.add_scalar(tag, v, step)
    w.close()
    files = glob.glob(str(tmp_path / "board" / "events.out.tfevents.*"))
    assert len(files) == 1 and os.path.getsize(files[0]) > 0
    got = tb_events.read_scalars(files[0])
    assert [(s, t) for s, t, _ in got] == [(s, t) for t, _, s in vals]
    assert all(math.isclose(g[2], v[1], rel_tol=1e-6) for g, v in zip(got, vals))
    raw = open(files[0], "rb").read()
    assert b"brain.Event:2" in raw[:64]                    # the file-version record TensorBoard requires first


def test_make_writer_falls_back_without_tensorboard(tmp_path):
    w = tb_events.make_writer(str(tmp_path / "b"))
    w.add_scalar("x", 1.0, 1)
    w.flush(); w.close()
    assert glob.glob(str(tmp_path / "b" / "events.out.tfevents.*"))


def test_roctx_marker_library_loads_and_ranges_nest():
    """the stage calls can be wrapped in roctx ranges (Runner conf train.roctx).  Here: the marker
    library loads, push / pop nest, and the StageTimer brackets a stage call with exactly one range also when the call raises."""
    from dynhor_amd import _lib
    from dynhor_amd.renderer import StageTimer
    R = _lib.roctx()
    if R is None:
        import pytest
        pytest.skip("no roctx library in this image")
    d0 = R.roctxRangePushA(b"outer")
    d1 = R.roctxRangePushA(b"inner")
    assert d1 == d0 + 1
    assert R.roctxRangePop() == d1 and R.roctxRangePop() == d0
    t = StageTimer()
    assert t.set_markers(True) is True
    calls = []
    t("stage_ok", lambda a, b: calls.append((a, b)) or 0, 1, 2)
    assert calls == [(1, 2)]
    try:
        t("stage_bad", lambda: -3)                    # a non-zero status raises DynhorHipError; the range must still be popped
    except _lib.DynhorHipError:
        pass
    assert R.roctxRangePushA(b"probe") == d0 and R.roctxRangePop() == d0
