"""CPU-only: the C-ABI library builds for gfx950, loads, exports every symbol include/lcd.h declares, and refuses to
run without a GPU instead of falling back to anything."""
import os
import re

import pytest


def test_library_builds_and_exports_every_declared_symbol():
    import rtabmap_amd
    from rtabmap_amd import capi
    L = rtabmap_amd.load()
    assert os.path.exists(rtabmap_amd.library_path())
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "lcd.h")).read()
    declared = set(re.findall(r"\b(lcd_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert L.lcd_abi_version() == 3


def test_shard_driver_builds_and_exports_every_declared_symbol():
    import ctypes
    from rtabmap_amd import build as b
    import rtabmap_amd
    rtabmap_amd.load()
    L = ctypes.CDLL(b.build_shard())
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "lcd_shard.h")).read()
    declared = set(re.findall(r"\b(lcd_shard_[a-z0-9_]+)\s*\(", header)) - {"lcd_shard_knn2_dev", "lcd_shard_frame_dev"}
    assert declared == {"lcd_shard_unique_id", "lcd_shard_comm_create", "lcd_shard_comm_destroy", "lcd_shard_last_error", "lcd_shard_frame"}
    for s in declared:
        assert hasattr(L, s), s


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import rtabmap_amd
    with pytest.raises(rtabmap_amd.LcdError):
        rtabmap_amd.Engine("f32", 64)


def test_product_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(__file__), "..", "rtabmap_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "liblcd_oracle" not in src, f
