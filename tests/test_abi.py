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
    assert L.lcd_abi_version() == 7


def test_shard_driver_builds_and_exports_every_declared_symbol():
    import ctypes
    from rtabmap_amd import build as b
    import rtabmap_amd
    rtabmap_amd.load()
    L = ctypes.CDLL(b.build_shard())
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "lcd_shard.h")).read()
    declared = set(re.findall(r"\b(lcd_shard_[a-z0-9_]+)\s*\(", header)) - {"lcd_shard_knn2_dev", "lcd_shard_frame_dev"}
    assert declared == {"lcd_shard_unique_id", "lcd_shard_comm_create", "lcd_shard_comm_create_transport", "lcd_shard_comm_destroy",
                        "lcd_shard_last_error", "lcd_shard_set_growth", "lcd_shard_owner_of", "lcd_shard_set_append", "lcd_shard_frame", "lcd_shard_frame_deferred",
                        "lcd_shard_flush", "lcd_shard_sig_remove"}
    for s in declared:
        assert hasattr(L, s), s


def test_p2p_library_builds_and_exports_every_declared_symbol():
    """liblcd_p2p.so (include/lcd_p2p.h): compiled for gfx950 here, every declared entry exported, the transport it hands out has the
    layout lcd_shard_comm_create_transport checks, and without a device nothing is created (no host-staged fallback inside)."""
    import ctypes as C
    import torch
    from rtabmap_amd import build as b
    L = C.CDLL(b.build_p2p())
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "lcd_p2p.h")).read()
    declared = set(re.findall(r"\b(lcd_p2p_[a-z0-9_]+)\s*\(", header))
    assert declared == {"lcd_p2p_create", "lcd_p2p_export", "lcd_p2p_connect", "lcd_p2p_destroy", "lcd_p2p_last_error", "lcd_p2p_set_wire",
                        "lcd_p2p_set_timeout_ms", "lcd_p2p_set_conservative_fences", "lcd_p2p_status", "lcd_p2p_clear_status", "lcd_p2p_all_gather", "lcd_p2p_all_reduce_sum_i64",
                        "lcd_p2p_transport"}
    for s in declared:
        assert hasattr(L, s), s
    L.lcd_p2p_create.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    assert L.lcd_p2p_create(2, 2, 1024, 1024, C.byref(h)) == 1 and not h.value          # rank outside the world: LCD_ERR_INVALID
    assert L.lcd_p2p_create(0, 17, 1024, 1024, C.byref(h)) == 1 and not h.value
    if not torch.cuda.is_available():
        assert L.lcd_p2p_create(0, 1, 1024, 1024, C.byref(h)) != 0 and not h.value      # no device: an error, not a fallback


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


def test_pipelined_filter_plan_covers_every_size():
    """The launch plan of a pipelined frame's filter (knn_bf16_plan_pipelined, host code: no device needed), over vocabulary sizes
    from 256 to 1.3M rows and frames of 1 .. 4096 descriptors: every row belongs to exactly one strip, a strip holds 1 .. 8 tiles
    and is never empty; for frames of up to 512 descriptors one-strip launches are resident at once, two workgroups to a compute
    unit with the distance tiles keeping theirs, and persistent launches leave the two tail workgroups and the tiles a unit each;
    the plan is the measured one at the headline size."""
    import ctypes as C
    import rtabmap_amd
    rtabmap_amd.load()
    lib = C.CDLL(rtabmap_amd.library_path())
    lib.lcd_debug_frame_plan.restype = C.c_int
    lib.lcd_debug_frame_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    out = (C.c_int * 5)()
    sizes = sorted(set(list(range(256, 4000, 97)) + list(range(4000, 130000, 1777)) + list(range(130000, 1300000, 41011)) +
                       [49000, 56320, 56321, 59000, 112640, 112641, 125000, 1000000]))
    for q in (1, 33, 64, 500, 512, 513, 1000, 1024, 4096):
        for together in (0, 1):
            for n in sizes:
                assert lib.lcd_debug_frame_plan(q, n, together, out) == 0
                tpb, nb, px, tiles, qchunks = list(out)
                n_tiles = (n + 31) // 32
                assert 1 <= tpb <= 8 and nb >= 1
                assert nb * tpb >= n_tiles > (nb - 1) * tpb, (q, n, tpb, nb)
                if px == 0 and q <= 512:                       # (larger frames bring more distance tiles than the chip has units anyway)
                    assert nb + tiles + 2 + 8 <= 512, (q, n, together, tpb, nb)      # everything resident at once, two to a unit
                    assert nb <= 2 * (256 - tiles), (q, n, together, tpb, nb)        # the tiles keep their units
                if px > 0 and q <= 512:
                    assert px + tiles + 2 <= 256, (q, n, together, px)               # a persistent workgroup owns its unit
    assert lib.lcd_debug_frame_plan(500, 49000, 1, out) == 0 and list(out)[:3] == [7, 219, 0]
    assert lib.lcd_debug_frame_plan(500, 59000, 1, out) == 0 and list(out)[:3] == [5, 369, 0]
    assert lib.lcd_debug_frame_plan(500, 125000, 1, out) == 0 and out[2] > 0


def test_every_option_key_the_library_accepts_is_documented_in_the_headers():
    """lcd_set_option's keys are part of the boundary: each one the engine compares against must be named in include/*.h"""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    keys = set()
    for path in glob.glob(os.path.join(root, "rtabmap_amd", "csrc", "*.hip")):
        keys |= set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', open(path).read()))
    assert len(keys) >= 8
    headers = "".join(open(p).read() for p in glob.glob(os.path.join(root, "include", "*.h")))
    missing = sorted(k for k in keys if '"%s"' % k not in headers)
    assert not missing, "option keys without a word in include/*.h: %s" % missing
