"""tools/isa_loop_histogram.py on a hand-made gfx950 listing: a loop is the span from a label to the LAST backward branch to it; its
instructions are counted by class; only loops with matrix-core instructions are reported unless --all is given."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LISTING = """
_ZN3lcd9my_kernelEv:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\ts_waitcnt lgkmcnt(0)
.LBB0_1:
\t.loc\t1 10 3
\tds_read_b128 v[0:3], v9
\ts_waitcnt lgkmcnt(0)
\tv_mfma_f32_32x32x16_f16 v[10:25], v[0:1], v[2:3], v[10:25]
\tv_and_or_b32 v30, v10, v31, s2
\tv_med3_i32 v32, v33, v34, v30
\tv_min_i32_e32 v33, v33, v30
\ts_add_i32 s3, s3, 1
\ts_cmp_lt_i32 s3, s6
\ts_cbranch_scc1 .LBB0_1
.LBB0_2:
\tglobal_load_dword v1, v[2:3], off
\tv_add_u32_e32 v4, v4, v1
\ts_add_i32 s7, s7, 1
\ts_cmp_lt_i32 s7, s8
\ts_cbranch_scc1 .LBB0_2
\ts_endpgm
.Lfunc_end0:
_ZN3lcd5otherEv:
.LBB1_1:
\tv_mfma_f32_32x32x16_f16 v[10:25], v[0:1], v[2:3], v[10:25]
\ts_cbranch_scc1 .LBB1_1
"""


def _run(tmp_path, *extra):
    p = tmp_path / "k.s"
    p.write_text(LISTING)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loop_histogram.py"), str(p), "my_kernel", "--min", "1"] + list(extra),
                          capture_output=True, text=True, check=True).stdout


def test_loops_are_found_and_counted(tmp_path):
    out = _run(tmp_path)
    loops = [l for l in out.splitlines() if l.startswith("loop ")]
    assert len(loops) == 1 and ".LBB0_1" in loops[0], out                      # the second loop has no matrix-core instruction; `other` is another kernel
    assert "9 instructions | mfma 1 valu 3 salu 2 lds 1 vmem 0 wait 1 barrier 0 branch 1" in loops[0], out
    assert "VALU issue >= 12 cycles per trip" in loops[0]
    out = _run(tmp_path, "--all", "--ops")
    loops = [l for l in out.splitlines() if l.startswith("loop ")]
    assert len(loops) == 2 and ".LBB0_2" in loops[1] and "vmem 1" in loops[1] and "mfma 0" in loops[1], out
    assert any(l.split() == ["1", "v_med3_i32"] for l in out.splitlines()), out
