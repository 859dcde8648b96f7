"""tools/store_wait_audit.py on a hand-made gfx950 listing: a wait that is reached with a store among the outstanding vector-memory
operations is reported with the source lines of the wait and of the store; a wait that leaves the store in flight is not."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LISTING = """
\t.file\t1 "/x" "body.cuh"
_ZN3lcd9my_kernelEv:
\t.loc\t1 10 3
\tglobal_load_dword v1, v[2:3], off
\t.loc\t1 11 3
\tglobal_store_dword v[4:5], v6, off
\t.loc\t1 12 3
\tglobal_load_dword v7, v[8:9], off
\t.loc\t1 13 3
\ts_waitcnt vmcnt(1)
\t.loc\t1 14 3
\ts_waitcnt vmcnt(0)
\t.loc\t1 15 3
\tglobal_load_dword v1, v[2:3], off
\t.loc\t1 16 3
\ts_waitcnt vmcnt(0)
\t.loc\t1 17 3
\tglobal_atomic_add v1, v2, s[0:1]
\t.loc\t1 18 3
\ts_waitcnt vmcnt(0)
\t.loc\t1 19 3
\tglobal_atomic_add v3, v1, v2, s[0:1] sc0
\t.loc\t1 20 3
\ts_waitcnt vmcnt(0)
.Lfunc_end0:
_ZN3lcd5otherEv:
\tglobal_store_dword v[4:5], v6, off
\ts_waitcnt vmcnt(0)
"""


def test_waits_behind_stores_are_found(tmp_path):
    p = tmp_path / "k.s"
    p.write_text(LISTING)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "store_wait_audit.py"), str(p), "my_kernel"], capture_output=True, text=True, check=True).stdout
    lines = [l for l in out.splitlines() if "<-" in l]
    # vmcnt(1) at :13 completes the load of :10 AND the store of :11 (in-order counter, one operation left in flight): reported;
    # vmcnt(0) at :14 only has the load of :12 left: not reported; the load / wait pair :15 / :16: not reported;
    # the non-returning atomic of :17 counts as a store (:18 reported), the returning one (sc0) of :19 as a load (:20 not)
    assert len(lines) == 2, out
    assert "[body.cuh:13]" in lines[0] and "[body.cuh:11]" in lines[0]
    assert "[body.cuh:18]" in lines[1] and "[body.cuh:17]" in lines[1]
    assert out.strip().endswith("2 waits behind stores in functions matching 'my_kernel'")
