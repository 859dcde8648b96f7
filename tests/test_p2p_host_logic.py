"""CPU-only: the host arithmetic of liblcd_p2p.so (no device): the all-reduce's slices partition [0, count) for every world size and count --
disjoint, in rank order, each starting on a multiple of 4 elements (16 bytes of the float wire, 32 of the integer wire), together covering
every element exactly once; and the export record is the size the header promises."""
import ctypes as C
import os
import re


def _lib():
    from rtabmap_amd import build as b
    L = C.CDLL(b.build_p2p())
    L.lcd_p2p_debug_chunk.restype = C.c_size_t
    L.lcd_p2p_debug_chunk.argtypes = [C.c_size_t, C.c_int]
    return L


def test_all_reduce_slices_partition_every_count():
    L = _lib()
    counts = list(range(0, 70)) + [99, 100, 101, 255, 256, 257, 1000, 1001, 4095, 4096, 4097, 100001, 1000001, (1 << 20) + 5, (1 << 24) + 3]
    for world in range(1, 17):
        for count in counts:
            chunk = L.lcd_p2p_debug_chunk(count, world)
            assert chunk % 4 == 0 and chunk * world >= count
            covered = 0
            for r in range(world):
                lo, hi = min(r * chunk, count), min((r + 1) * chunk, count)
                assert lo % 4 == 0 or lo == count
                assert lo == covered or lo == count          # in rank order, no gap
                covered = max(covered, hi)
            assert covered == count
            # no rank's share exceeds the even share by more than the rounding
            assert chunk <= (count + world - 1) // world + 3


def test_header_constants_match_the_library():
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "lcd_p2p.h")).read()
    assert int(re.search(r"#define LCD_P2P_HANDLE_BYTES (\d+)", header).group(1)) == 128     # sizeof(Export) is static_assert'ed against it
    assert int(re.search(r"#define LCD_P2P_MAX_WORLD (\d+)", header).group(1)) == 16
    src = open(os.path.join(os.path.dirname(__file__), "..", "rtabmap_amd", "csrc", "p2p_exchange.hip")).read()
    assert "static_assert(sizeof(Export) == LCD_P2P_HANDLE_BYTES" in src
    # the flag lines of the three kinds of exchange and the mailbox do not overlap for the largest world
    assert 3 * 64 * 16 <= 4096
