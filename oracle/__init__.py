"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(rtabmap_amd) never does.  See oracle/lcd_oracle.cpp for what is restated and from which reference file:line.
"""
from .oracle import *  # noqa: F401,F403
