#!/usr/bin/env python3
"""List VGPR/SGPR/LDS/spill figures of every kernel from a --save-temps gfx950 .s file."""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda key: (re.search(r"\." + key + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if pat and not re.search(pat, name):
        continue
    print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}")
