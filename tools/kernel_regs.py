#!/usr/bin/env python3
"""VGPRs / scratch / LDS of the kernels in a hipcc -save-temps assembly file (the amdhsa metadata).

  hipcc ... -save-temps -c lh_kernels_f64_richards.hip && tools/kernel_regs.py <file>.s [SUBSTRING]
"""
import re
import sys

s = open(sys.argv[1], errors="ignore").read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
meta = s[s.rindex("amdhsa.kernels"):]
for blk in meta.split("  - .agpr_count")[1:]:
    nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if sub not in nm:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print("vgpr=%-4s sgpr=%-4s scratch=%-4s lds=%-6s %s" % (g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"),
                                                       g("group_segment_fixed_size"), nm[:150]))
