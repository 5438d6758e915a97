#!/usr/bin/env python
"""bench.py's committed SQ constants carry the hash of the correspondence kernel's sources they were measured on (PMC_PROFILED_HASH): run this after the SQ passes
of tools/gpu_round.sh have been repeated on the present sources (or after an edit that does not change the kernel's code, e.g. a comment)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
h = bench.pmc_sources_hash()
p = os.path.join(ROOT, "bench.py")
s = open(p).read()
s2 = re.sub(r'PMC_PROFILED_HASH = "[0-9a-f]+"', 'PMC_PROFILED_HASH = "%s"' % h, s)
open(p, "w").write(s2)
print(h, "(changed)" if s2 != s else "(unchanged)")
