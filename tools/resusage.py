"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stderr saved to a file)."""
import re
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
for b in blocks:
    name = b.split('\n')[0][:72]

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    print("%-72s vgpr=%s agpr=%s spill=%s scratch=%s occ=%s lds=%s" % (
        name, g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'),
        g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
