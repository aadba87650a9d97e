"""Register / LDS use of the conv3x3_patch_kernel instances from hipcc -Rpass-analysis=kernel-resource-usage output.
Usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c conv3x3_patch.hip 2> res.txt; python tools/resusage.py res.txt [f8]"""
import re, sys
txt = open(sys.argv[1]).read()
only_f8 = len(sys.argv) > 2
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0]
    m = re.search(r'conv3x3_patch_kernelI(\w+?)Li(\d+)ELi(\d)ELi(\d+)ELi(\d)ELb(\d)ELi(\d)ELb(\d)', name)
    if not m or (only_f8 and m.group(5) == '0'):
        continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print("T=%s BN=%s OCC=%s SUB=%s F8=%s PRE=%s TAPS=%s DMAP=%s" % m.groups(), 'vgpr', g(' VGPRs'), 'agpr', g('AGPRs'),
          'spill', g('VGPRs Spill'), 'scratch', g(r'ScratchSize \[bytes/lane\]'), 'lds', g(r'LDS Size \[bytes/block\]'))
