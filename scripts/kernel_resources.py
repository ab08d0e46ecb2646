"""Register / scratch / occupancy table of the search kernels, from hipcc's kernel-resource-usage remarks:
   hipcc ... -Rpass-analysis=kernel-resource-usage -c roar_search_lds_ip.hip -o /dev/null 2> res.txt; python3 scripts/kernel_resources.py res.txt"""
import re, sys
t = open(sys.argv[1]).read()
for b in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = b.split()[0]
    m = re.search(r'search_v2_kernelILi(\d+)ELb(\d)ELi(\d+)ELb(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)(?:ELb(\d))?E', name)
    if not m:
        continue
    mode, ip, nch, tail, w, rp, prof, hub, gated, stats = m.groups()
    if len(sys.argv) > 2 and nch not in sys.argv[2].split(','):
        continue
    def g(k):
        r = re.search(re.escape(k) + r': (\d+)', b)
        return r.group(1) if r else '?'
    print("mode", mode, "NCH", nch, "W", w, "RP", rp, "PROF", prof, "HUB", hub, "GATED", gated, "STATS", stats, ": VGPR", g('VGPRs'), "SGPR", g('SGPRs'),
          "scratch", g('ScratchSize [bytes/lane]'), "occ", g('Occupancy [waves/SIMD]'), "sgpr spill", g('SGPRs Spill'), "vgpr spill", g('VGPRs Spill'))
