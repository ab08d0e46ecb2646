"""Registers / scratch / occupancy of every instantiation in one search translation unit, from the compiler's own report:
  python scripts/kernel_resources.py [roar_search_lds_ip] [extra hipcc flags ...]  -> one line per kernel (demangled template arguments)"""
import os, re, subprocess, sys
tu = sys.argv[1] if len(sys.argv) > 1 else "roar_search_lds_ip"
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mysteryann_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage",
       "-c", os.path.join(csrc, tu + ".hip"), "-o", "/dev/null"] + sys.argv[2:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    m = re.search(r"(\w+)<(.*)>\(", dem)
    short = f"{m.group(1)}<{m.group(2)}>" if m else dem[:80]
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    scr, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{short:110s} VGPRs {g('VGPRs'):4d} SGPRs {g('SGPRs'):4d} scratch {scr:4d} B/lane  waves/SIMD {occ}  LDS {lds}")
