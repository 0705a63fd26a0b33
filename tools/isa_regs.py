"""Register / LDS / scratch use per kernel of one translation unit's gfx950 assembly (hipcc --save-temps=obj output).
usage: python tools/isa_regs.py /tmp/isa/gemm_t64-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in s.split("- .agpr_count:")[1:]:
    ag = b.split("\n")[0].strip()
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    vg = re.search(r"\.vgpr_count:\s+(\d+)", b).group(1)
    sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1)
    sc = re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt in dem:
        print(f"{dem[:110]:110s} vgpr {vg:>3s} agpr {ag:>3s} spill {sp} scratch {sc}")
