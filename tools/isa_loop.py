"""Skeleton (loads, waits, barriers, LDS traffic, MFMAs, branches) of one kernel of a --save-temps .s file.
usage: python tools/isa_loop.py file.s mangled_kernel_name [max_lines]"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ":")
j = s.index("s_endpgm", i)
keep = [l for l in s[i:j].split("\n") if re.search(r"buffer_load|global_load|waitcnt|s_barrier|s_cbranch|^\.LBB|mfma|ds_write_b128|ds_read_b128|scratch_|v_mov_b32.*;", l)]
out, prev, cnt = [], None, 0
for l in keep:
    k = re.sub(r"\s+", " ", l.strip())
    k = re.sub(r"\b[vas]\[?\d+(:\d+)?\]?", "R", k)
    k = re.sub(r"offset:\d+", "", k)
    if k == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{cnt}x {prev}")
        prev, cnt = k, 1
out.append(f"{cnt}x {prev}")
print("\n".join(out[: int(sys.argv[3]) if len(sys.argv) > 3 else 200]))
