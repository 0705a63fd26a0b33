"""Per-kernel resource table of libmic_hip.so from the compiler's own listing (hipcc -Rpass-analysis=kernel-resource-usage, which the
Makefile leaves in csrc/build/<unit>.res): VGPRs / AGPRs / SGPRs, scratch, spills, occupancy (waves per SIMD), static LDS.

  python tools/kernel_resources.py            table of the current build beside the committed one, differences marked
  python tools/kernel_resources.py --update   rewrite tests/golden/kernel_resources.json from the current build

The guard (tests/test_kernel_resources_cpu.py) fails on scratch or spills in any kernel, on an occupancy below the committed one, and
on kernels the table does not know — the two silent regressions of round 4 (272 B of scratch in the non-PLAIN 256x256 epilogues, a
waterfall loop around every LDS-DMA request) both showed in exactly these numbers first."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "multilingual-image-captioning_amd", "csrc", "build")
TABLE = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")
FIELDS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
          "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}


def parse_res(path):
    """{mangled kernel name: {field: int}} of one .res file"""
    out, cur = {}, None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+) \[-Rpass-analysis", line)
        if m and cur is not None and m.group(1) in FIELDS:
            try:
                cur[FIELDS[m.group(1)]] = int(m.group(2))
            except ValueError:
                pass
    return out


def demangle(names):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, timeout=60)
            d = r.stdout.split("\n")
            if r.returncode == 0 and len(d) >= len(names):
                return dict(zip(names, d))
        except (OSError, subprocess.SubprocessError):
            pass
    return {n: n for n in names}  # (mangled names: still unique keys, the table then has to come from the same machine)


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)  # template arguments stay, the parameter list goes


def current():
    """{unit: {kernel (demangled, short): fields}} of the build directory"""
    units = {}
    for f in sorted(os.listdir(BUILD)) if os.path.isdir(BUILD) else []:
        if f.endswith(".res"):
            k = parse_res(os.path.join(BUILD, f))
            dm = demangle(list(k))
            units[f[:-4]] = {short(dm[n]): v for n, v in k.items()}
    return units


def compare(cur, ref):
    """list of (severity, message); severity "fail" = what the guard refuses"""
    msgs = []
    for unit, ks in cur.items():
        for name, v in ks.items():
            tag = f"{unit}: {name}"
            if v.get("scratch", 0) > 0 or v.get("vgpr_spill", 0) > 0:
                msgs.append(("fail", f"{tag}: scratch {v.get('scratch')} B/lane, {v.get('vgpr_spill')} VGPRs spilled"))
            r = ref.get(unit, {}).get(name)
            if r is None:
                msgs.append(("fail", f"{tag}: not in the committed table (python tools/kernel_resources.py --update)"))
                continue
            if v.get("occupancy", 0) < r.get("occupancy", 0):
                msgs.append(("fail", f"{tag}: occupancy {r['occupancy']} -> {v['occupancy']} waves/SIMD (vgpr {r.get('vgpr')}+{r.get('agpr')} -> {v.get('vgpr')}+{v.get('agpr')})"))
            if v.get("sgpr_spill", 0) > r.get("sgpr_spill", 0):  # (SGPRs spill into VGPR lanes, not memory: cheap, but a trend to see)
                msgs.append(("fail", f"{tag}: SGPR spills {r.get('sgpr_spill', 0)} -> {v['sgpr_spill']}"))
            for f in ("vgpr", "agpr", "sgpr", "lds", "occupancy"):
                if v.get(f) != r.get(f):
                    msgs.append(("note", f"{tag}: {f} {r.get(f)} -> {v.get(f)}"))
    for unit, ks in ref.items():
        for name in ks:
            if name not in cur.get(unit, {}):
                msgs.append(("note", f"{unit}: {name}: in the committed table, not in this build"))
    return msgs


def main():
    cur = current()
    if not cur:
        sys.exit(f"no .res files under {BUILD}: build first (make -C multilingual-image-captioning_amd/csrc)")
    if "--update" in sys.argv:
        os.makedirs(os.path.dirname(TABLE), exist_ok=True)
        json.dump(cur, open(TABLE, "w"), indent=0, sort_keys=True)
        print(f"wrote {TABLE}: {sum(len(v) for v in cur.values())} kernels in {len(cur)} units")
        return
    ref = json.load(open(TABLE)) if os.path.exists(TABLE) else {}
    print(f"{'unit':<12} {'kernel':<88} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'scr':>4} {'occ':>3} {'lds':>6}")
    for unit, ks in cur.items():
        for name, v in sorted(ks.items()):
            print(f"{unit:<12} {name[:88]:<88} {v.get('vgpr', 0):>4} {v.get('agpr', 0):>4} {v.get('sgpr', 0):>4} {v.get('scratch', 0):>4} {v.get('occupancy', 0):>3} {v.get('lds', 0):>6}")
    for sev, m in compare(cur, ref):
        print(f"[{sev}] {m}")


if __name__ == "__main__":
    main()
