"""Regression guard for the HIP kernels (no GPU needed): the compiler's own per-kernel resource listing of the current build
(csrc/build/*.res, written by every `make`) against the committed table tests/golden/kernel_resources.json.  Fails on scratch memory
or spilled VGPRs in ANY kernel, on an occupancy (waves per SIMD) below the committed one, on more spilled SGPRs, and on kernels the
table does not know (`python tools/kernel_resources.py --update` after a deliberate change — the diff of the table is then part of the
commit).  Round 4 shipped two regressions that these numbers show at compile time: 272 B of scratch in the non-PLAIN 256x256 GEMM
epilogues, and a waterfall loop (extra SGPRs / VGPRs) around every LDS-DMA request of the persistent four-wave kernel."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_kernels_keep_their_registers_scratch_and_occupancy():
    import json

    import kernel_resources as KR

    csrc = os.path.join(ROOT, "multilingual-image-captioning_amd", "csrc")
    # an object without its listing (a tree built before the Makefile wrote them) would leave make with nothing to do: drop it
    bdir = os.path.join(csrc, "build")
    for f in (os.listdir(bdir) if os.path.isdir(bdir) else []):
        if f.endswith(".o") and not os.path.exists(os.path.join(bdir, f[:-2] + ".res")):
            os.remove(os.path.join(bdir, f))
    subprocess.run(["make", "-C", csrc, "-j8"], check=True, capture_output=True)  # incremental: nothing to do after build()
    cur = KR.current()
    assert cur, "no resource listings under csrc/build"
    ref = json.load(open(KR.TABLE))
    msgs = KR.compare(cur, ref)
    fails = [m for sev, m in msgs if sev == "fail"]
    assert not fails, "\n".join(fails)
    n = sum(len(v) for v in cur.values())
    assert n >= 150, n
    # the kernels the step time hangs on: whole register file, accumulators in AGPRs, nothing in scratch
    w4 = [v for k, v in cur["gemm_w4"].items() if "gemm_w4_kernel" in k]
    assert len(w4) >= 5 and all(v["agpr"] == 256 and v["scratch"] == 0 and v["occupancy"] == 1 for v in w4), w4


def test_guard_catches_scratch_and_occupancy_drops():
    import kernel_resources as KR

    ref = {"u": {"k<1>": dict(vgpr=128, agpr=0, sgpr=40, scratch=0, occupancy=4, sgpr_spill=0, vgpr_spill=0, lds=0)}}
    ok = {"u": {"k<1>": dict(ref["u"]["k<1>"], vgpr=120)}}
    assert not [m for s, m in KR.compare(ok, ref) if s == "fail"]
    for bad in (dict(scratch=272), dict(vgpr_spill=3), dict(occupancy=3), dict(sgpr_spill=8)):
        cur = {"u": {"k<1>": dict(ref["u"]["k<1>"], **bad)}}
        assert [m for s, m in KR.compare(cur, ref) if s == "fail"], bad
    assert [m for s, m in KR.compare({"u": {"new<2>": ref["u"]["k<1>"]}}, ref) if s == "fail"]  # unknown kernel
