"""bench.py reads the committed PMC summaries (profiles/r*_pmc_hbm_traffic.json) for `roofline.traffic`: every such file must carry
the keys it reads (a missing key there would end the driver's bench run with a KeyError after the timed steps)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_pmc_summaries_carry_the_keys_bench_reads():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    assert files
    for f in files:
        d = json.load(open(f))
        name = os.path.basename(f)
        if "generate" in name:
            att = d.get("attention", d)
            assert att["bytes_per_launch"] > 0, name
            if "gemm" in d:
                assert d["gemm"]["bytes_per_launch"] > 0, name
        else:
            assert d["bytes_per_launch"] > 0, name
        assert isinstance(d.get("summary", ""), str) and isinstance(d.get("commit", ""), str)


def test_bench_parses_and_defaults_to_one_gpu():
    import ast

    src = open(os.path.join(ROOT, "bench.py")).read()
    ast.parse(src)
    assert 'ap.add_argument("--gpus", type=int, default=1)' in src and '"--steps", type=int, default=10' in src
