"""`python bench.py --gpus N` must start its own ranks (the driver's N=1-style command line with N > 1) and print one JSON
line.  A gpurun box has one GPU: MIC_BENCH_SHARE_GPU0=1 puts both ranks on cuda:0 with gloo as the gradient transport, so the
launcher, the rank plumbing, the bucketed reducer, the sparse embedding exchange and the JSON contract are all exercised
(the RCCL transport itself needs two devices)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=600):
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--small", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra,
                       env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    return r


def test_bench_self_launches_two_ranks(dev):
    r = _run(["--gpus", "2"], {"MIC_BENCH_SHARE_GPU0": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "images/sec"
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2" and "functional check" in d["config"]["workload"]
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["beam4_generate"]["n_gpus"] == 2
    assert abs(d["final_loss"]) < 100


def test_bench_emulated_comm_leg(dev):
    """`--emulate-comm N`: the 1-GPU step with every bucket's exchange replaced by a kernel holding the collective stream's CUs for
    the projected all-reduce time; reported beside the headline, labelled as what it is"""
    r = _run(["--emulate-comm", "2,8", "--no-generate", "--no-roofline"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    e = d["comm_emulated"]
    assert "NOT a scaling result" in e["note"] and set(e["worlds"]) == {"2", "8"}
    for w in e["worlds"].values():
        assert w["ms_per_step"] > 0 and w["comm_cus"] == 32 and w["comm_dtype"] == "fp32"  # (a reduced model's exchange hides at any N)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak"


def test_bench_refuses_more_ranks_than_gpus(dev):
    import torch

    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(n), "--no-generate"])
    assert r.returncode != 0 and "MIC_BENCH_SHARE_GPU0" in (r.stderr + r.stdout)


def test_bench_single_gpu_line_has_both_rooflines_and_dense_variant(dev):
    r = _run(["--dense-captions"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and "dense" in d["data"] and d["roofline"]["bound"] == "mfma"
    g = d["beam4_generate"]
    # the leg's roofline names the dominant kernel class by time (the GEMMs); the decode attention and the whole step sit beside it
    assert g["roofline"]["bound"] == "mfma" and "gemm" in g["roofline"]["kernel"] and 0 < g["roofline"]["frac"] < 1
    assert g["roofline"]["head"]["achieved"] > 0 and g["roofline"]["layers"]["launches_per_step"] >= 6  # grouped q/k/v launches counted
    assert g["roofline_attention"]["bound"] == "hbm" and 0 < g["roofline_attention"]["frac"] < 1
    assert 0 < g["step_vs_roofline"]["hbm_frac"] < 1
    assert "EXECUTED" in d["model_tflops_note"] and "dense_equivalent_tflops_per_gpu" in d


def test_bench_default_line_carries_the_dense_caption_step(dev):
    """ragged captions (the headline): TF/s from executed FLOPs, and the measured dense-caption step beside it (full-size only: the
    reduced debug model skips that leg)"""
    r = _run(["--no-generate", "--no-roofline"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "ragged" in d["data"] and d["dense_captions"] is None and "dense_equivalent_tflops_per_gpu" in d
    # beside the headline (inputs resident): the host's issue time of one step on an idle GPU, and the step with the batch copied from
    # pinned host memory inside the loop (main.py:773-775)
    assert 0 < d["host_issue_ms_per_step"] and d["host_loop_ms_per_step"] > 0
    assert d["h2d_inclusive"]["ms_per_step"] > 0 and d["h2d_inclusive"]["steps"] == 2 and d["fp8_train"] is None  # (fp8 leg: full size only)


def test_bench_fp8_line(dev):
    """`--dtype fp8` (configs[4]): the line names the workload and the GEMM dtype, and carries the fp8 GEMMs' own rate"""
    r = _run(["--dtype", "fp8", "--no-generate", "--steps", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["dtype"] == "fp8" and d["config"]["workload"].startswith("configs[4]") and "e4m3" in d["config"]["gemm_dtype"]
    assert d["roofline"]["peak"] == 5000.0 and d["roofline"]["fp8_gemms"]["launches_per_step"] > 0 and d["fp8_train"] is None
    assert abs(d["final_loss"]) < 100
