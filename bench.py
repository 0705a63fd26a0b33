"""bench.py — train-step throughput of the ViT-B/32 + mBART-large-50 captioner on MI355X (BASELINE.json configs[1]:
bf16 train step, batch 64 per GPU, 224x224 images, seq_len 64), one process per GPU, plus the beam-4 captions/sec leg
(configs[3]) as `beam4_generate`.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 ...            # starts its own 8 ranks (torch.distributed.run as a CHILD process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = forward + loss + backward + gradient all-reduce (RCCL, N > 1) + AdamW on synthetic data (random-init weights of
the full architecture; fused dropout active as in training).  Prints ONE JSON line on rank 0.
Variants: --dtype fp8 (configs[4]: e4m3/e5m2 QKV / FFN / LM-head GEMMs), --dense-captions (n = 62 tokens in every caption: no padded
label positions, the dense upper bound of SURVEY §8d), --pmc-traffic (re-measure roofline.traffic with two rocprofv3
counter passes of this same command as child processes).  With --gpus 1 the line also carries `comm_emulated`: the same step
with the gradient exchange among 2 / 4 / 8 ranks EMULATED on this one GPU (a kernel holding 32 CUs of a CU-masked collective
stream for the projected all-reduce time of every bucket; one fresh child process per N) — a scheduling probe, not a scaling
result; --emulate-comm '' switches it off, --emulate-main N makes the timed trainer itself emulate (profiling aid).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_GFLOP_PER_SAMPLE = 201.3  # 3 x 67.1 GF fwd, dense, padding not discounted
PEAK_TFLOPS = {"bf16": 2500.0, "fp8": 5000.0, "f32": 157.3}  # MI355X_MICROARCH.md: dense MFMA peaks
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s is what a copy kernel reaches)
BEAM_GFLOP_PER_CAPTION = 230.0  # 4 rows x 63 steps x 868.5 MF + encoder/cross-KV once
DECODE_STEP_TFLOP, DECODE_STEP_GB = 0.89, 3.1  # one decoder step at 1024 rows (batch 256 x 4 beams), bf16


_T0 = time.time()


def note(msg: str):
    """progress on stderr (stdout carries the ONE JSON line)"""
    print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """threads for the CPU baselines: the cores this process may run on (cgroup / affinity aware), capped at 64 — a pool box can
    expose a few hundred logical CPUs to a container that is scheduled on a fraction of them, and torch's intra-op pool
    degrades badly when oversubscribed"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 quota
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 64))


def rccl_channels_from_log(path: str):
    """the channel count RCCL itself reports in its INIT log (NCCL_DEBUG=INFO): "... N coll channels ..." or, failing that, the
    largest "Channel xx/NN" denominator; None when the log has neither"""
    import re

    try:
        txt = open(path, errors="replace").read()
    except OSError:
        return None
    m = re.findall(r"(\d+) coll channels", txt)
    if m:
        return max(int(x) for x in m)
    m = re.findall(r"Channel \d+/(\d+)", txt)
    return max(int(x) for x in m) if m else None


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` (never exec from
    a process that may touch the GPU; nothing here has imported torch yet) and pass its exit code on."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def synth_batch(B, T, V, img, seed, lang_ids=(250004, 250008, 250003, 250005), dense=False):
    """SURVEY §8(d) synthetic inputs: N(0,1) pixels clipped to [-1.8, 2.2]; labels [lang, n tokens, eos, pad...] with
    ragged n ~ U{8..62} (dense: n = T-2 everywhere); decoder inputs by shift_tokens_right."""
    import numpy as np

    rng = np.random.default_rng(seed)
    px = np.clip(rng.standard_normal((B, img, img, 3), dtype=np.float32), -1.8, 2.2)
    labels = np.full((B, T), 1, dtype=np.int64)
    mask = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        n = T - 2 if dense else int(rng.integers(8, T - 1))
        labels[b, 0] = lang_ids[b % 4] if V > 250008 else V - 4 + (b % 4)
        labels[b, 1:1 + n] = rng.integers(4, min(V, 250000), n)
        labels[b, 1 + n] = 2
        mask[b, :n + 2] = 1
    dec_in = np.full_like(labels, 1)
    dec_in[:, 1:] = labels[:, :-1]
    # int32 ids / masks, as the reference's device arrays are (JAX without x64 narrows the tokenizer's int64 arrays on the HOST, in
    # device_put / shard, main.py:773-775): the narrowing is collate work, not three cast launches at the start of every step
    return {"pixel_values": px, "input_ids": labels.astype(np.int32), "attention_mask": mask.astype(np.int32),
            "decoder_input_ids": dec_in.astype(np.int32)}


# ---------------------------------------------------------------------------------------------- CPU baselines (oracle)
_CPU_PARAMS = {}


def _cpu_oracle_params():
    """full-size oracle parameters, initialised once for both CPU legs (the init is ~10 s of the baselines' budget)"""
    from oracle import model_ref as M

    if "p" not in _CPU_PARAMS:
        rc = M.RefConfig()
        _CPU_PARAMS["rc"], _CPU_PARAMS["p"] = rc, M.init_params(rc, seed=0)
    return _CPU_PARAMS["rc"], _CPU_PARAMS["p"]


def cpu_baseline_train(budget_s=14.0, B=4, min_timed=3):
    """The oracle (torch-CPU fp32 restatement of the reference path; NOT Flax) on this box's host cores: train step
    (fwd + bwd + AdamW) images/s at B=4 on the full-size model.  Bounded sample: one warm-up iteration, then at least `min_timed`
    timed iterations and as many more as fit into `budget_s` of timed work; the spread of the per-iteration rates is stated."""
    import numpy as np
    import torch

    from oracle import train_ref

    cores = host_threads()
    torch.set_num_threads(cores)
    rc, p0 = _cpu_oracle_params()
    p = dict(p0)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    b = synth_batch(B, 64, rc.vocab_size, rc.image_size, 7)
    t = {k: (torch.from_numpy(v) if v.dtype == np.float32 else torch.from_numpy(v.astype(np.int64))) for k, v in b.items()}  # (torch's CPU losses index with int64)
    times, warm, n_warm = [], 0, 1
    for it in range(12):
        t0 = time.time()
        _, g = train_ref.loss_and_grads(rc, p, t["pixel_values"], t["input_ids"], t["attention_mask"], t["decoder_input_ids"])
        with torch.no_grad():
            for k in p:
                p[k], m[k], v2[k] = train_ref.adamw_update(p[k], g[k], m[k], v2[k], it, 5e-5)
        dt = time.time() - t0
        note(f"cpu baseline (train) iteration {it}: {dt:.1f} s")
        if it < n_warm:
            warm += 1
            continue
        times.append(dt)
        if len(times) >= min_timed and sum(times) + dt > budget_s:
            break
    rates = [B / x for x in times]
    return {"value": round(B * len(times) / sum(times), 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "min": round(min(rates), 3), "max": round(max(rates), 3), "timed_iterations": len(times),
            "sample": f"oracle (torch-CPU fp32 restatement, not Flax) train step fwd+bwd+AdamW, full-size model, B={B}, "
                      f"{warm} warm-up + {len(times)} timed iterations ({sum(times):.1f} s of timed work; value = images / total time, min / max = "
                      f"slowest / fastest iteration), {cores} threads of {os.cpu_count()} logical CPUs"}


def cpu_baseline_beam(budget_s=14.0):
    """Beam-4 captions/s of the oracle at B=8 (evaluation.py:80-94 shape: num_beams 4, max_length 64, forced BOS), two timed calls.
    Bounded: when the first 3 decoder steps predict more than `budget_s` for the two calls, max_length is shortened and said so."""
    import numpy as np
    import torch

    from oracle import generation_ref as G
    from oracle import model_ref as M

    cores = host_threads()
    torch.set_num_threads(cores)
    rc, p0 = _cpu_oracle_params()
    p = dict(p0)
    p["final_logits_bias"] = p["final_logits_bias"].clone()
    p["final_logits_bias"][0, rc.eos_token_id] = -1e9  # as in the GPU leg: always the full number of steps
    B, K = 8, 4
    px = torch.from_numpy(np.clip(np.random.default_rng(99).standard_normal((B, rc.image_size, rc.image_size, 3), dtype=np.float32), -1.8, 2.2))

    def run(L):
        t0 = time.time()
        with torch.no_grad():
            ehs, _ = M.encode(rc, p, px, int32_cast=True)
        r = G.generate(lambda rows: G.ModelStepper(rc, p, ehs.repeat_interleave(K, 0), L), B, G.GenDefaults(), num_beams=K,
                       max_length=L, forced_bos_token_id=250004)
        return time.time() - t0, r.steps

    t4, _ = run(4)  # warm-up and probe: encoder + 3 steps
    note(f"cpu baseline (beam-4) probe of 3 steps: {t4:.1f} s")
    # the first steps of a call are the cheap ones (measured on the pool's hosts: 0.2 s per step in the probe, 0.67 s per step over
    # a 49-step call): budget with 3.3 x the probe's figure.  TWO timed calls share the budget, so the line carries a spread.
    per_step = 3.3 * t4 / 4.0
    calls = 2
    L = 64 if per_step * 63 * calls <= budget_s else max(6, int(budget_s / calls / per_step))
    vals, tot_t, tot_steps = [], 0.0, 0
    for _ in range(calls):
        dt, steps = run(L)
        note(f"cpu baseline (beam-4) call of {steps} steps: {dt:.1f} s")
        vals.append(B / (dt * 63.0 / steps))  # captions/s for the full 63-step caption, extrapolated linearly when shortened
        tot_t, tot_steps = tot_t + dt, tot_steps + steps
    return {"value": round(calls * B / (tot_t * 63.0 * calls / tot_steps), 4), "unit": "captions/sec", "cores": cores, "kind": "port",
            "min": round(min(vals), 4), "max": round(max(vals), 4), "timed_calls": calls,
            "sample": f"oracle (torch-CPU fp32 restatement, not Flax) beam-4 generate, full-size model, B={B}, max_length {L} "
                      f"({calls} timed calls of {tot_steps // calls} decoder steps, {tot_t:.1f} s together" + ("" if L == 64 else ", scaled to 63 steps")
                      + f"; min / max = the two calls), after a 3-step warm-up call, {cores} threads of {os.cpu_count()} logical CPUs"}


# ---------------------------------------------------------------------------------------------- beam-4 leg
def bench_generate(model, cfg, dev, batch=256, langs=(250004, 250008, 250003, 250005), max_length=64, seed=99, roofline=True):
    """BASELINE configs[3]: beam-4 `.generate`, 4 forced-BOS languages (en/fr/de/es, one call each like evaluation.py:80-94),
    max_len 64, KV-cached, batch 256 on one GPU.
    final_logits_bias[eos] = -1e9 keeps every run at exactly 63 decoder steps (ForcedEOS still fires at the last step)."""
    import numpy as np
    import torch

    from mic_amd import ops

    st = model.store
    eos = cfg.mbart_config.eos_token_id
    st.f32("flb")[eos] = -1e9
    st.refresh_lp()
    rng = np.random.default_rng(seed)
    img = cfg.clip_vision_config.image_size
    px = torch.from_numpy(np.clip(rng.standard_normal((batch, img, img, 3), dtype=np.float32), -1.8, 2.2)).to(dev)
    V = cfg.mbart_config.vocab_size
    langs = [l if l < V else V - 4 + i for i, l in enumerate(langs)]
    for _ in range(2):  # warm-up: the first call allocates the decode plan (with MIC_DECODE_GRAPHS=1 the second captures its step graphs)
        out = model.generate(px, forced_bos_token_id=langs[0], num_beams=4, max_length=max_length)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for lang in langs:
        out = model.generate(px, forced_bos_token_id=lang, num_beams=4, max_length=max_length)
        assert out["steps"] == max_length - 1, out["steps"]
        n += batch
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = len(langs) * (max_length - 1)
    ms_step = dt / steps * 1e3
    res = {"metric": "beam-4 captions/sec (configs[3]: batch 256, 4 beams, max_len 64, forced BOS, KV-cached)",
           "value": round(n / dt, 1), "unit": "captions/sec", "ms_per_decoder_step": round(ms_step, 3),
           "langs": len(langs), "batch": batch, "model_tflops": round(n * BEAM_GFLOP_PER_CAPTION / dt / 1e3, 1)}
    if not roofline:
        return res
    # one more, untimed call with every decode-attention and GEMM launch bracketed by HIP events on the launch stream
    recs = []
    orig_attn, orig_gemm, orig_grouped = ops.attn_decode, ops.gemm, ops.gemm_grouped
    es = 2 if model.dtype == torch.bfloat16 else 4

    def ev():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed_attn(q, kc, vc, o, R, H, max_len, cur, *, ldq, ldo, ldc=None, src_row=None, row_div=1):
        e0, e1 = ev()
        e0.record()
        r = orig_attn(q, kc, vc, o, R, H, max_len, cur, ldq=ldq, ldo=ldo, ldc=ldc, src_row=src_row, row_div=row_div)
        e1.record()
        slots = min(cur + 1, max_len)
        rows = R if src_row is not None or row_div == 1 else (R + row_div - 1) // row_div  # distinct cache rows read
        by = (rows * slots * 2 * H * 64 + 2 * R * H * 64) * es  # K and V of every valid slot once + q in + context out
        recs.append(("attn", by, e0, e1))
        return r

    def timed_gemm(a, b, o, M, N, K, **kw):
        e0, e1 = ev()
        e0.record()
        r = orig_gemm(a, b, o, M, N, K, **kw)
        e1.record()
        recs.append(("head" if N == st.Vpad else "gemm", 2.0 * M * N * K, e0, e1))
        return r

    def timed_grouped(arg_list):  # the fused q/k/v projection of a decoder layer: one launch, three problems
        e0, e1 = ev()
        e0.record()
        r = orig_grouped(arg_list)
        e1.record()
        recs.append(("gemm", sum(2.0 * g.M * g.N * g.K for g in arg_list), e0, e1))
        return r

    ops.attn_decode, ops.gemm, ops.gemm_grouped = timed_attn, timed_gemm, timed_grouped
    try:
        model.generate(px, forced_bos_token_id=langs[0], num_beams=4, max_length=max_length)
        torch.cuda.synchronize()
    finally:
        ops.attn_decode, ops.gemm, ops.gemm_grouped = orig_attn, orig_gemm, orig_grouped

    def agg(kind):
        sel = [r for r in recs if r[0] == kind]
        return sum(r[1] for r in sel), max(sum(r[2].elapsed_time(r[3]) for r in sel) * 1e-3, 1e-12), max(len(sel), 1)

    ab, at, an = agg("attn")
    hf, ht, hn = agg("head")
    gf, gt, gn = agg("gemm")
    nst = max_length - 1
    # the dominant kernel class of a decoder step BY TIME is the MFMA GEMM (LM head + the layers' projections): that is the
    # leg's `roofline`; the decode attention (HBM-bound) and the whole step are reported beside it
    peak = PEAK_TFLOPS["bf16"] if model.dtype == torch.bfloat16 else PEAK_TFLOPS["f32"]
    res["roofline"] = {"bound": "mfma", "kernel": "gemm_bf16_kernel + gemm_d2_kernel (every GEMM launch of a decoder step: LM head + layer projections)",
                       "achieved": round((hf + gf) / (ht + gt) / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                       "frac": round((hf + gf) / (ht + gt) / 1e12 / peak, 4), "traffic": None,
                       "ms_per_step": round((ht + gt) / nst * 1e3, 3), "launches_per_step": round((hn + gn) / nst, 1),
                       "gflop_per_step": round((hf + gf) / nst / 1e9, 1),
                       "head": {"achieved": round(hf / ht / 1e12, 1), "frac": round(hf / ht / 1e12 / peak, 4),
                                "ms_per_step": round(ht / nst * 1e3, 3), "launches_per_step": round(hn / nst, 1)},
                       "layers": {"achieved": round(gf / gt / 1e12, 1), "frac": round(gf / gt / 1e12 / peak, 4),
                                  "ms_per_step": round(gt / nst * 1e3, 3), "launches_per_step": round(gn / nst, 1)},
                       "note": "HIP events around every GEMM launch (plain and grouped) of one untimed generate call; flops = sum of 2MNK"}
    res["roofline_attention"] = {"bound": "hbm", "kernel": "attn_decode_kernel + attn_decode_group_kernel", "achieved": round(ab / at / 1e9, 1),
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ab / at / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                                 "launches_per_step": round(an / nst, 1), "ms_per_step": round(at / nst * 1e3, 3),
                                 "algorithmic_MB_per_launch": round(ab / an / 1e6, 2),
                                 "note": "algorithmic bytes = K and V of every valid cache slot once (cross K/V once per image, shared by "
                                         "its beams) + q + output; HIP events around every launch of one untimed generate call"}
    for name in ("r6_generate_pmc_hbm_traffic.json", "r5_generate_pmc_hbm_traffic.json", "r4_generate_pmc_hbm_traffic.json", "r3_generate_pmc_hbm_traffic.json", "r2_generate_pmc_hbm_traffic.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if batch == 256 and os.path.exists(pmc):  # HBM-side bytes per launch from separate rocprofv3 --pmc passes over this leg (committed)
            t = json.load(open(pmc))
            src = t.get("summary", "profiles/" + name.replace(".json", ".txt")) + " (commit " + t.get("commit", "<see git log of the file>") + ")"
            ra = res["roofline_attention"]
            att = t.get("attention", t)  # r2 file: attention figures at the top level
            ra["traffic"] = att["bytes_per_launch"]
            ra["frac_on_counter_bytes"] = round(att["bytes_per_launch"] * an / at / 1e9 / PEAK_HBM_GBS, 4)
            ra["traffic_unit"] = ("HBM-side bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, avg over the decode-attention launches of a step; beams "
                                  "that share a prefix hit L2 for the shared slots, hence below the algorithmic figure)")
            ra["traffic_source"] = src
            if "gemm" in t:
                res["roofline"]["traffic"] = t["gemm"]["bytes_per_launch"]
                res["roofline"]["traffic_unit"] = "HBM-side bytes per GEMM launch (PMC FETCH_SIZE x2 + WRITE_SIZE, avg over a decoder step's GEMM launches)"
                res["roofline"]["traffic_source"] = src
            break
    scale = batch * 4 / 1024.0  # SURVEY's step figures are for 1024 rows
    res["step_vs_roofline"] = {"mfma_frac": round(DECODE_STEP_TFLOP * scale / (ms_step * 1e-3) / PEAK_TFLOPS["bf16"], 4),
                               "hbm_frac": round(DECODE_STEP_GB * scale / (ms_step * 1e-3) / PEAK_HBM_GBS, 4),
                               "note": f"whole decoder step: {DECODE_STEP_TFLOP} TFLOP and {DECODE_STEP_GB} GB per 1024 rows (SURVEY 8d) over ms_per_decoder_step"}
    return res


def pmc_traffic(argv, kernel_prefix="gemm_"):
    """roofline.traffic measured for THIS command: two child rocprofv3 passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of
    `bench.py --steps 1 --warmup 1`, summed over the GEMM kernels, FETCH_SIZE doubled (gfx950 correction of the microarch
    guide), both x1024 B, divided by the number of GEMM launches."""
    import csv
    import glob
    import shutil
    import tempfile

    tot, launches = 0.0, 0
    # the child is a fresh single-GPU process: no launcher variables (it must not try to join this job's rendezvous), no --gpus
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE")
    child_argv, skip = [], False
    for a in argv:
        if skip:
            skip = False
        elif a in ("--gpus", "--steps", "--warmup", "--emulate-comm"):
            skip = True
        elif a != "--pmc-traffic" and not a.startswith(("--gpus=", "--steps=", "--warmup=", "--emulate-comm=")):
            child_argv.append(a)
    for ctr, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        d = tempfile.mkdtemp(prefix="mic_pmc_", dir="/tmp")
        env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith(("MASTER_", "TORCHELASTIC_"))}
        env["TMPDIR"] = "/tmp"
        cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__)] + \
              child_argv + ["--steps", "1", "--warmup", "1", "--no-roofline", "--no-generate", "--no-cpu-baseline", "--no-dense-leg", "--no-extra-legs", "--emulate-comm", "0"]
        subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        n = 0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr and kernel_prefix in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"]) * 1024.0 * factor
                    n += 1
        launches = max(launches, n)
        shutil.rmtree(d, ignore_errors=True)
    # `tot` and `launches` both run over the pass's 2 steps (1 warm-up + 1): bytes per launch = tot / launches
    return (tot / launches, launches / 2) if launches else (None, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"])
    ap.add_argument("--grad-comm", default="fp32", choices=["auto", "fp32", "bf16"],
                    help="gradient exchange precision for --gpus > 1: fp32 (default) = the reference's pmean (main.py:698); bf16 = half the xGMI "
                         "bytes, opt-in; auto = bf16 where the PROJECTED fp32 all-reduce would not hide under backward (N = 2, 4: "
                         "mic_amd.train.choose_comm_dtype), fp32 elsewhere")
    ap.add_argument("--emulate-comm", default=None, help="(default 2,4,8; off for --small) ""world sizes whose gradient exchange is EMULATED on this one GPU after the timed region "
                    "(a kernel holding --comm-cus CUs for the projected all-reduce time of every bucket): reported as `comm_emulated`, a "
                    "scheduling probe, not a scaling result; '' or 0 = off; only with --gpus 1")
    ap.add_argument("--emulate-main", type=int, default=0, help="profiling aid: the TIMED trainer itself runs with the emulated exchange among N ranks "
                    "(the line is labelled; not a benchmark)")
    ap.add_argument("--comm-cus", type=int, default=None, help="RCCL channel cap = CUs the collectives may hold (NCCL_MAX_NCHANNELS, exported before "
                    "init_process_group; default mic_amd.train.COMM_CUS_DEFAULT); the GEMM tile planner is sized for the rest")
    ap.add_argument("--fp8-scaling", default="delayed", choices=["delayed", "current"],
                    help="--dtype fp8: scale from the previous step's amax (one pass per tensor) or from the current amax (two)")
    ap.add_argument("--dense-captions", action="store_true", help="every caption has T-2 tokens (no padded label positions): dense upper bound")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-generate", action="store_true", help="skip the beam-4 captions/sec leg")
    ap.add_argument("--no-dense-leg", action="store_true", help="skip the short dense-caption timing reported beside the headline")
    ap.add_argument("--no-fp8-leg", action="store_true", help="skip the configs[4] (fp8 GEMMs) step timed in a child process beside the bf16 headline")
    ap.add_argument("--no-extra-legs", action="store_true", help="profiling / A-B aid: only the timed region (no idle-GPU issue timing, no host-batch "
                    "(h2d_inclusive) leg, no fp8 leg) — the step count of a kernel trace is then exactly warmup + steps")
    ap.add_argument("--gen-batch", type=int, default=256)
    ap.add_argument("--generate-only", action="store_true", help="profiling aid: only the beam-4 leg")
    ap.add_argument("--pmc-traffic", action="store_true", help="measure roofline.traffic now (two rocprofv3 child passes, ~2 min)")
    ap.add_argument("--small", action="store_true", help="reduced model (debug only; result is NOT the benchmark)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL; must be set before the HIP runtime starts
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share = bool(os.environ.get("MIC_BENCH_SHARE_GPU0"))  # debugging aid: all ranks on cuda:0, gradients over gloo (not a benchmark)
    if world > 1 and not share and torch.cuda.device_count() < world:
        if rank == 0:
            print(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible; set MIC_BENCH_SHARE_GPU0=1 to run all "
                  "ranks on cuda:0 over gloo (a functional check, not a benchmark)", file=sys.stderr)
        sys.exit(2)
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_cap, rccl_log, rccl_reported = 0, None, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if share:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                # enforce what the tile planner assumes: RCCL held to --comm-cus channels (one persistent block = one CU each), and
                # have rank 0 read back what RCCL reports (its INIT log) instead of trusting the cap
                from importlib import import_module

                rccl_cap = import_module("mic_amd.train").configure_rccl(args.comm_cus if args.comm_cus is not None else 32)
                if rank == 0 and "NCCL_DEBUG" not in os.environ:
                    rccl_log = f"/tmp/mic_rccl_init_{os.getpid()}.log"
                    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE=rccl_log)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                # the first collective creates the RCCL communicator (ring / tree set-up over xGMI): fail here, with the cause, not
                # somewhere inside the first train step
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                assert int(probe.item()) == world, f"all-reduce probe returned {probe.item()} for world {world}"
                if rank == 0:
                    rccl_reported = rccl_channels_from_log(rccl_log) if rccl_log else None
                    note(f"RCCL up over {world} ranks: channel cap NCCL_MAX_NCHANNELS={rccl_cap}, RCCL reports {rccl_reported} channels")
        except Exception as e:  # name the cause and leave with a non-zero code (the launcher then ends the other ranks)
            print(f"bench.py rank {rank}/{world} (device {local_rank}): torch.distributed / RCCL initialisation failed: {type(e).__name__}: {e}\n"
                  f"  MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} "
                  f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')} visible GPUs={torch.cuda.device_count()}\n"
                  "  (RCCL needs one distinct GPU per rank and dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0; NCCL_DEBUG=INFO prints its own log)",
                  file=sys.stderr, flush=True)
            sys.exit(3)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import mic_amd  # noqa: F401
    from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, ops

    if args.small:
        cfg = CLIPVisionMBartConfig(mbart_config=dict(vocab_size=5003, d_model=256, decoder_layers=2, decoder_attention_heads=4, decoder_ffn_dim=512),
                                    clip_vision_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=64, patch_size=32))
    else:
        cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=dtype, device=dev)
    B, T = args.batch, 64
    lr_fn = create_learning_rate_fn(train_ds_size=10_000_000, train_batch_size=B * world, num_train_epochs=7, num_warmup_steps=1000, learning_rate=5e-5)
    tkw = {}
    if args.dtype == "fp8":
        tkw["gemm_dtype"] = "fp8"
        tkw["fp8_scaling"] = args.fp8_scaling
    if args.comm_cus is not None:
        tkw["comm_cus"] = args.comm_cus
    comm_arg = {"auto": "auto", "fp32": None, "bf16": torch.bfloat16}[args.grad_comm]
    tr = Trainer(model, lr_fn, seed=42, grad_comm_dtype=comm_arg, emulate_comm=args.emulate_main, **tkw)
    V, img = cfg.mbart_config.vocab_size, cfg.clip_vision_config.image_size
    batches = [synth_batch(B, T, V, img, 1234 + rank * 100 + i, dense=args.dense_captions) for i in range(2)]
    # inputs resident in HBM before the timed region
    dbatches = [{k: torch.from_numpy(v).to(dev) for k, v in b.items()} for b in batches]
    from mic_amd import loss_rows, packed_rows

    def collate_extras(b, db):
        """collate-side: positions that carry loss (the LM head runs only there) and the packed-row description of the batch (the
        decoder runs only on the valid caption positions); host work, outside the timed region like the reference's collate_fn"""
        idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
        db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
        pk = packed_rows(b["attention_mask"], b["decoder_input_ids"])
        if pk is not None:
            db["packed_rows"] = tuple(torch.from_numpy(t).to(dev) for t in pk)

    for b, db in zip(batches, dbatches):
        collate_extras(b, db)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.generate_only:
        print(json.dumps(bench_generate(model, cfg, dev, batch=args.gen_batch, roofline=not args.no_roofline)))
        return

    if rank == 0:
        note("model and batches ready; warm-up")
    for i in range(args.warmup):
        tr.train_step(dbatches[i % 2])
    barrier()
    if rank == 0:
        note("timed steps")
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = tr.train_step(dbatches[i % 2])
    t_issue = time.perf_counter() - t0  # the host's share: all K steps queued (nothing in a step waits for the GPU)
    reducer_host_ms = tr.reducer.host_s * 1e3  # ... of which inside GradReducer.progress / finish, last timed step
    barrier()
    dt = time.perf_counter() - t0
    loss = float(out["loss"])
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    images_per_sec = world * B * args.steps / dt

    # host time to ISSUE one step with the GPU idle (queue empty at the start: no back-pressure; `host_loop_ms_per_step` above is the
    # figure of a loop whose queue is full, i.e. it mostly measures the GPU): median of 3, each behind a device sync
    issue_idle = []
    for i in range(0 if args.no_extra_legs else 3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr.train_step(dbatches[i % 2])
        issue_idle.append(time.perf_counter() - t1)
    barrier()
    issue_idle_ms = sorted(issue_idle)[1] * 1e3 if issue_idle else None

    # the step as main.py:773-775 pays it: the batch arrives in HOST memory every iteration.  Pinned host buffers, the copy of batch
    # i+1 on a copy stream beside step i (two device slots), the step waits for its own batch's event.  Reported beside the
    # headline (`value` keeps the inputs resident, as the bench contract asks); never part of `value`.
    h2d = None
    if not args.emulate_main and not args.no_extra_legs:
        pinned = []
        for b, db in zip(batches, dbatches):
            hp = {k: torch.from_numpy(v).pin_memory() for k, v in b.items()}
            for k in ("loss_rows", "packed_rows"):
                if k in db:
                    hp[k] = tuple(t.cpu().pin_memory() for t in db[k])
            pinned.append(hp)
        NSLOT = 4  # device slots (slot j holds batches of parity j % 2: the two synthetic batches differ in their row counts): the host waits
        #            (event.synchronize — no barrier packet on the copy stream) for the step that last read one
        slots = [{k: (tuple(torch.empty_like(t, device=dev) for t in v) if isinstance(v, tuple) else torch.empty_like(v, device=dev))
                  for k, v in pinned[j % 2].items()} for j in range(NSLOT)]
        # the copy stream: HIP multiplexes a process's streams onto a few hardware queues, and a NEW stream created behind the trainer's
        # role streams measured +1.2 ms per step (the copy shared a queue with step work) where the same copy beside the step is free
        # (tools/h2d_probe.py: 0.74 ms alone, 17.90 against 17.97 ms beside the step).  At world 1 the reducer's collective stream
        # exists and idles: the batch travels there.
        copy_stream = ops.role_stream(dev, "collective") if world == 1 else torch.cuda.Stream(device=dev)
        done = [None] * NSLOT

        def stage(i):
            if done[i % NSLOT] is not None:
                done[i % NSLOT].synchronize()  # the step that last read this slot (four steps back: the host is never that far ahead)
            with torch.cuda.stream(copy_stream):
                for k, v in pinned[i % 2].items():
                    if isinstance(v, tuple):
                        for dst, src in zip(slots[i % NSLOT][k], v):
                            dst.copy_(src, non_blocking=True)
                    else:
                        slots[i % NSLOT][k].copy_(v, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return ev

        def inclusive(nsteps):
            ev = stage(0)
            for i in range(nsteps):
                torch.cuda.current_stream().wait_event(ev)
                if i + 1 < nsteps:
                    ev = stage(i + 1)
                tr.train_step(slots[i % NSLOT])
                d = torch.cuda.Event()
                d.record()
                done[i % NSLOT] = d

        inclusive(2)
        barrier()
        t1 = time.perf_counter()
        inclusive(args.steps)
        barrier()
        hdt = time.perf_counter() - t1
        if world > 1:
            tmax = torch.tensor([hdt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            hdt = float(tmax.item())
        mb = sum(v.numel() * v.element_size() for v in pinned[0].values() if not isinstance(v, tuple)) / 1e6
        h2d = {"ms_per_step": round(hdt / args.steps * 1e3, 3), "images_per_sec": round(world * B * args.steps / hdt, 1), "steps": args.steps,
               "note": f"the same step with every batch copied from pinned host memory inside the timed loop ({mb:.1f} MB per step; main.py:773-775 "
                       "shards the host batch onto the devices every iteration): batch i+1 travels on a copy stream beside step i"}
        del slots, pinned

    # the same step on dense captions (every caption 62 tokens: no padded label positions, the configuration BASELINE.md §4's
    # "padding not discounted" FLOP count describes) — a second, shorter timed region on every rank, reported beside the headline
    dense, db2 = None, None
    if not args.dense_captions and not args.no_dense_leg and not args.small:
        dsteps = max(2, min(args.steps, 6))
        db2 = []
        for i in range(2):
            b = synth_batch(B, T, V, img, 4321 + rank * 100 + i, dense=True)
            d2 = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
            collate_extras(b, d2)
            db2.append(d2)
        for i in range(2):
            tr.train_step(db2[i % 2])
        barrier()
        t0 = time.perf_counter()
        for i in range(dsteps):
            tr.train_step(db2[i % 2])
        barrier()
        ddt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([ddt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            ddt = float(tmax.item())
        dense = {"ms_per_step": round(ddt / dsteps * 1e3, 3), "images_per_sec": round(world * B * dsteps / ddt, 1), "steps": dsteps,
                 "model_tflops_per_gpu": round(TRAIN_GFLOP_PER_SAMPLE * B * dsteps / ddt / 1e3, 1),
                 "note": "every caption 62 tokens + language id + eos: all 64 positions carry loss, the LM head runs on every row; "
                         "201.3 GFLOP per sample are executed in full; roofline_frac = the GEMM fraction of peak of THIS step "
                         "(instrumented like `roofline`), i.e. the kernels' efficiency when no row can be packed away"}

    # this trainer's own emulation figures (bench.py --emulate-main N: the child processes of the `comm_emulated` legs, and the profiling aid)
    emulation = None
    if args.emulate_main and world == 1 and tr.reducer.emulate is not None:
        from mic_amd.train import allreduce_ms

        tr.reducer.emulate["timing"] = True  # one more step with the stand-in kernels bracketed by events
        tr.train_step(dbatches[0])
        barrier()
        busy = sum(a.elapsed_time(b_) for a, b_ in tr.reducer.emulated_events)
        tr.reducer.emulate["timing"] = False
        emulation = {"world": args.emulate_main, "comm_dtype": "bf16" if tr.grad_comm_dtype is not None else "fp32", "comm_cus": tr.comm_cus,
                     "reducer_host_ms_per_step": round(tr.reducer.host_s * 1e3, 3),  # host time inside GradReducer.progress / finish (last step)
                     "projected_allreduce_ms_per_step": round(tr.reducer.emulated_ms, 2), "collective_stream_busy_ms_per_step": round(busy, 2),
                     "projected_allreduce_ms_fp32": round(allreduce_ms(4.0 * model.store.numel, args.emulate_main), 2)}

    if rank == 0:
        note(f"{images_per_sec:.1f} images/s; roofline step")
    roofline = None
    peak = PEAK_TFLOPS[args.dtype]

    def gemm_roofline_step(pair):
        """one lead-in step + one step whose GEMM launches (plain and grouped) are bracketed by HIP events on the launch stream, on
        rank 0; the other ranks run the same two steps plainly (they contain collectives).  Returns the event records on rank 0."""
        if rank != 0:
            tr.train_step(pair[1])
            tr.train_step(pair[0])
            torch.cuda.synchronize()
            return None
        recs = []
        orig, orig_g = ops.gemm, ops.gemm_grouped

        def timed_gemm(a, b, out, M, N, K, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(a, b, out, M, N, K, **kw)
            e1.record()
            recs.append((2.0 * M * N * K, e0, e1, str(a.dtype)))
            return r

        def timed_grouped(arg_list):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_g(arg_list)
            e1.record()
            recs.append((sum(2.0 * g.M * g.N * g.K for g in arg_list), e0, e1, "grouped"))
            return r

        # the instrumented step keeps the weight-gradient GEMMs on the main stream: on their own stream (the default) they overlap
        # the next layer's kernels and an event pair would time two kernels sharing the chip, not the kernel
        # ... and AdamW as one launch after backward: per bucket on its own stream (the default) it shares HBM with the GEMMs it overlaps
        eng = model.engine
        overlap, eng.dw_overlap = eng.dw_overlap, False
        opt_overlap, on_ready = tr.overlap_optimizer, tr.reducer.on_ready
        if world == 1:
            tr.overlap_optimizer, tr.reducer.on_ready = False, None
        # an un-instrumented step goes first WITHOUT a sync in between: the host then issues the instrumented step while the GPU
        # is still busy, so no event pair contains the host's issue time of its kernel (two event records + a ctypes launch cost
        # about as much host time as a 25-us GEMM runs)
        try:
            ops.gemm, ops.gemm_grouped = orig, orig_g
            tr.train_step(pair[1])
            ops.gemm, ops.gemm_grouped = timed_gemm, timed_grouped
            tr.train_step(pair[0])
            torch.cuda.synchronize()
        finally:  # an exception must not leave the trainer or the ops module patched for the beam-4 leg
            eng.dw_overlap = overlap
            tr.overlap_optimizer, tr.reducer.on_ready = opt_overlap, on_ready
            ops.gemm, ops.gemm_grouped = orig, orig_g
        return recs

    if not args.no_roofline:
        recs_dense = gemm_roofline_step(db2) if dense is not None else None
        if rank == 0 and recs_dense:
            fd, md = sum(r[0] for r in recs_dense), sum(r[1].elapsed_time(r[2]) for r in recs_dense)
            dense["roofline_frac"] = round(fd / (md * 1e-3) / 1e12 / peak, 4)
            dense["gemm_ms_per_step"] = round(md, 3)
            dense["gemm_gflop_per_step"] = round(fd / 1e9, 1)
    db2 = None
    recs = gemm_roofline_step(dbatches) if not args.no_roofline else None
    if not args.no_roofline and rank == 0:
        # dominant kernel = the MFMA GEMM (gemm_bf16_kernel and, for the LM head / all-layer cross k/v forward, gemm_w4_kernel): achieved = sum(2MNK) / sum(duration) over every
        # launch of the instrumented step
        flops = sum(r[0] for r in recs)
        ms = sum(r[1].elapsed_time(r[2]) for r in recs)
        ach = flops / (ms * 1e-3) / 1e12
        kname = {"bf16": "gemm_bf16_kernel + gemm_w4_kernel + gemm_d2_kernel (every GEMM launch of the step)", "f32": "gemm_f32_kernel", "fp8": "gemm_bf16_kernel + gemm_fp8_kernel"}[args.dtype]
        roofline = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None, "launches_per_step": len(recs), "gemm_ms_per_step": round(ms, 3),
                    "gemm_gflop_per_step": round(flops / 1e9, 1),
                    # the whole step against the same peak: every executed GEMM FLOP of the instrumented step over the TIMED ms_per_step
                    # (attention, LayerNorm, loss, optimizer and every gap count as time, not as work)
                    "whole_step_frac": round(flops / (dt / args.steps) / 1e12 / peak, 4)}
        if dense is not None and "roofline_frac" in dense:
            roofline["note"] = ("with packed decoder rows the executed GEMM FLOPs fall faster than the GEMM time (the one-round launches of "
                                "the N = 1024 projections are latency-bound: fewer tiles, same duration), so this fraction sits below the "
                                f"same kernels' fraction on dense captions (dense_captions.roofline_frac = {dense['roofline_frac']})")
        if args.dtype == "fp8":
            f8 = [r for r in recs if "float8" in r[3]]
            if f8:
                f8f, f8t = sum(r[0] for r in f8), sum(r[1].elapsed_time(r[2]) for r in f8) * 1e-3
                roofline["fp8_gemms"] = {"achieved": round(f8f / f8t / 1e12, 1), "frac_of_fp8_peak": round(f8f / f8t / 1e12 / PEAK_TFLOPS["fp8"], 4),
                                         "launches_per_step": len(f8), "ms_per_step": round(f8t * 1e3, 3)}
            roofline["note"] = "peak = dense fp8 MFMA peak; the QKV / FFN projections and the tied LM head run in fp8 (configs[4]; MIC_FP8_HEAD), the attention out-projections and the patch embedding in bf16"
        if args.pmc_traffic and world > 1:
            note("--pmc-traffic is a single-GPU measurement (rocprofv3 child passes of this command); skipped under a launcher")
        if args.pmc_traffic and world == 1:
            tb, nl = pmc_traffic(sys.argv[1:])
            roofline["traffic"] = None if tb is None else int(tb)
            roofline["traffic_unit"] = "HBM bytes per GEMM launch (PMC FETCH_SIZE x2 + WRITE_SIZE, x1024 B), measured now by two rocprofv3 child passes"
            roofline["traffic_launches_per_step"] = nl
        else:
            # HBM bytes per launch come from separate rocprofv3 --pmc passes (a profiler cannot wrap this very process); the
            # committed summary of the last such passes over this same command is reported with the commit it was taken at.
            for name in ("r6_train_pmc_hbm_traffic.json", "r5_train_pmc_hbm_traffic.json", "r4_train_pmc_hbm_traffic.json", "r3_train_pmc_hbm_traffic.json", "r2_train_pmc_hbm_traffic.json", "r1_train_pmc_hbm_traffic.json"):
                pmc = os.path.join(ROOT, "profiles", name)
                if args.dtype == "bf16" and B == 64 and not args.small and not args.dense_captions and os.path.exists(pmc):
                    t = json.load(open(pmc))
                    roofline["traffic"] = t["bytes_per_launch"]
                    roofline["traffic_unit"] = "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, avg over the step's GEMM launches)"
                    roofline["traffic_source"] = (t.get("summary", "profiles/" + name.replace(".json", ".txt")) + " (committed counter passes of this command at commit "
                                                  + t.get("commit", "<see git log of the file>") + "; --pmc-traffic re-measures)")
                    break
    if world > 1:
        dist.barrier()

    if rank == 0:
        note("beam-4 leg")
    gen = None
    if not args.no_generate:
        # generation is replicas-only (no collective): every rank decodes its own 256 images; report the sum
        g = bench_generate(model, cfg, dev, batch=args.gen_batch if not args.small else 8, roofline=(rank == 0 and not args.no_roofline))
        if world > 1:
            tv = torch.tensor([g["value"]], dtype=torch.float64, device=dev)
            dist.all_reduce(tv, op=dist.ReduceOp.SUM)
            g["value"] = round(float(tv.item()), 1)
        g["n_gpus"] = world
        gen = g
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.small:
        note(f"CPU baselines on {host_threads()} threads")
        try:
            cpu = cpu_baseline_train()
        except Exception as e:  # the baseline is a reported figure, never a dependency of the GPU number
            cpu = {"value": None, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        if gen is not None:
            try:
                gen["cpu_baseline"] = cpu_baseline_beam()
            except Exception as e:
                gen["cpu_baseline"] = {"value": None, "unit": "captions/sec", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}

    # The same step with an EMULATED gradient exchange among N ranks on this one GPU, one fresh child process per N (`--emulate-main N`:
    # exactly one Trainer per process, as a data-parallel rank has): every bucket's all-reduce is replaced by a kernel that holds
    # `comm_cus` CUs (a CU-masked collective stream) for the time a ring all-reduce of that bucket is projected to take over xGMI
    # (mic_amd.train.allreduce_ms), the bucket's optimizer pass waits for it as it would for RCCL, and the GEMM tile planner is sized
    # for the remaining CUs.  What it shows: whether the step's scheduling (bucket order, optimizer stream, CU budget) holds up when
    # a collective stream is busy beside backward.  NOT a scaling result: no byte crosses a link.
    emulated = None
    emu_arg = args.emulate_comm if args.emulate_comm is not None else ("" if args.small else "2,4,8")
    emu_worlds = [int(x) for x in str(emu_arg).split(",") if x.strip() not in ("", "0", "1")]
    if rank == 0 and world == 1 and emu_worlds and args.dtype != "f32" and not args.emulate_main:
        note(f"emulated-exchange legs for N = {emu_worlds} (one child process each)")
        torch.cuda.synchronize()
        emulated = {"note": "1-GPU step (a fresh process per N, 6 timed steps) with every bucket's all-reduce replaced by a kernel holding comm_cus CUs on a "
                            "CU-masked stream for the projected ring all-reduce time of that bucket (2 (N-1)/N x bytes over N-1 xGMI links at 64 GB/s "
                            "each); optimizer passes wait for it as for RCCL; GEMM tile planner sized for 256 - comm_cus CUs.  A scheduling probe, NOT a "
                            "scaling result", "baseline_ms_per_step": round(dt / args.steps * 1e3, 3), "worlds": {}}
        drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE")
        env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith(("MASTER_", "TORCHELASTIC_"))}
        for ew in emu_worlds:
            cmd = [sys.executable, os.path.abspath(__file__), "--emulate-main", str(ew), "--emulate-comm", "0", "--steps", "6", "--warmup", "3",
                   "--no-generate", "--no-cpu-baseline", "--no-roofline", "--no-dense-leg", "--no-extra-legs", "--dtype", args.dtype, "--grad-comm", args.grad_comm,
                   "--batch", str(args.batch)] + (["--small"] if args.small else []) + (["--comm-cus", str(args.comm_cus)] if args.comm_cus is not None else [])
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                cd = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                e = dict(cd["emulation"])
                e.pop("world", None)
                emulated["worlds"][str(ew)] = {"ms_per_step": cd["ms_per_step"], "images_per_sec_per_gpu": cd["value"], **e}
            except Exception as ex:  # a reported figure, never a dependency of the headline number
                emulated["worlds"][str(ew)] = {"ms_per_step": None, "error": f"{type(ex).__name__}: {ex}"[:300]}

    # configs[4] beside the headline: the same step with the QKV / FFN projections and the LM head as fp8 GEMMs, in a fresh child process (one Trainer
    # per process, like a data-parallel rank), so the driver's record carries it
    fp8_leg = None
    if rank == 0 and world == 1 and args.dtype == "bf16" and not args.no_fp8_leg and not args.no_extra_legs and not args.small and not args.emulate_main and not args.dense_captions:
        note("fp8 leg (child process)")
        torch.cuda.synchronize()
        drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE")
        env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith(("MASTER_", "TORCHELASTIC_"))}
        cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "fp8", "--steps", str(args.steps), "--warmup", str(max(args.warmup, 3)), "--no-generate",
               "--no-cpu-baseline", "--no-dense-leg", "--no-extra-legs", "--emulate-comm", "0", "--batch", str(args.batch)] + (["--no-roofline"] if args.no_roofline else [])
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            cd = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            fp8_leg = {"metric": "train images/sec, configs[4]: QKV / FFN projections of both towers and the tied LM head as OCP fp8 GEMMs (e4m3 forward, e5m2 "
                                 "gradients, delayed per-tensor scaling — dlogits under a closed-form scale, label entries exact —, fp32 accumulate), "
                                 "everything else as in the headline step",
                       "value": cd["value"], "unit": "images/sec", "ms_per_step": cd["ms_per_step"], "steps": cd["steps"], "final_loss": cd["final_loss"],
                       "vs_bf16_same_box": round(cd["value"] / images_per_sec, 4), "roofline": cd.get("roofline")}
        except Exception as ex:  # a reported figure, never a dependency of the headline number
            fp8_leg = {"value": None, "error": f"{type(ex).__name__}: {ex}"[:300]}

    if rank == 0:
        # executed work: the LM head (forward, dE, dX: 3 x 2 x rows x V x d) runs only on the label positions that carry loss
        n_loss = sum(int(b["attention_mask"].sum()) for b in batches) / len(batches)
        d_model = cfg.mbart_config.d_model
        dense_flops = TRAIN_GFLOP_PER_SAMPLE * 1e9 * B if not args.small else 0.0  # (the FLOP model is for the full-size network)
        step_flops = max(dense_flops - 6.0 * (B * T - n_loss) * V * d_model, 0.0)
        packed = tr.pack_rows and args.dtype in ("bf16", "fp8") and not args.dense_captions
        if packed:  # the decoder layers run on the valid rows only: 14 d^2 multiply-adds per row and layer (qkv 3, so 1, cq 1, co 1, fc1 4, fc2 4), x3 for fwd + bwd
            step_flops = max(step_flops - 6.0 * 14.0 * d_model * d_model * cfg.mbart_config.decoder_layers * (B * T - n_loss), 0.0)
        head = ("logits/CE on all label positions (dense captions: every position carries loss)" if args.dense_captions else
                "logits/CE on the label positions with loss mask 1 only (exact; ragged captions n~U{8..62})")
        line = {
            "metric": "train images/sec, ViT-B/32+mBART-50 (bf16 train step, batch 64/GPU, 224x224, seq_len 64)",
            "value": round(images_per_sec, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (random-init weights, N(0,1) pixels, " + ("dense 62-token" if args.dense_captions else "ragged") + " random captions)",
            "config": {"workload": ("configs[4]" if args.dtype == "fp8" else "configs[1]") + ": ViT-B/32 + mBART-large-50 train step (fwd+loss+bwd+all-reduce+AdamW), "
                                   f"per-GPU batch {B}, 224x224 NHWC fp32 pixels, seq_len {T}, dropout 0.1" + (" [SMALL DEBUG MODEL]" if args.small else "") + (f" [EMULATED {args.emulate_main}-rank exchange on one GPU: profiling aid, not a benchmark]" if args.emulate_main else "")
                                   + (" [ALL RANKS SHARE cuda:0 OVER gloo: functional check, not a benchmark]" if share and world > 1 else ""),
                       "global_batch": B * world, "seq_len": T, "parallelism": f"dp{world}",
                       "grad_allreduce": (f"{'bf16' if tr.grad_comm_dtype is not None else 'fp32'} flat buckets (--grad-comm {args.grad_comm}), RCCL, side stream, "
                                          f"{len(tr.buckets)} buckets <= 128 MB" + (f", RCCL capped at {rccl_cap} channels (reports {rccl_reported}), GEMM tile planner sized for {256 - tr.comm_cus} CUs" if world > 1 and tr.comm_cus else "")),
                       "optimizer": ("AdamW per gradient bucket behind backward" + (f", on a stream masked to {os.environ.get('MIC_OPT_CUS', '96')} CUs" if tr.reducer.step_stream is not None else "")
                                     + ("; tied embedding updated in two row passes (rows without / with sparse gradient)" if tr._split_shared else "")) if tr.overlap_optimizer else "AdamW, one launch after backward",
                       "decoder_rows": ("valid caption positions only (packed rows: padded positions neither carry loss nor are attended to — exact; "
                                        f"{n_loss:.0f} of {B * T} rows per step)" if (tr.pack_rows and args.dtype in ("bf16", "fp8") and not args.dense_captions) else f"all {B * T} positions"),
                       "lm_head": head, "gemm_dtype": "fp8 e4m3 (fwd) / e5m2 (grads) for QKV + FFN + the tied LM head (MIC_FP8_HEAD=" + os.environ.get("MIC_FP8_HEAD", "all") + "), bf16 elsewhere" if args.dtype == "fp8" else args.dtype},
            "host_issue_ms_per_step": None if issue_idle_ms is None else round(issue_idle_ms, 3),  # host time to enqueue ONE step on an idle GPU (median of 3): the host's real share
            "host_loop_ms_per_step": round(t_issue / args.steps * 1e3, 3),  # the timed loop's host side with the queue full (back-pressured by the GPU: NOT the host's cost)
            "h2d_inclusive": h2d,
            "reducer_host_ms_per_step": round(reducer_host_ms, 3),  # of which inside GradReducer.progress / finish (last timed step: bucket events, exchanges, optimizer passes)
            "model_tflops_per_gpu": round(step_flops * args.steps / dt / 1e12, 1),
            "model_tflops_note": f"EXECUTED model FLOPs per step ({step_flops / 1e12:.2f} TF: dense {dense_flops / 1e12:.2f} TF of SURVEY 8d minus the LM-head "
                                 + ("and decoder-layer " if packed else "") + f"work on the {B * T - n_loss:.0f} padded label positions, whose loss weight is 0"
                                 + (" and which no valid position attends to" if packed else "") + ") / ms_per_step; dense_equivalent divides the full dense count "
                                 "by the same time (NOT executed work); dense_captions is the measured step when every position carries loss",
            "dense_equivalent_tflops_per_gpu": round(dense_flops * args.steps / dt / 1e12, 1),
            "dense_captions": dense, "fp8_train": fp8_leg,
            "comm_emulated": emulated, "emulation": emulation,
            "final_loss": round(loss, 4),
            "roofline": roofline, "cpu_baseline": cpu, "beam4_generate": gen,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
