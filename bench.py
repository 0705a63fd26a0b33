"""bench.py — train-step throughput of the ViT-B/32 + mBART-large-50 captioner on MI355X (BASELINE.json configs[1]:
bf16 train step, batch 64 per GPU, 224x224 images, seq_len 64), one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = forward + loss + backward + gradient all-reduce (RCCL, N > 1) + AdamW on synthetic data (random-init weights of
the full architecture; fused dropout active as in training).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_GFLOP_PER_SAMPLE = 201.3  # 3 x 67.1 GF fwd, dense, padding not discounted
PEAK_BF16_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA peak


def synth_batch(B, T, V, img, seed, lang_ids=(250004, 250008, 250003, 250005)):
    """SURVEY §8(d) synthetic inputs: N(0,1) pixels clipped to [-1.8, 2.2]; labels [lang, n tokens, eos, pad...] with
    ragged n ~ U{8..62}; decoder inputs by shift_tokens_right."""
    import numpy as np

    rng = np.random.default_rng(seed)
    px = np.clip(rng.standard_normal((B, img, img, 3), dtype=np.float32), -1.8, 2.2)
    labels = np.full((B, T), 1, dtype=np.int64)
    mask = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(8, T - 1))
        labels[b, 0] = lang_ids[b % 4] if V > 250008 else V - 4 + (b % 4)
        labels[b, 1:1 + n] = rng.integers(4, min(V, 250000), n)
        labels[b, 1 + n] = 2
        mask[b, :n + 2] = 1
    dec_in = np.full_like(labels, 1)
    dec_in[:, 1:] = labels[:, :-1]
    return {"pixel_values": px, "input_ids": labels, "attention_mask": mask, "decoder_input_ids": dec_in}


def cpu_baseline(seconds_budget=25.0):
    """The oracle (torch-CPU fp32 restatement of the reference path; NOT Flax) timed on this box's host cores: fwd+bwd
    images/s at B=8 on the full-size model, bounded sample."""
    import torch

    from oracle import model_ref as M
    from oracle import train_ref

    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    rc = M.RefConfig()
    p = M.init_params(rc, seed=0)
    import numpy as np

    b = synth_batch(8, 64, rc.vocab_size, rc.image_size, 7)
    t = {k: torch.from_numpy(v) for k, v in b.items()}
    n, t_used = 0, 0.0
    for it in range(4):
        t0 = time.time()
        train_ref.loss_and_grads(rc, p, t["pixel_values"], t["input_ids"], t["attention_mask"], t["decoder_input_ids"])
        dt = time.time() - t0
        if it > 0:
            n += 8
            t_used += dt
        if t_used > seconds_budget:
            break
    return {"value": round(n / t_used, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle (torch-CPU fp32 restatement, not Flax) fwd+bwd, full-size model, B=8 x {n // 8} timed iters after 1 warm-up, no optimizer"}


BEAM_GFLOP_PER_CAPTION = 230.0  # 4 rows x 63 steps x 868.5 MF + encoder/cross-KV once


def bench_generate(model, cfg, dev, batch=256, langs=(250004, 250008, 250003, 250005), max_length=64, seed=99):
    """BASELINE configs[3]: beam-4 `.generate`, 4 forced-BOS languages (en/fr/de/es, one call each like evaluation.py:80-94),
    max_len 64, KV-cached, batch 256 on one GPU.
    final_logits_bias[eos] = -1e9 keeps every run at exactly 63 decoder steps (ForcedEOS still fires at the last step)."""
    import numpy as np
    import torch

    st = model.store
    eos = cfg.mbart_config.eos_token_id
    st.f32("flb")[eos] = -1e9
    st.refresh_lp()
    rng = np.random.default_rng(seed)
    img = cfg.clip_vision_config.image_size
    px = torch.from_numpy(np.clip(rng.standard_normal((batch, img, img, 3), dtype=np.float32), -1.8, 2.2)).to(dev)
    V = cfg.mbart_config.vocab_size
    langs = [l if l < V else V - 4 + i for i, l in enumerate(langs)]
    out = model.generate(px, forced_bos_token_id=langs[0], num_beams=4, max_length=max_length)  # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for lang in langs:
        out = model.generate(px, forced_bos_token_id=lang, num_beams=4, max_length=max_length)
        assert out["steps"] == max_length - 1, out["steps"]
        n += batch
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"metric": "beam-4 captions/sec (configs[3]: batch 256, 4 beams, max_len 64, forced BOS, KV-cached)",
            "value": round(n / dt, 1), "unit": "captions/sec", "ms_per_decoder_step": round(dt / (len(langs) * (max_length - 1)) * 1e3, 3),
            "langs": len(langs), "batch": batch, "model_tflops": round(n * BEAM_GFLOP_PER_CAPTION / dt / 1e3, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--grad-comm", default="fp32", choices=["fp32", "bf16"],
                    help="gradient exchange precision for --gpus > 1 (fp32 = the reference's pmean; bf16 = opt-in, halves xGMI bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-generate", action="store_true", help="skip the beam-4 captions/sec leg")
    ap.add_argument("--gen-batch", type=int, default=256)
    ap.add_argument("--generate-only", action="store_true", help="profiling aid: only the beam-4 leg")
    ap.add_argument("--small", action="store_true", help="reduced model (debug only; result is NOT the benchmark)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL; must be set before the HIP runtime starts
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MIC_BENCH_SHARE_GPU0"):  # debugging aid: all ranks on cuda:0, gradients over gloo (not a benchmark)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("MIC_BENCH_SHARE_GPU0"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import mic_amd  # noqa: F401
    from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, ops

    if args.small:
        cfg = CLIPVisionMBartConfig(mbart_config=dict(vocab_size=5003, d_model=256, decoder_layers=2, decoder_attention_heads=4, decoder_ffn_dim=512),
                                    clip_vision_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=64, patch_size=32))
    else:
        cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=dtype, device=dev)
    B, T = args.batch, 64
    lr_fn = create_learning_rate_fn(train_ds_size=10_000_000, train_batch_size=B * world, num_train_epochs=7, num_warmup_steps=1000, learning_rate=5e-5)
    tr = Trainer(model, lr_fn, seed=42, grad_comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else None)
    V, img = cfg.mbart_config.vocab_size, cfg.clip_vision_config.image_size
    batches = [synth_batch(B, T, V, img, 1234 + rank * 100 + i) for i in range(2)]
    # inputs resident in HBM before the timed region
    dbatches = [{k: torch.from_numpy(v).to(dev) for k, v in b.items()} for b in batches]
    from mic_amd import loss_rows

    for b, db in zip(batches, dbatches):  # collate-side: positions that carry loss (the LM head runs only there)
        idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
        db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.generate_only:
        print(json.dumps(bench_generate(model, cfg, dev, batch=args.gen_batch)))
        return

    for i in range(args.warmup):
        tr.train_step(dbatches[i % 2])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = tr.train_step(dbatches[i % 2])
    barrier()
    dt = time.perf_counter() - t0
    loss = float(out["loss"])
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    images_per_sec = world * B * args.steps / dt

    roofline = None
    if not args.no_roofline and rank != 0:
        tr.train_step(dbatches[0])  # the instrumented extra step below contains collectives: every rank takes part
        torch.cuda.synchronize()
    if not args.no_roofline and rank == 0:
        # dominant kernel = the bf16 MFMA GEMM (gemm_bf16_kernel): every launch of one extra, untimed step is bracketed
        # by HIP events on the launch stream; achieved = sum(2MNK) / sum(duration).
        recs = []
        orig = ops.gemm

        def timed_gemm(a, b, out, M, N, K, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(a, b, out, M, N, K, **kw)
            e1.record()
            recs.append((2.0 * M * N * K, e0, e1))
            return r

        orig_g = ops.gemm_grouped

        def timed_grouped(arg_list):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_g(arg_list)
            e1.record()
            recs.append((sum(2.0 * g.M * g.N * g.K for g in arg_list), e0, e1))
            return r

        ops.gemm = timed_gemm
        ops.gemm_grouped = timed_grouped
        tr.train_step(dbatches[0])
        torch.cuda.synchronize()
        ops.gemm, ops.gemm_grouped = orig, orig_g
        flops = sum(r[0] for r in recs)
        ms = sum(r[1].elapsed_time(r[2]) for r in recs)
        ach = flops / (ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "gemm_bf16_kernel" if dtype == torch.bfloat16 else "gemm_f32_kernel",
                    "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else 157.3, "unit": "TFLOP/s",
                    "frac": round(ach / (PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else 157.3), 4), "traffic": None,
                    "launches_per_step": len(recs), "gemm_ms_per_step": round(ms, 3),
                    "gemm_gflop_per_step": round(flops / 1e9, 1)}
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (a profiler cannot run inside this process); the
        # committed summary of the last such pass over this same command is reported, with its provenance.
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_train_pmc_hbm_traffic.json")
        if dtype == torch.bfloat16 and B == 64 and not args.small and os.path.exists(pmc):
            t = json.load(open(pmc))
            roofline["traffic"] = t["bytes_per_launch"]
            roofline["traffic_unit"] = "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, avg over the step's GEMM launches)"
            roofline["traffic_source"] = t["summary"]
    if world > 1:
        dist.barrier()

    gen = None
    if not args.no_generate:
        # generation is replicas-only (no collective): every rank decodes its own 256 images; report the sum
        g = bench_generate(model, cfg, dev, batch=args.gen_batch if not args.small else 8)
        if world > 1:
            tv = torch.tensor([g["value"]], dtype=torch.float64, device=dev)
            dist.all_reduce(tv, op=dist.ReduceOp.SUM)
            g["value"] = round(float(tv.item()), 1)
        g["n_gpus"] = world
        gen = g
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.small:
        try:
            cpu = cpu_baseline()
        except Exception as e:  # the baseline is a reported figure, never a dependency of the GPU number
            cpu = {"value": None, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}

    if rank == 0:
        step_flops = TRAIN_GFLOP_PER_SAMPLE * 1e9 * B
        line = {
            "metric": "train images/sec, ViT-B/32+mBART-50 (bf16 train step, batch 64/GPU, 224x224, seq_len 64)",
            "value": round(images_per_sec, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (random-init weights, N(0,1) pixels, ragged random captions)",
            "config": {"workload": "configs[1]: ViT-B/32 + mBART-large-50 train step (fwd+loss+bwd+all-reduce+AdamW), "
                                   f"per-GPU batch {B}, 224x224 NHWC fp32 pixels, seq_len {T}, dropout 0.1" + (" [SMALL DEBUG MODEL]" if args.small else ""),
                       "global_batch": B * world, "seq_len": T, "parallelism": f"dp{world}", "grad_allreduce": f"{args.grad_comm} flat buckets, RCCL, side stream",
                       "lm_head": "logits/CE on the label positions with loss mask 1 only (exact; ragged captions n~U{8..62})"},
            "model_tflops_per_gpu": round(step_flops * args.steps / dt / 1e12, 1),
            "final_loss": round(loss, 4),
            "roofline": roofline, "cpu_baseline": cpu, "beam4_generate": gen,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
