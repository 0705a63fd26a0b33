"""Per-shape breakdown of the train step's GEMM launches (HIP-event timed, one instrumented step).
usage: python tools/gemm_breakdown.py [--batch 64]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mic_amd  # noqa: E402
from mic_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, loss_rows

    cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
    model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=torch.bfloat16, device=dev)
    tr = Trainer(model, create_learning_rate_fn(10_000_000, a.batch, 7, 1000, 5e-5), seed=42)
    batches = []
    for i in range(2):
        b = bench.synth_batch(a.batch, 64, cfg.mbart_config.vocab_size, cfg.clip_vision_config.image_size, 1234 + i)
        db = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
        idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
        db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
        batches.append(db)
    for _ in range(2):
        tr.train_step(batches[0])
    torch.cuda.synchronize()
    recs = []
    og, ogg = ops.gemm, ops.gemm_grouped

    def tg(x, w, out, M, N, K, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = og(x, w, out, M, N, K, **kw)
        e1.record()
        lay = ("T" if kw.get("a_kmajor") else "N") + ("N" if kw.get("b_kmajor") else "T")
        fl = ",".join(k for k in ("bias", "act", "zout", "zin", "residual", "accumulate", "dropout_p", "split_k") if (torch.is_tensor(kw.get(k)) or kw.get(k)))
        recs.append(((M, N, K, lay, fl, str(out.dtype)[6:]), 2.0 * M * N * K, e0, e1))
        return r

    def tgg(lst):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = ogg(lst)
        e1.record()
        recs.append((("grouped", len(lst), max(g.K for g in lst), "", "", ""), sum(2.0 * g.M * g.N * g.K for g in lst), e0, e1))
        return r

    ops.gemm, ops.gemm_grouped = tg, tgg
    tr.train_step(batches[1])
    torch.cuda.synchronize()
    ops.gemm, ops.gemm_grouped = og, ogg
    agg = collections.OrderedDict()
    for key, fl, e0, e1 in recs:
        c = agg.setdefault(key, [0, 0.0, 0.0])
        c[0] += 1
        c[1] += e0.elapsed_time(e1) * 1e3
        c[2] += fl
    tot = sum(c[1] for c in agg.values())
    print(f"{'M':>6} {'N':>7} {'K':>7} lay {'flags':28s} {'out':8s} calls   avg_us   TF/s  ms/step")
    for key, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{key[0]!s:>6} {key[1]!s:>7} {key[2]!s:>7} {key[3]:3s} {key[4]:28s} {key[5]:8s} {n:5d} {us / n:8.1f} {fl / us / 1e6:6.0f} {us / 1e3:8.3f}")
    print(f"total {tot / 1e3:.3f} ms, {sum(c[2] for c in agg.values()) / tot / 1e6:.0f} TF/s")


if __name__ == "__main__":
    main()
