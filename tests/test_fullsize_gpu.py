"""Full-size parity (ViT-B/32 + mBART-large-50 shapes: d=1024/768, 12+12 layers, V=250054 -> Vpad 250112) on the GPU box:
HIP path vs the CPU oracle on the same seeded random weights and inputs.  Small batch (B=2, T=64) so the oracle finishes
in seconds on the box's host cores; exercises the real tile counts, the vocab tail and the 64-bit offsets of the 2 GB
logits buffer that the reduced-config tests cannot."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(dev):
    from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration
    from mic_amd.params import unflatten_tree
    from oracle import model_ref as M

    torch.set_num_threads(min(64, torch.get_num_threads()))
    rc = M.RefConfig(gelu="tanh", decoder_ln_eps=1e-6)
    p = M.init_params(rc, seed=7, perturb_ln=True)
    cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
    tree = unflatten_tree({k: v.numpy() for k, v in p.items()})
    models = {}
    for dt in (torch.float32, torch.bfloat16):
        m = FlaxCLIPVisionMBartForConditionalGeneration(cfg, dtype=dt, device=dev)
        m.params = tree
        models[dt] = m
    g = torch.Generator().manual_seed(5)
    B, T = 2, 64
    px = torch.randn(B, 224, 224, 3, generator=g).clamp(-1.8, 2.2)
    labels = torch.full((B, T), 1, dtype=torch.int64)
    mask = torch.zeros((B, T), dtype=torch.int64)
    for b, n in enumerate((62, 23)):
        labels[b, 0] = 250004 + b
        labels[b, 1:1 + n] = torch.randint(4, 250000, (n,), generator=g)
        labels[b, 1 + n] = 2
        mask[b, :n + 2] = 1
    labels[0, 5] = 250053  # last vocabulary row (tail of the padded table)
    dec_in = torch.full_like(labels, 1)
    dec_in[:, 1:] = labels[:, :-1]
    with torch.no_grad():
        ref_logits = M.forward_logits(rc, p, px, dec_in, mask)
    return rc, p, models, (px, labels, mask, dec_in), ref_logits


def test_fullsize_logits(full):
    rc, p, models, (px, labels, mask, dec_in), ref = full
    valid = mask.bool()
    for dt, (tmax, tmean) in ((torch.float32, (3e-4, 3e-5)), (torch.bfloat16, (4e-2, 5e-3))):
        out = models[dt](px.numpy(), dec_in.numpy(), mask.numpy())[0]
        assert tuple(out.shape) == (2, 64, 250054)
        d = (out.float().cpu() - ref)[valid].abs()
        s = ref[valid].abs().max()
        assert (d.max() / s).item() < tmax and (d.mean() / s).item() < tmean, (dt, (d.max() / s).item(), (d.mean() / s).item())


def test_fullsize_bf16_every_stage_within_1e3_of_bf16_storage_oracle(full):
    """north-star "logits within 1e-3 (bf16)" at ViT-B/32 + mBART-large-50 size, asserted where it can hold: EVERY stored tensor
    of the bf16 forward pass (261 stages from im2col to the 250 054-wide logits) lies within its own final rounding + 1e-3 of
    the stage's scale of the bf16-storage oracle's restatement of that stage, evaluated on the pass's own stored inputs
    (tests/util_bf16_stages.py explains why end-to-end agreement at 1e-3 is not a property of the format)."""
    from util_bf16_stages import report, run_stages

    rc, p, models, (px, labels, mask, dec_in), _ = full
    stages = run_stages(models[torch.bfloat16], rc, p, px, dec_in, mask)
    msg, bad = report(stages, 1e-3)
    print("[bf16 stages, full size]", msg)
    assert len(stages) == 4 + 8 * 12 + 1 + 2 + 13 * 12 + 2 and not bad, (msg, bad[:5])
    assert max(s[3] for s in stages) < 0.02


def _tail(d, s):
    """(99.9th percentile, rms) of |d| / s: the maximum over millions of elements is an extreme-value statistic — 3.1e-3 in one
    evaluation order, 6.2e-3 in the next — the upper tail and the rms are properties of the error distribution"""
    f = (d.flatten() / s).float()
    if f.numel() > 4_000_000:
        f = f[:: f.numel() // 4_000_000 + 1]
    return torch.quantile(f, 0.999).item(), f.pow(2).mean().sqrt().item()


def _self_distance(fn):
    """distance between two valid evaluation orders of the bf16-storage oracle: fp32 vs float64 accumulation of every Linear"""
    from oracle import model_ref_bf16 as E

    with torch.no_grad():
        a = fn()
        E.ACC64 = True
        try:
            b = fn()
        finally:
            E.ACC64 = False
    return a, b


def test_fullsize_bf16_logits_as_close_to_the_bf16_storage_oracle_as_the_oracle_is_to_itself(full):
    """end to end the stored bf16 logits cannot agree with ANY bf16 oracle to 1e-3: the oracle evaluated with float64 instead of
    fp32 accumulation (an fp32-ulp-sized perturbation) already differs from itself by the bf16 resolution after 24 layers of
    rounding points.  Asserted: the HIP path is no further from the oracle than that self-distance — mean within 1.5x, maximum
    within one bf16 ulp of the largest logit (2^-7 of the scale; maxima are single rounding flips, quantised in ulps) + 1e-3 —
    i.e. it is one more valid evaluation order of the same bf16 arithmetic; and the self-distance is indeed above 1e-3."""
    from oracle import model_ref_bf16 as E

    rc, p, models, (px, labels, mask, dec_in), _ = full
    ref, ref64 = _self_distance(lambda: E.forward_logits(rc, p, px, dec_in, mask))
    out = models[torch.bfloat16](px.numpy(), dec_in.numpy(), mask.numpy())[0]
    valid = mask.bool()
    got = out[valid.to(out.device)].float().cpu()
    s = ref[valid].abs().max()
    d_hip = (got - E.rb(ref[valid])).abs()
    d_self = (E.rb(ref64[valid]) - E.rb(ref[valid])).abs()
    hip = ((d_hip.max() / s).item(), (d_hip.mean() / s).item())
    own = ((d_self.max() / s).item(), (d_self.mean() / s).item())
    print(f"[fullsize bf16 logits, stored vs stored] HIP vs oracle: max {hip[0]:.2e} mean {hip[1]:.2e}; oracle(f32 acc) vs oracle(f64 acc): "
          f"max {own[0]:.2e} mean {own[1]:.2e}")
    t_hip, t_own = _tail(d_hip, s), _tail(d_self, s)
    print(f"[fullsize bf16 logits] (q99.9, rms) of |d| / scale: HIP vs oracle {t_hip[0]:.2e} {t_hip[1]:.2e}, oracle vs itself {t_own[0]:.2e} {t_own[1]:.2e}")
    assert own[0] > 1e-3, own
    assert hip[0] < 2.0 ** -7 + 1e-3 and hip[1] < 1.5 * own[1], (hip, own)
    assert t_hip[0] < 1.5 * t_own[0] and t_hip[1] < 1.5 * t_own[1], (t_hip, t_own)  # (measured 1.0x; a subsampled tail statistic: real margin)


@pytest.mark.parametrize("fold", [False, True])
def test_fullsize_bf16_cached_decode_as_close_to_the_bf16_storage_oracle_as_the_oracle_is_to_itself(full, fold):
    """four cached decoder steps at full size, explicit-LayerNorm launches and LayerNorm-folded GEMMs separately, each against the
    restatement of ITS arithmetic; same criterion as the teacher-forced test above (12 layers of rounding points per step).
    Measured: the teacher-forced pass sits AT the oracle's self-distance (q99.9 2.22e-3 / rms 6.7e-4 both), the cached steps 1.2x above
    it (2.31e-3 / 7.2e-4 against 1.93e-3 / 5.9e-4): the self-distance varies ONE thing (fp32 vs float64 accumulation in the Linears),
    the decode kernels differ from the oracle's order in several (row statistics as 2^20 fixed-point sums, exp2-based softmax over the
    cache, the fused q/k/v launch) — each a valid evaluation order of the same bf16 arithmetic, together a slightly wider spread."""
    from oracle import model_ref_bf16 as E

    rc, p, models, (px, labels, mask, dec_in), _ = full
    model = models[torch.bfloat16]
    keep = model.engine.decode_ln_fold
    model.engine.decode_ln_fold = fold
    try:
        pc = E.compute_copy(p)
        enc = model.encode(px.numpy(), _int32_cast=False)
        ehs_hip = enc.last_hidden_state.float().cpu()
        B, S = 2, 4
        ids = dec_in[:, :S]

        def oracle_steps():
            ckv = E.cross_kv(rc, pc, ehs_hip)
            st = E.DecodeState(rc, B, S + 2)
            return torch.stack([E.decode_step(rc, pc, st, ids[:, t:t + 1], torch.full((B, 1), t), ehs_hip, ln_fold=fold, cross_kv=ckv)[:, 0]
                                for t in range(S)], 1)

        ref, ref64 = _self_distance(oracle_steps)
        cache = model.init_cache(B, S + 2, enc)
        got = []
        for t in range(S):
            out = model.decode(ids[:, t:t + 1].numpy(), enc, decoder_position_ids=np.full((B, 1), t), past_key_values=cache)
            cache = out.past_key_values
            got.append(out.logits[:, 0].float().cpu())
        got = torch.stack(got, 1)
        s = ref.abs().max()
        d_hip, d_self = (got - E.rb(ref)).abs(), (E.rb(ref64) - E.rb(ref)).abs()
        hip, own = ((d_hip.max() / s).item(), (d_hip.mean() / s).item()), ((d_self.max() / s).item(), (d_self.mean() / s).item())
        t_hip, t_own = _tail(d_hip, s), _tail(d_self, s)
        print(f"[fullsize bf16 cached decode fold={fold}] HIP vs oracle: max {hip[0]:.2e} mean {hip[1]:.2e} q99.9 {t_hip[0]:.2e} rms {t_hip[1]:.2e}; "
              f"oracle vs itself (f64 acc): max {own[0]:.2e} mean {own[1]:.2e} q99.9 {t_own[0]:.2e} rms {t_own[1]:.2e}")
        assert hip[0] < 2.0 ** -7 + 1e-3 and hip[1] < 1.5 * own[1], (hip, own)
        # the maximum over 2 M logits is an extreme-value statistic (round 4 read 6.2e-3 here against a self-distance of 3.1e-3, and
        # 5.9e-3 against 5.9e-3 in the teacher-forced test): what is pinned is the error DISTRIBUTION — its upper tail and its rms
        # are the oracle's own
        assert t_hip[0] < 1.5 * t_own[0] and t_hip[1] < 1.5 * t_own[1], (t_hip, t_own)  # (measured 1.20x / 1.22x)
    finally:
        model.engine.decode_ln_fold = keep


_REF = {}


def _ref_loss_and_grads(full):
    """the fp32 oracle's loss and gradients on the fixture batch (autograd over oracle.model_ref at full size: computed once)"""
    from oracle import train_ref

    if "g" not in _REF:
        rc, p, models, (px, labels, mask, dec_in), _ = full
        _REF["g"] = train_ref.loss_and_grads(rc, p, px, labels, mask, dec_in)
    return _REF["g"]


def test_fullsize_fp8_train_step_against_fp32_oracle(full):
    """BASELINE configs[4] at ViT-B/32 + mBART-large-50 size: the QKV / FFN projections of both towers as e4m3 / e5m2 GEMMs
    (forward, dX, dW; per-tensor scaling, first pass = current amax) against the fp32 oracle — loss within 3e-2, every large
    gradient leaf with cosine > 0.9, all leaves together > 0.97 (the tolerances of the reduced-size test, tests/test_fp8_gpu.py)."""
    from mic_amd import loss_rows

    rc, p, models, (px, labels, mask, dec_in), _ = full
    ref_loss, ref_g = _ref_loss_and_grads(full)
    model = models[torch.bfloat16]
    eng = model.engine
    eng.set_gemm_dtype("fp8")
    try:
        d = model._dev
        B, T = labels.shape
        pos = torch.arange(T, dtype=torch.int32, device=model.device)[None].expand(B, T).contiguous()
        idx, rl = loss_rows(mask.numpy(), labels.numpy())
        loss = eng.loss_and_grads(d(px, torch.float32), d(dec_in, torch.int32).reshape(-1), pos.reshape(-1), d(mask, torch.int32),
                                  d(labels, torch.int32).reshape(-1), B, T, rows=(d(idx, torch.int32), len(idx)), row_labels=d(rl, torch.int32))
        torch.cuda.synchronize()
        assert abs(loss.item() - ref_loss.item()) < 3e-2 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
        got = model.store.export_flat("grad")
        cos = lambda a, b: torch.nn.functional.cosine_similarity(a.reshape(-1).double(), b.reshape(-1).double(), dim=0).item()
        worst = {k: cos(torch.from_numpy(got[k]).reshape(rg.shape), rg) for k, rg in ref_g.items() if rg.abs().max().item() > 1e-5 and rg.numel() >= 4096}
        bad = {k: v for k, v in worst.items() if v < 0.9}
        allc = cos(torch.cat([torch.from_numpy(got[k]).reshape(-1) for k in ref_g]), torch.cat([v.reshape(-1) for v in ref_g.values()]))
        print(f"[fullsize fp8] loss {loss.item():.4f} vs {ref_loss.item():.4f}; worst leaf cosine {min(worst.values()):.3f} over {len(worst)} leaves; all leaves {allc:.4f}")
        assert not bad, sorted(bad.items(), key=lambda kv: kv[1])[:6]
        assert allc > 0.97, allc
        assert eng._w8["dec11.fc1"][0].dtype == torch.float8_e4m3fn and float(eng._w8["dec11.fc1"][2][1]) > 0  # the fp8 copies carry a scale
        # second pass, same batch and weights: every tensor now has a scale history, so the producers emit the fp8 operands
        # themselves (fused emission) under scales equal to the first pass's current ones — same bytes, same loss and gradients up to
        # the order of the fp32 atomics
        assert eng.fp8_fused and not eng._a8_ready
        loss2 = eng.loss_and_grads(d(px, torch.float32), d(dec_in, torch.int32).reshape(-1), pos.reshape(-1), d(mask, torch.int32),
                                   d(labels, torch.int32).reshape(-1), B, T, rows=(d(idx, torch.int32), len(idx)), row_labels=d(rl, torch.int32))
        torch.cuda.synchronize()
        assert len(eng._a8_ready) > 150 and abs(loss2.item() - loss.item()) < 1e-5 * abs(loss.item()), (loss2.item(), loss.item())
        got2 = model.store.export_flat("grad")
        for k in ref_g:
            a, b2 = got2[k].astype(np.float64), got[k].astype(np.float64)
            assert np.abs(a - b2).max() <= 2e-4 * max(np.abs(b2).max(), 1e-30), (k, np.abs(a - b2).max(), np.abs(b2).max())
    finally:
        eng.set_gemm_dtype(None)
        eng.free_buffers()


def test_fullsize_loss_and_gradients_f32(full):
    from mic_amd import loss_rows

    rc, p, models, (px, labels, mask, dec_in), _ = full
    model = models[torch.float32]
    ref_loss, ref_g = _ref_loss_and_grads(full)
    d = model._dev
    B, T = labels.shape
    pos = torch.arange(T, dtype=torch.int32, device=model.device)[None].expand(B, T).contiguous()
    idx, rl = loss_rows(mask.numpy(), labels.numpy())
    loss = model.engine.loss_and_grads(d(px, torch.float32), d(dec_in, torch.int32).reshape(-1), pos.reshape(-1), d(mask, torch.int32),
                                       d(labels, torch.int32).reshape(-1), B, T, rows=(d(idx, torch.int32), len(idx)),
                                       row_labels=d(rl, torch.int32))
    torch.cuda.synchronize()
    assert abs(loss.item() - ref_loss.item()) < 5e-5 * abs(ref_loss.item())
    got = model.store.export_flat("grad")
    for k in ("model/shared/embedding", "final_logits_bias", "model/decoder/layers/11/fc2/kernel", "model/decoder/layers/0/self_attn/q_proj/kernel",
              "model/decoder/layers/5/encoder_attn/v_proj/kernel", "model/decoder/embed_positions/embedding", "model/visual_projection/kernel",
              "model/encoder/vision_model/encoder/layers/0/mlp/fc1/kernel", "model/encoder/vision_model/embeddings/patch_embedding/kernel",
              "model/encoder/vision_model/embeddings/class_embedding", "model/decoder/layers/3/final_layer_norm/scale"):
        rg = ref_g[k]
        e = ((torch.from_numpy(got[k]).reshape(rg.shape) - rg).abs().max() / rg.abs().max()).item()
        assert e < 1e-3, (k, e)


def test_fullsize_greedy_ids_exact_f32(full):
    from oracle import generation_ref as G
    from oracle import model_ref as M

    rc, p, models, (px, *_), _ = full
    with torch.no_grad():
        ehs, _ = M.encode(rc, p, px)
    L = 6
    kw = dict(max_length=L, num_beams=1, forced_bos_token_id=250004)
    ref = G.generate(lambda rows: G.ModelStepper(rc, p, ehs, L), 2, G.GenDefaults(), **kw)
    out = models[torch.float32].generate(px.numpy(), **kw)
    assert np.array_equal(out.sequences.cpu().numpy(), ref)
    kw = dict(max_length=5, num_beams=4, forced_bos_token_id=250008)
    ref = G.generate(lambda rows: G.ModelStepper(rc, p, ehs.repeat_interleave(4, 0), 5), 2, G.GenDefaults(), **kw)
    out = models[torch.float32].generate(px.numpy(), **kw)
    assert np.array_equal(out.sequences.cpu().numpy(), ref.sequences)
    assert np.allclose(out.scores.cpu().numpy(), ref.scores, rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- BASELINE.json sizes: size-independent properties
def _train_batch(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    px = torch.randn(B, 224, 224, 3, generator=g).clamp(-1.8, 2.2)
    labels = torch.full((B, T), 1, dtype=torch.int64)
    mask = torch.zeros((B, T), dtype=torch.int64)
    for b in range(B):
        n = int(torch.randint(8, 63, (1,), generator=g))
        labels[b, 0] = 250003 + (b % 4)
        labels[b, 1:1 + n] = torch.randint(4, 250000, (n,), generator=g)
        labels[b, 1 + n] = 2
        mask[b, :n + 2] = 1
    dec_in = torch.full_like(labels, 1)
    dec_in[:, 1:] = labels[:, :-1]
    return px, labels, mask, dec_in


def _grads(model, px, labels, mask, dec_in, compact):
    from mic_amd import loss_rows

    d = model._dev
    B, T = labels.shape
    pos = torch.arange(T, dtype=torch.int32, device=model.device)[None].expand(B, T).contiguous()
    kw = {}
    if compact:
        idx, rl = loss_rows(mask.numpy(), labels.numpy())
        kw = dict(rows=(d(idx, torch.int32), len(idx)), row_labels=d(rl, torch.int32))
    loss = model.engine.loss_and_grads(d(px, torch.float32), d(dec_in, torch.int32).reshape(-1), pos.reshape(-1), d(mask, torch.int32),
                                       d(labels, torch.int32).reshape(-1), B, T, **kw)
    torch.cuda.synchronize()
    return float(loss), model.store.grad.clone()


def test_baseline_train_batch64_properties_bf16(full):
    """configs[1] (bf16, batch 64, seq 64, full model): (a) the compacted LM head (only rows with loss mask 1) gives the same
    loss and gradients as the dense head; (b) data-parallel linearity — the batch gradient is the token-weighted mean of the
    two half-batch gradients (what the per-rank masked mean + gradient mean of main.py:679, 698 relies on, up to the
    per-rank normalisation)."""
    rc, p, models, _, _ = full
    model = models[torch.bfloat16]
    px, labels, mask, dec_in = _train_batch(64, 64, seed=11)
    l_c, g_c = _grads(model, px, labels, mask, dec_in, compact=True)
    l_d, g_d = _grads(model, px, labels, mask, dec_in, compact=False)
    assert abs(l_c - l_d) < 2e-6 * abs(l_d)
    st = model.store
    for name in ("shared", "flb", "dec11.fc2.w", "dec0.qkv.w", "vit0.fc1.w", "patch.w", "dec.ln_f.g"):
        s = st.segs[name]
        a, b = g_c[s.offset: s.offset + s.numel], g_d[s.offset: s.offset + s.numel]
        # bf16 activation gradients (one rounding of dh differs), dW row order.  The head's dX GEMM splits K by the row-tile count
        # (6 slabs on the 2.4 k compacted rows, 4 on all 4096): two fp32 summation orders, a handful of dh elements round the other
        # way in bf16, and the flips travel down 24 layers of bf16 gradients — the far end of the chain (the ViT's first layers) sees
        # 1.1e-2 of its largest entry where rounds 1-4 (32 slabs either way) saw 0.8e-2
        assert ((a - b).abs().max() / b.abs().max()).item() < (2e-2 if name in ("patch.w", "vit0.fc1.w") else 1e-2), name
    h = 32
    la, ga = _grads(model, px[:h], labels[:h], mask[:h], dec_in[:h], compact=True)
    lb, gb = _grads(model, px[h:], labels[h:], mask[h:], dec_in[h:], compact=True)
    na, nb = int(mask[:h].sum()), int(mask[h:].sum())
    assert abs((na * la + nb * lb) / (na + nb) - l_c) < 2e-4 * abs(l_c)  # bf16: tile configuration (summation order) varies with the row count
    gm = (na * ga + nb * gb) / (na + nb)
    for name in ("shared", "dec11.fc2.w", "dec0.qkv.w", "vit0.fc1.w", "vp.w"):
        s = st.segs[name]
        a, b = gm[s.offset: s.offset + s.numel], g_c[s.offset: s.offset + s.numel]
        # bf16 dlogits are scaled by 1/n_half vs 1/n_batch before rounding and flow back through bf16 activations gradients:
        # percent-level noise relative to the largest entry; a normalisation or reduction bug would be O(1)
        assert ((a - b).abs().max() / b.abs().max()).item() < 5e-2, name
        assert (torch.nn.functional.cosine_similarity(a, b, dim=0)).item() > 0.9995, name


def test_baseline_beam4_batch256_properties_bf16(full):
    """configs[3] (beam 4, batch 256, max_len 64, forced BOS, KV-cached): run-to-run determinism, permutation equivariance
    over the batch (every row is computed independently of its neighbours), and the structure gen:665-990 guarantees."""
    rc, p, models, _, _ = full
    model = models[torch.bfloat16]
    st = model.store
    st.f32("flb")[rc.eos_token_id] = -1e9  # as bench.py: exactly 63 decoder steps
    st.refresh_lp()
    try:
        g = torch.Generator().manual_seed(21)
        px = torch.randn(256, 224, 224, 3, generator=g).clamp(-1.8, 2.2)
        kw = dict(max_length=64, num_beams=4, forced_bos_token_id=250008)
        a = model.generate(px, **kw)
        b = model.generate(px, **kw)
        assert a["steps"] == 63 and torch.equal(a.sequences, b.sequences) and torch.equal(a.scores, b.scores)
        perm = torch.randperm(256, generator=g)
        c = model.generate(px[perm], **kw)
        assert torch.equal(c.sequences.cpu(), a.sequences.cpu()[perm]) and torch.equal(c.scores.cpu(), a.scores.cpu()[perm])
        seq = a.sequences.cpu()
        assert tuple(seq.shape) == (256, 64) and (seq[:, 0] == 2).all() and (seq[:, 1] == 250008).all()
        assert (seq[:, 63] == rc.eos_token_id).all()           # ForcedEOS at the last step overrides the -1e9 bias
        assert ((seq[:, 2:63] != rc.eos_token_id) & (seq[:, 2:63] != rc.pad_token_id)).all()
        assert (a.scores.cpu() < 0).all() and torch.isfinite(a.scores.cpu()).all()
    finally:
        st.f32("flb")[rc.eos_token_id] = 0.0
        st.refresh_lp()


# ---------------------------------------------------------------- the benchmarked dtype (bf16) pinned to the fp32 oracle
@pytest.fixture(scope="module")
def decisive(dev):
    """Full-size model whose logits have a trained-model-like spread: random init with the tied embedding scaled by 4
    (logit std ~2.5 instead of ~0.6), so that an argmax / top-k decision has a margin worth comparing across precisions
    (with the plain random init 250 054 nearly-tied logits make every precision, fp32 summation order included, pick
    different tokens)."""
    from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration
    from mic_amd.params import unflatten_tree
    from oracle import model_ref as M

    torch.set_num_threads(min(64, torch.get_num_threads()))
    rc = M.RefConfig(gelu="tanh", decoder_ln_eps=1e-6)
    p = M.init_params(rc, seed=11, perturb_ln=True)
    p["model/shared/embedding"] = p["model/shared/embedding"] * 4.0
    cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
    tree = unflatten_tree({k: v.numpy() for k, v in p.items()})
    models = {}
    for dt in (torch.float32, torch.bfloat16):
        m = FlaxCLIPVisionMBartForConditionalGeneration(cfg, dtype=dt, device=dev)
        m.params = tree
        models[dt] = m
    g = torch.Generator().manual_seed(15)
    px = torch.randn(8, 224, 224, 3, generator=g).clamp(-1.8, 2.2)
    with torch.no_grad():
        ehs, _ = M.encode(rc, p, px)
    return rc, p, models, px, ehs


def _prefix_len(a, b):
    ne = np.nonzero(a != b)[0]
    return len(a) if ne.size == 0 else int(ne[0])


def test_fullsize_generate_to_max_length_64_vs_oracle(decisive):
    """configs[3]'s call shape at full model size, batch 8, to max_length 64 (63 decoder steps), greedy and beam-4 with a
    forced BOS language code, against the oracle's restatement of gen:422-535 / 665-990 on the CPU model oracle.
      float32 mode: token ids bit-exact over all 64 positions, beam scores within 1e-4 relative.
      bfloat16 mode (what bench.py times): the same call; one different decision changes the rest of that caption, so the
      floors are on (i) the tokens before the first divergence and (ii) the beam score of the returned hypothesis."""
    from oracle import generation_ref as G

    rc, p, models, px, ehs = decisive
    B, L = px.shape[0], 64
    res = {}
    for K in (1, 4):
        kw = dict(max_length=L, num_beams=K, forced_bos_token_id=250004 + K)
        ref = G.generate(lambda rows: G.ModelStepper(rc, p, ehs.repeat_interleave(K, 0) if K > 1 else ehs, L), B, G.GenDefaults(), **kw)
        ref_seq = ref if K == 1 else ref.sequences
        out32 = models[torch.float32].generate(px.numpy(), **kw)
        got32 = out32.sequences.cpu().numpy()
        assert got32.shape == (B, L) and np.array_equal(got32, ref_seq), (K, np.argwhere(got32 != ref_seq)[:5])
        if K > 1:
            assert out32["steps"] == ref.steps
            assert np.allclose(out32.scores.cpu().numpy(), ref.scores, rtol=1e-4, atol=1e-4)
        out16 = models[torch.bfloat16].generate(px.numpy(), **kw)
        got16 = out16.sequences.cpu().numpy()
        agree = float((got16 == ref_seq).mean())
        prefix = [_prefix_len(got16[b], ref_seq[b]) for b in range(B)]
        res[K] = (agree, prefix)
        print(f"bf16 vs fp32 oracle, num_beams={K}: token agreement {agree:.3f}, common prefix per caption {prefix}")
        assert (got16[:, 0] == 2).all() and (got16[:, 1] == 250004 + K).all()
        assert min(prefix) >= 2 and float(np.mean(prefix)) >= BF16_MIN_MEAN_PREFIX[K], (K, prefix)
        assert agree >= BF16_MIN_AGREEMENT[K], (K, agree)
        if K > 1:
            s16, s32 = out16.scores.cpu().numpy(), ref.scores
            rel = np.abs(s16 - s32) / np.abs(s32)
            print(f"   beam scores: max relative difference {rel.max():.4f}")
            assert rel.max() < 1e-2, rel  # length-normalised log-probability of the returned hypothesis


# floors for the bf16 leg above, set from the measured values (printed by the test) with margin
# measured on MI355X (round 2): agreement 1.000 and 64-token common prefixes for all 8 captions, greedy and beam-4, score
# differences < 1e-4 relative.  NOTE what that does and does not show: a randomly initialised tied-embedding model copies the
# fed token with a wide margin, so these captions exercise the whole 63-step pipeline at full size (cache indirection, beam
# bookkeeping, 250 054-wide top-k, forced tokens) but not close calls; close calls are covered teacher-forced in
# test_fullsize_bf16_logit_error_absolute and on a TRAINED reduced model in test_generate_gpu.py.
BF16_MIN_MEAN_PREFIX = {1: 60.0, 4: 60.0}
BF16_MIN_AGREEMENT = {1: 0.95, 4: 0.95}


def test_fullsize_bf16_logit_error_absolute(decisive, full):
    """North-star: "logits within 1e-3 (bf16)".  Measured against the fp32 oracle at full size, teacher-forced, absolute:
    bf16 storage rounds a logit of magnitude 2..4 to a multiple of 2^-6 = 0.0156 (half an ulp = 7.8e-3 > 1e-3) before any
    arithmetic error, so 1e-3 absolute is not representable in this dtype; what IS asserted: the error stays within a few ulps
    of the stored format and the top-1 decision agrees where the fp32 margin exceeds the measured error."""
    from oracle import model_ref as M

    for name, (rc, p, models, px) in (("decisive", decisive[:4]), ("plain-init", (full[0], full[1], full[2], full[3][0]))):
        g = torch.Generator().manual_seed(3)
        B, T = 2, 64
        pxx = px[:B]
        ids = torch.randint(4, 250000, (B, T), generator=g)
        mask = torch.ones(B, T, dtype=torch.int64)
        with torch.no_grad():
            ref = M.forward_logits(rc, p, pxx, ids, mask)
        out = models[torch.bfloat16](pxx.numpy(), ids.numpy(), mask.numpy())[0].float().cpu()
        d = (out - ref).abs()
        scale = ref.abs().max().item()
        ulp = 2.0 ** (np.floor(np.log2(scale)) - 7)  # bf16 spacing at the largest logit
        top_ref = ref.topk(2, dim=-1)
        margin = (top_ref.values[..., 0] - top_ref.values[..., 1])
        same = (out.argmax(-1) == top_ref.indices[..., 0])
        clear = margin > 4 * d.max()
        print(f"{name}: |logit| max {scale:.3f} (bf16 ulp there {ulp:.4f}); abs err max {d.max():.4f} mean {d.mean():.5f} "
              f"= {d.max() / scale:.2e} / {d.mean() / scale:.2e} of scale; top-1 agreement {same.float().mean():.3f}")
        assert d.max().item() < 6 * ulp and d.mean().item() < 0.6 * ulp, (name, d.max().item(), d.mean().item(), ulp)
        assert bool(same[clear].all())


def test_baseline_train_batch64_linearity_equal_token_halves_bf16(full):
    """Data-parallel linearity at configs[1] size with the per-rank normalisation taken out of the comparison: both half
    batches carry the same number of loss tokens, so 1/n_half = 2/n_batch is an exact power-of-two factor on every bf16
    dlogit and the mean of the two half-batch gradients must reproduce the full-batch gradient up to summation order —
    a much tighter bound than the 5e-2 the unequal halves need."""
    rc, p, models, _, _ = full
    model = models[torch.bfloat16]
    px, labels, mask, dec_in = _train_batch(64, 64, seed=12)
    # rows 32..63 get the caption lengths of rows 0..31 (fresh tokens): equal token counts per half
    g = torch.Generator().manual_seed(99)
    for b in range(32):
        n = int(mask[b].sum()) - 2
        labels[32 + b] = 1
        mask[32 + b] = 0
        labels[32 + b, 0] = 250003 + (b % 4)
        labels[32 + b, 1:1 + n] = torch.randint(4, 250000, (n,), generator=g)
        labels[32 + b, 1 + n] = 2
        mask[32 + b, :n + 2] = 1
    dec_in = torch.full_like(labels, 1)
    dec_in[:, 1:] = labels[:, :-1]
    assert int(mask[:32].sum()) == int(mask[32:].sum())
    l_c, g_c = _grads(model, px, labels, mask, dec_in, compact=True)
    la, ga = _grads(model, px[:32], labels[:32], mask[:32], dec_in[:32], compact=True)
    lb, gb = _grads(model, px[32:], labels[32:], mask[32:], dec_in[32:], compact=True)
    assert abs((la + lb) / 2 - l_c) < 1e-4 * abs(l_c)
    gm = (ga + gb) / 2  # pmean of the per-rank gradients (main.py:698)
    st = model.store
    worst = {}
    for name in ("shared", "flb", "dec11.fc2.w", "dec6.cq.w", "dec0.qkv.w", "vit11.fc1.w", "vit0.fc1.w", "vp.w", "patch.w"):
        s = st.segs[name]
        a, b = gm[s.offset: s.offset + s.numel], g_c[s.offset: s.offset + s.numel]
        worst[name] = ((a - b).abs().max() / b.abs().max()).item()
        assert torch.nn.functional.cosine_similarity(a, b, dim=0).item() > 0.9995, name
    print("equal-token halves, max |mean of halves - batch| / max|batch| per segment:", {k: round(v, 5) for k, v in worst.items()})
    assert max(worst.values()) < LINEARITY_TOL, worst


LINEARITY_TOL = 2.5e-2  # measured worst segment 1.6e-2 (dec6.cq.w, patch.w); the unequal-halves test above needs 5e-2


def test_fullsize_bf16_decisions_after_training_follow_fp32_wherever_the_margin_allows(full):
    """The round-2 review's point: full-size bf16 / fp32 token agreement was shown on a randomly initialised model, whose top-1 is
    the copy of the fed token by a wide margin.  Here the full-size model is TRAINED (bf16 trainer, packed rows, AdamW) on the
    synthetic rule task of tests/test_generate_gpu.py and evaluated at two points: HALF-TRAINED (the first step whose loss is below
    4: the copy behaviour is gone, the rule is not learned yet — predictions are input-dependent and uncertain, i.e. close calls)
    and TRAINED (60 steps, loss 0.02-0.1).  At both points the weights are evaluated teacher-forced by the fp32 HIP path (bit-checked
    against the oracle elsewhere in this file) and the bf16 HIP path.  Asserted: the bf16 top-1 equals the fp32 top-1 at every
    position whose fp32 top-1 / top-2 margin exceeds 4x the measured bf16 logit error; the trained model's greedy captions are
    identical in both precisions."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_generate_gpu import _rule_batch

    from mic_amd import Trainer, create_learning_rate_fn
    from mic_amd.params import unflatten_tree

    rc, p, models, _, _ = full
    m16, m32 = models[torch.bfloat16], models[torch.float32]
    ex, elabels, emask, edec = _rule_batch(rc, list(range(8)))

    def compare(tag):
        m32.params = m16.params
        l32 = m32(ex.numpy(), edec.numpy(), emask.numpy())[0].float().cpu()
        l16 = m16(ex.numpy(), edec.numpy(), emask.numpy())[0].float().cpu()
        valid = emask.bool()
        top2 = l32.topk(2, dim=-1).values
        margin = (top2[..., 0] - top2[..., 1])[valid]
        err = (l16 - l32).abs().amax(-1)[valid]
        same = (l16.argmax(-1) == l32.argmax(-1))[valid]
        safe = margin > 4 * err
        copied = (l32.argmax(-1) == edec)[valid].float().mean().item()
        print(f"[fullsize {tag}] fp32 top-1 copies the fed token at {copied:.2f} of the positions; margins: median {margin.median():.3f}, "
              f"{int((margin < 4 * err).sum())} of {margin.numel()} positions within 4x the bf16 error (max error {err.max():.3f}); top-1 agreement "
              f"{same.float().mean():.3f} overall, {same[safe].float().mean():.3f} over the {int(safe.sum())} positions with margin > 4x error")
        assert copied < 0.5 and bool(same[safe].all()) and int(safe.sum()) >= 8
        return float(same.float().mean())

    try:
        tr = Trainer(m16, create_learning_rate_fn(10_000, 1, 1, 10, 2e-4), seed=5)
        g = torch.Generator().manual_seed(0)
        half = None
        for step in range(60):
            cls = torch.randint(0, 8, (32,), generator=g).tolist()
            px, labels, mask, dec_in = _rule_batch(rc, cls)
            out = tr.train_step({"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()})
            if half is None and float(out["loss"]) < 4.0:
                half = (step, float(out["loss"]))
                print(f"[fullsize half-trained] step {step}: loss {half[1]:.2f}")
                compare("half-trained")
        loss = float(out["loss"])
        assert half is not None and np.isfinite(loss) and loss < 0.5, (half, loss)  # 0.02-0.1 from run to run (fp32 atomics in training)
        compare("trained")
        kw = dict(max_length=12, num_beams=1, decoder_start_token_id=rc.vocab_size - 10, forced_eos_token_id=None)
        s32 = m32.generate(ex.numpy(), **kw).sequences.cpu().numpy()
        s16 = m16.generate(ex.numpy(), **kw).sequences.cpu().numpy()
        print(f"[fullsize trained] loss {loss:.3f}; greedy: {int((s32 == s16).all(1).sum())}/8 captions identical in bf16 and fp32; they follow the rule "
              f"on {(s32[:, 1:9] == elabels[:, 1:9].numpy()).mean():.2f} of the positions")
        assert np.array_equal(s32, s16)
    finally:  # the module's other tests use the fixture's original weights
        tree = unflatten_tree({k: v.numpy() for k, v in p.items()})
        m16.params = tree
        m32.params = tree
        m16.engine.free_buffers()



# ---------------------------------------------------------------- the BENCHMARKED path pinned at the benchmarked size (round 4)
def _bench_like_batch(seed, B=64, T=64):
    """bench.synth_batch: N(0,1) pixels clipped to [-1.8, 2.2], ragged captions n ~ U{8..62}"""
    import bench

    return bench.synth_batch(B, T, 250054, 224, seed)


def _fresh_bf16_model(dev, p, dropout=0.0):
    from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration
    from mic_amd.params import unflatten_tree

    cfg = CLIPVisionMBartConfig(mbart_config=dict(dropout=dropout), clip_vision_config={})
    m = FlaxCLIPVisionMBartForConditionalGeneration(cfg, dtype=torch.bfloat16, device=dev)
    m.params = unflatten_tree({k: v.numpy() for k, v in p.items()})
    return m


def test_benchmarked_trainer_path_equals_the_padded_serial_path_at_batch64(full, dev, monkeypatch):
    """What bench.py times — `Trainer` defaults: packed decoder rows, per-bucket AdamW on a 96-CU stream beside backward, the tied
    embedding updated in two row passes (in <= 128 MB pieces), weight-gradient GEMMs on their own stream — against the plain path:
    padded [B*T] rows, dW on the main stream, ONE AdamW launch after backward.  configs[1] size (batch 64, seq 64, full model), the
    bench's own ragged batches, dropout off (the fused dropout masks are keyed by row index, which packing changes).  First-step loss
    the same number, every gradient segment within 2e-4 of its scale, 3 optimizer steps: same losses and the same weights up to what
    Adam's normalised update makes of last-bit gradient differences."""
    from mic_amd import Trainer, create_learning_rate_fn

    rc, p, *_ = full
    lr, steps = 1e-4, 3
    lr_fn = create_learning_rate_fn(10_000_000, 64, 7, 0, lr)
    batches = [_bench_like_batch(1234 + i) for i in range(steps)]
    res = {}
    for name in ("bench", "plain"):
        if name == "plain":
            monkeypatch.setenv("MIC_DW_OVERLAP", "0")
        model = _fresh_bf16_model(dev, p)
        tr = Trainer(model, lr_fn, seed=42) if name == "bench" else Trainer(model, lr_fn, seed=42, pack_rows=False, overlap_optimizer=False)
        if name == "bench":
            assert tr.pack_rows and tr.overlap_optimizer and tr._split_shared and tr.reducer.step_stream is not None and model.engine.dw_overlap
            assert len([b for b in tr.buckets if b[1] <= tr._sh[1]]) == 8  # the tied embedding in 8 pieces of <= 128 MB
        else:
            assert not tr.pack_rows and not tr.overlap_optimizer and not model.engine.dw_overlap
        losses, g1 = [], None
        for i, b in enumerate(batches):
            losses.append(float(tr.train_step(b)["loss"]))
            assert (tr._pack is not None) == (name == "bench")
            if i == 0:
                torch.cuda.synchronize()
                g1 = model.store.grad.clone()
        torch.cuda.synchronize()
        res[name] = (losses, g1, model.store.master.clone(), {k: (s_.offset, s_.numel) for k, s_ in model.store.segs.items()})
        n_valid = int(batches[0]["attention_mask"].sum())
        del tr, model
    monkeypatch.delenv("MIC_DW_OVERLAP", raising=False)
    (la, ga, pa, segs), (lb, gb, pb, _) = res["bench"], res["plain"]
    assert 2000 < n_valid < 3000
    assert la[0] == lb[0], (la[0], lb[0])                      # same weights, same per-row arithmetic: the same number
    worst = {}
    for k, (off, n) in segs.items():
        a, b = ga[off: off + n], gb[off: off + n]
        sc = b.abs().max().item()
        if sc > 0:
            worst[k] = ((a - b).abs().max() / sc).item()
    bad = {k: v for k, v in worst.items() if v > 2e-4}
    print(f"[bench path vs plain path] losses {la} vs {lb}; worst gradient segment {max(worst.values()):.2e} of its scale over {len(worst)} segments")
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    assert np.allclose(la, lb, rtol=2e-3), (la, lb)
    diff = (pa - pb).abs()
    frac = (diff > 1e-6).float().mean().item()
    print(f"[bench path vs plain path] master weights after {steps} steps: max |diff| {diff.max().item():.2e} (lr {lr}), differing by > 1e-6: {frac:.4f}")
    assert diff.max().item() <= 2.5 * lr * steps and frac < 0.2, (diff.max().item(), frac)


def test_fullsize_packed_rows_against_the_fp32_oracle_batch8(full, dev):
    """The packed decoder path (what the benchmark runs) next to the fp32 ORACLE at full size, batch 8 of the bench's ragged
    captions: loss within 2e-2, gradient direction of every large leaf and of all leaves together (the tolerances of
    tests/test_model_gpu.py::test_loss_and_grads_match_oracle at the reduced size)."""
    from mic_amd import loss_rows, packed_rows
    from oracle import train_ref

    rc, p, models, _, _ = full
    model = models[torch.bfloat16]
    b = _bench_like_batch(77, B=8)
    px, labels, mask, dec_in = (torch.from_numpy(b[k]) if b[k].dtype == np.float32 else torch.from_numpy(b[k].astype(np.int64))
                                for k in ("pixel_values", "input_ids", "attention_mask", "decoder_input_ids"))  # (the CPU oracle indexes with int64)
    ref_loss, ref_g = train_ref.loss_and_grads(rc, p, px, labels, mask, dec_in)
    d = model._dev
    B, T = labels.shape
    idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
    q_off, q_len, ids_p, pos_p = (d(t, torch.int32) for t in packed_rows(b["attention_mask"], b["decoder_input_ids"]))
    assert len(idx) < B * T
    loss = model.engine.loss_and_grads(d(px, torch.float32), ids_p, pos_p, None, d(labels, torch.int32).reshape(-1), B, T,
                                       rows=(d(idx, torch.int32), len(idx)), row_labels=d(rl, torch.int32), pack=(q_off, q_len, len(idx)))
    torch.cuda.synchronize()
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    got = model.store.export_flat("grad")
    cos = lambda a, b_: torch.nn.functional.cosine_similarity(a.reshape(-1).double(), b_.reshape(-1).double(), dim=0).item()
    leaf = {k: cos(torch.from_numpy(got[k]).reshape(rg.shape), rg) for k, rg in ref_g.items() if rg.abs().max().item() > 1e-5 and rg.numel() >= 4096}
    allc = cos(torch.cat([torch.from_numpy(got[k]).reshape(-1) for k in ref_g]), torch.cat([v.reshape(-1) for v in ref_g.values()]))
    print(f"[fullsize packed bf16 vs fp32 oracle, B=8] loss {loss.item():.4f} vs {ref_loss.item():.4f}; worst leaf cosine {min(leaf.values()):.4f} "
          f"over {len(leaf)} leaves; all leaves {allc:.5f}")
    bad = {k: v for k, v in leaf.items() if v < 0.99}
    assert not bad, sorted(bad.items(), key=lambda kv: kv[1])[:6]
    assert allc > 0.995, allc
    model.engine.free_buffers()


def test_fullsize_head_backward_nt_launches_equal_the_kmajor_launches(full):
    """round 5: the LM head's backward as NT launches of the four-wave kernel over k-contiguous copies (dlogits^T from the transposing
    CE backward, h^T, E^T) against the k-major launches of rounds 1-4 (`Engine.head_nt = False`) on the same weights and batch, at
    full size (the reduced test models have too small a vocabulary to take the NT path): same loss bits (the forward is the same), the
    logits-bias gradient (column sums of the same bf16 dlogits, different summation order), the tied embedding's gradient and — through
    dX — a decoder and a ViT weight gradient.  Also: E^T follows a weight update (ParamStore.version)."""
    rc, p, models, (px, labels, mask, dec_in), _ = full
    model = models[torch.bfloat16]
    eng, st = model.engine, model.store
    keep = eng.head_nt
    res = {}
    try:
        for nt in (True, False):
            eng.head_nt = nt
            res[nt] = _grads(model, px, labels, mask, dec_in, compact=True)
        assert eng._bufs.get(("w.sharedT", st.d, st.Vpad, torch.bfloat16)) is not None  # the NT pass built E^T
        assert res[True][0] == res[False][0]
        # rows of the embedding no decoder input id touches hold dE alone: the two launches agree to fp32 summation order there
        sh = st.segs["shared"]
        untouched = torch.ones(st.Vpad, dtype=torch.bool)
        untouched[dec_in.reshape(-1)] = False
        ga, gb = (res[k][1][sh.offset: sh.offset + sh.numel].view(st.Vpad, st.d)[untouched.to(res[k][1].device)] for k in (True, False))
        assert ((ga - gb).abs().max() / gb.abs().max()).item() < 2e-5
        # (the tied embedding's gradient = dE, fp32 summation order only, + the input-embedding rows scattered in at the END of the
        # bf16 backward chain: those rows carry dX's rounding differences like every weight gradient behind the head)
        for name, tol in (("flb", 2e-5), ("shared", 3e-2), ("dec11.fc2.w", 1e-2), ("dec0.qkv.w", 1e-2), ("vit0.fc1.w", 2e-2)):
            s = st.segs[name]
            a, b = res[True][1][s.offset: s.offset + s.numel], res[False][1][s.offset: s.offset + s.numel]
            e = ((a - b).abs().max() / b.abs().max()).item()
            assert e < tol, (name, e)  # fp32 summation order up to dX; behind it bf16 roundings of dh that fall the other way
        # E^T is rebuilt when the weights move: perturb the embedding's compute copy, bump the version, compare with a fresh transpose
        eng.head_nt = True
        w = st.w("shared")
        w[:16].mul_(2.0)
        model.invalidate_params_cache()
        ET = eng.shared_T()
        torch.cuda.synchronize()
        assert torch.equal(ET[:, : st.Vpad], w.T)
    finally:
        eng.head_nt = keep
        from mic_amd.params import unflatten_tree

        model.params = unflatten_tree({k: v.numpy() for k, v in p.items()})  # (the fixture is shared: put the weights back)
