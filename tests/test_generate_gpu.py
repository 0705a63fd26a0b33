"""Generation parity on the GPU: KV-cached decode steps, greedy and beam-4 `.generate()` of the HIP path vs the oracle's
restatement of the reference algorithm (`generation_clip_vision_utils.py`) driving the CPU model oracle.
float32 mode: token ids bit-exact, beam scores within 1e-4.  bfloat16 mode: ids are compared against the oracle
as an agreement rate (bf16 logits tie massively; exactness is defined in float32 — SURVEY §7 hard parts)."""
import numpy as np
import pytest
import torch

from util_small import batch, make_pair

pytestmark = pytest.mark.gpu


def _oracle_gen(rc, p, px, B, **kw):
    from oracle import generation_ref as G
    from oracle import model_ref as M

    with torch.no_grad():
        ehs, _ = M.encode(rc, p, px)
    K = kw.get("num_beams", 1)
    L = kw["max_length"]
    return G.generate(lambda rows: G.ModelStepper(rc, p, ehs.repeat_interleave(rows // B, 0), L), B, G.GenDefaults(), **kw)


def test_cached_decode_steps_match_oracle(dev):
    """decode() with past_key_values (modeling:519-651): per-step logits vs the oracle's static-cache decode."""
    from oracle import model_ref as M

    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6)
    B, L = 3, 8
    px, *_ = batch(rc, B, 12, seed=11)
    with torch.no_grad():
        ehs, _ = M.encode(rc, p, px)
    enc = model.encode(px.numpy())
    state = M.DecodeState(rc, B, L)
    cache = model.init_cache(B, L, enc)
    g = torch.Generator().manual_seed(0)
    with pytest.raises(ValueError):
        model.decode(np.zeros((B, 1), dtype=np.int32), enc, past_key_values=cache)  # positions are mandatory with a cache
    for t in range(5):
        tok = torch.randint(4, rc.vocab_size, (B, 1), generator=g)
        pos = torch.full((B, 1), t)
        with torch.no_grad():
            ref = M.decode_step(rc, p, state, tok, pos, ehs)
        out = model.decode(tok.numpy(), enc, decoder_position_ids=pos.numpy(), past_key_values=cache)
        cache = out.past_key_values
        err = (out.logits.cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-4, (t, err)
    # without a cache: teacher-forced decode == __call__'s decoder
    ids = torch.randint(4, rc.vocab_size, (B, 6), generator=g)
    out = model.decode(ids.numpy(), enc)
    with torch.no_grad():
        h = M.decoder_forward(rc, p, ids, torch.ones_like(ids), torch.arange(6)[None].expand(B, 6), ehs)
        ref = M.lm_head(rc, p, h)
    assert ((out.logits.cpu() - ref).abs().max() / ref.abs().max()).item() < 2e-4


@pytest.mark.parametrize("kw", [dict(max_length=10), dict(max_length=12, forced_bos_token_id=996),
                                dict(max_length=9, decoder_start_token_id=997, forced_eos_token_id=None), dict(max_length=10, min_length=4)])
def test_greedy_ids_exact(dev, kw):
    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6)
    B = 3
    px, *_ = batch(rc, B, 12, seed=21)
    ref = _oracle_gen(rc, p, px, B, num_beams=1, **kw)
    out = model.generate(px.numpy(), num_beams=1, **kw)
    assert np.array_equal(out.sequences.cpu().numpy(), ref)


@pytest.mark.parametrize("kw", [dict(max_length=10, forced_bos_token_id=996), dict(max_length=12), dict(max_length=8, decoder_start_token_id=998),
                                dict(max_length=10, num_beams=2, length_penalty=1.0, early_stopping=False),
                                dict(max_length=9, num_beams=5), dict(max_length=8, num_beams=8, forced_bos_token_id=995),
                                dict(max_length=9, num_beams=7, length_penalty=0.6),
                                # wider than 8 beams: 2K > 16 candidates per row stream the logits (row_lse_topk's 32-wide build), the
                                # bookkeeping kernel runs 512 threads per image
                                dict(max_length=8, num_beams=12), dict(max_length=7, num_beams=16, forced_bos_token_id=994, length_penalty=0.8),
                                # wider than 16: the 64-wide build of the streaming top-k, four (beam, candidate) pairs per bookkeeping thread
                                dict(max_length=7, num_beams=20), dict(max_length=6, num_beams=32, length_penalty=1.2)])
def test_beam_ids_exact(dev, kw):
    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6)
    B = 3
    px, *_ = batch(rc, B, 12, seed=22)
    kw = dict(kw)
    kw.setdefault("num_beams", 4)
    ref = _oracle_gen(rc, p, px, B, **kw)
    out = model.generate(px.numpy(), **kw)
    assert out["steps"] == ref.steps
    assert np.array_equal(out.sequences.cpu().numpy(), ref.sequences), (out.sequences, ref.sequences)
    assert np.allclose(out.scores.cpu().numpy(), ref.scores, rtol=1e-4, atol=1e-4)


def test_beam_early_finish_eos_bias(dev):
    """Push EOS up through final_logits_bias so beams finish early: exercises the finished-merge, the -1e7 arithmetic,
    early_stopping termination and the improvement test (gen:798-820, 889-940)."""
    from mic_amd.params import flatten_tree, unflatten_tree

    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6)
    p = dict(p)
    flb = p["final_logits_bias"].clone()
    flb[0, rc.eos_token_id] = 6.0
    p["final_logits_bias"] = flb
    model.params = unflatten_tree({k: v.numpy() for k, v in p.items()})
    B = 4
    px, *_ = batch(rc, B, 12, seed=23)
    for kw in (dict(max_length=12, num_beams=4, decoder_start_token_id=999), dict(max_length=12, num_beams=3, early_stopping=False, decoder_start_token_id=999),
               dict(max_length=12, num_beams=5, decoder_start_token_id=999)):  # 5 = the mbart-large-50 config default ("model has beam size 5", main.py:723)
        ref = _oracle_gen(rc, p, px, B, **kw)
        out = model.generate(px.numpy(), **kw)
        assert out["steps"] == ref.steps, (out["steps"], ref.steps)
        assert np.array_equal(out.sequences.cpu().numpy(), ref.sequences)
        assert np.allclose(out.scores.cpu().numpy(), ref.scores, rtol=1e-4, atol=1e-4)
    assert (ref.sequences == rc.eos_token_id).any()  # finished hypotheses keep EOS (unlike greedy)


def test_decode_plan_graph_replay_matches_oracle(dev, monkeypatch):
    """Launch-free decoder steps (MIC_DECODE_GRAPHS=1): call 1 of a (batch, beams, max_length, processors) configuration runs eagerly, call 2 captures
    every step t >= 2 into a hipGraph and replays it, call 3 only replays.  Every call — different images, different forced-BOS
    ids (step 1 is never captured), with and without early finishes — must give the oracle's ids, scores and step count, and
    the same as the eager path (MIC_DECODE_GRAPHS=0)."""
    from mic_amd.params import unflatten_tree

    monkeypatch.setenv("MIC_DECODE_GRAPHS", "1")   # opt-in (measured: no gain on this stack, see generation_clip_vision_utils.py)
    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6)
    B = 3
    for use_eos_bias in (False, True):
        if use_eos_bias:  # beams finish early: the device-side stop flag and the "no-op after stop" launches inside replayed graphs
            p = dict(p)
            flb = p["final_logits_bias"].clone()
            flb[0, rc.eos_token_id] = 6.0
            p["final_logits_bias"] = flb
            model.params = unflatten_tree({k: v.numpy() for k, v in p.items()})
        for kw in (dict(max_length=12, num_beams=4), dict(max_length=20, num_beams=1)):
            model.release_decode_plans()
            for call, (seed, bos) in enumerate([(31, 996), (32, 995), (33, 994), (34, 996)]):
                px, *_ = batch(rc, B, 12, seed=seed)
                k2 = dict(kw, forced_bos_token_id=bos)
                ref = _oracle_gen(rc, p, px, B, **k2)
                out = model.generate(px.numpy(), **k2)
                plan = next(iter(model._decode_plans.values()))
                assert plan.calls == call + 1 and len(model._decode_plans) == 1
                if call >= 1:
                    assert len(plan.graphs) > 0 and 1 not in plan.graphs   # steps >= 2 captured, step 1 never
                ref_seq = ref.sequences if kw["num_beams"] > 1 else ref
                assert np.array_equal(out.sequences.cpu().numpy(), ref_seq), (use_eos_bias, kw, call)
                if kw["num_beams"] > 1:
                    assert out["steps"] == ref.steps
                    assert np.allclose(out.scores.cpu().numpy(), ref.scores, rtol=1e-4, atol=1e-4)
            # the same last call without graphs: identical
            monkeypatch.setenv("MIC_DECODE_GRAPHS", "0")
            eager = model.generate(px.numpy(), **k2)
            monkeypatch.setenv("MIC_DECODE_GRAPHS", "1")
            assert torch.equal(eager.sequences, out.sequences)
    # at most _MAX_PLANS plans are kept
    px, *_ = batch(rc, B, 12, seed=40)
    for L in (6, 7, 8):
        model.generate(px.numpy(), max_length=L, num_beams=2)
    assert len(model._decode_plans) == 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ns", [2, 4])
def test_sliced_beam_search_equals_the_single_chain(dev, monkeypatch, dtype, ns):
    """MIC_DECODE_SLICES=n cuts the images of a beam search into n slices whose decoder-step chains run as parallel branches of
    the step's hipGraph (own streams, state, KV cache and scratch buffers).  Images never interact (gen:665-990 is per batch
    item), so ids, scores and the step count must be those of the single chain and of the oracle — on the eager first call, the
    capturing second call and the replaying third; with beams of ONE slice finishing early while the other keeps going; and
    after Engine.free_buffers() dropped everything the captured graphs pointed at."""
    from mic_amd.params import unflatten_tree

    rc, p, model = make_pair(dtype, dev, gelu="tanh", decoder_ln_eps=1e-6)
    B, K = 8, 4
    p = dict(p)
    flb = p["final_logits_bias"].clone()
    flb[0, rc.eos_token_id] = 5.0   # some beams finish early
    p["final_logits_bias"] = flb
    model.params = unflatten_tree({k: v.numpy() for k, v in p.items()})
    kw = dict(max_length=14, num_beams=K)
    for call, (seed, bos) in enumerate([(51, 996), (52, 995), (53, 994)]):
        px, *_ = batch(rc, B, 12, seed=seed)
        k2 = dict(kw, forced_bos_token_id=bos)
        monkeypatch.setenv("MIC_DECODE_SLICES", "1")
        one = model.generate(px.numpy(), **k2)
        monkeypatch.setenv("MIC_DECODE_SLICES", str(ns))
        cut = model.generate(px.numpy(), **k2)
        plan = [v for k, v in model._decode_plans.items() if k[3] == ns][0]
        assert len(plan.subs) == ns and plan.calls == call + 1
        if call >= 1:
            assert len(plan.graphs) > 0  # sliced chains are graph branches by default
        assert torch.equal(one.sequences, cut.sequences), (call, ns)
        assert torch.equal(one.scores, cut.scores) and one["steps"] == cut["steps"]
        if dtype == torch.float32:
            ref = _oracle_gen(rc, p, px, B, **k2)
            assert np.array_equal(cut.sequences.cpu().numpy(), ref.sequences) and cut["steps"] == ref.steps
            assert np.allclose(cut.scores.cpu().numpy(), ref.scores, rtol=1e-4, atol=1e-4)
    # only the first half of the images finishes early: the stop flags of the slices differ, the loop ends when ALL have stopped
    px2 = px.clone()
    px2[B // 2:] = px[: B // 2].flip(0) * 0.3
    monkeypatch.setenv("MIC_DECODE_SLICES", "1")
    one = model.generate(px2.numpy(), **k2)
    monkeypatch.setenv("MIC_DECODE_SLICES", str(ns))
    cut = model.generate(px2.numpy(), **k2)
    assert torch.equal(one.sequences, cut.sequences) and torch.equal(one.scores, cut.scores) and one["steps"] == cut["steps"]
    # free_buffers() drops the plans (their graphs point into the engine's buffers); the next call rebuilds everything
    model.engine.free_buffers()
    assert not getattr(model, "_decode_plans", None)
    again = model.generate(px2.numpy(), **k2)
    assert torch.equal(again.sequences, cut.sequences) and torch.equal(again.scores, cut.scores)


def test_decode_layernorm_fold_matches_explicit_layernorms(dev):
    """bfloat16 decoder steps with the LayerNorms folded around the GEMMs (default) against the explicit LayerNorm kernels
    (engine.decode_ln_fold = False): same cache protocol, logits equal up to the one bf16 rounding the fold skips (the
    normalised activations are never rounded) — 2e-2 of the logit scale over 6 cached steps, and the same greedy tokens
    wherever the explicit path's top-2 margin exceeds that error.  float32 mode never folds (bit-exact tests above)."""
    rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, d_layers=3)
    eng = model.engine
    assert eng.decode_ln_fold
    B, L = 5, 8
    px, *_ = batch(rc, B, 12, seed=61)
    enc = model.encode(px.numpy())
    g = torch.Generator().manual_seed(5)
    toks = torch.randint(4, rc.vocab_size - 20, (L, B, 1), generator=g).numpy().astype(np.int32)
    outs = {}
    for fold in (True, False):
        eng.decode_ln_fold = fold
        cache = model.init_cache(B, L, enc)
        logits = []
        for t in range(6):
            pos = np.full((B, 1), t, dtype=np.int32)
            o = model.decode(toks[t], enc, decoder_position_ids=pos, past_key_values=cache)
            cache = o.past_key_values
            logits.append(torch.as_tensor(o.logits).float().cpu().reshape(B, -1))
        outs[fold] = torch.stack(logits)
    eng.decode_ln_fold = True
    a, b = outs[True], outs[False]
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err < 2e-2 * scale, (err, scale)
    top2 = b.topk(2, dim=-1).values
    decisive = (top2[..., 0] - top2[..., 1]) > 2 * err
    assert decisive.float().mean().item() > 0.5   # the comparison below is not vacuous
    assert torch.equal(a.argmax(-1)[decisive], b.argmax(-1)[decisive])
    # the fold's operands follow the weights: after a parameter update the folded copies are rebuilt
    v0 = model.store.version
    model.invalidate_params_cache()
    assert model.store.version == v0 + 1


def _rule_batch(rc, classes, T=12, n=8):
    """synthetic captioning task with a learnable, deterministic answer: the image is a constant colour that encodes a class c,
    the caption is lang, s_1 = 100 + c, s_{t+1} = 100 + (3 (s_t - 100) + c + 1) mod 200, ..., eos"""
    B = len(classes)
    px = torch.zeros(B, rc.image_size, rc.image_size, 3)
    labels = torch.full((B, T), rc.pad_token_id, dtype=torch.int64)
    mask = torch.zeros((B, T), dtype=torch.int64)
    for b, c in enumerate(classes):
        px[b, :, :, 0], px[b, :, :, 1], px[b, :, :, 2] = (c & 1) * 2.0 - 1.0, ((c >> 1) & 1) * 2.0 - 1.0, ((c >> 2) & 1) * 2.0 - 1.0
        s = 100 + c
        labels[b, 0] = rc.vocab_size - 10
        for t in range(n):
            labels[b, 1 + t] = s
            s = 100 + (3 * (s - 100) + c + 1) % 200
        labels[b, 1 + n] = rc.eos_token_id
        mask[b, : n + 2] = 1
    dec_in = torch.full_like(labels, rc.pad_token_id)
    dec_in[:, 1:] = labels[:, :-1]
    return px, labels, mask, dec_in


def test_generate_bf16_agrees_with_fp32_oracle_on_a_trained_model(dev):
    """bfloat16 is what bench.py times; a randomly initialised model cannot tell whether its decisions survive the precision
    change (its top-1 is the copy of the fed token by a wide margin, everything else is a 250k-way near-tie).  So: train the
    reduced model with the float32 HIP path on a deterministic synthetic task until it has real, input-dependent decisions,
    then decode the TRAINED weights three ways — fp32 oracle on the CPU (the reference algorithm), fp32 HIP, bf16 HIP.
    Asserted: the training loss is small; fp32 HIP ids == oracle ids; bf16 HIP: at least 6 of 8 captions identical to the oracle's
    (greedy and beam-4), beam scores of identical hypotheses within 2e-2, and teacher-forced top-1 identical wherever the fp32
    margin exceeds 4x the measured bf16 logit error."""
    from mic_amd import Trainer, create_learning_rate_fn
    from mic_amd.params import flatten_tree, unflatten_tree

    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0)
    tr = Trainer(model, create_learning_rate_fn(10_000, 1, 1, 20, 3e-3), seed=3)
    g = torch.Generator().manual_seed(0)
    for step in range(500):
        cls = torch.randint(0, 8, (32,), generator=g).tolist()
        px, labels, mask, dec_in = _rule_batch(rc, cls)
        out = tr.train_step({"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()})
    final = float(out["loss"])
    assert final < 0.05, final  # the task is learned: predictions now have margins
    trained = model.params
    p2 = {k: torch.from_numpy(np.asarray(v)) for k, v in flatten_tree(trained).items()}
    _, _, m16 = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0)
    m16.params = unflatten_tree({k: v.numpy() for k, v in p2.items()})
    cls = list(range(8))
    px, labels, *_ = _rule_batch(rc, cls)
    lang = rc.vocab_size - 10
    from oracle import model_ref as M

    for kw in (dict(max_length=12, num_beams=1, decoder_start_token_id=lang, forced_eos_token_id=None),
               dict(max_length=12, num_beams=4, decoder_start_token_id=lang, forced_eos_token_id=None)):
        K = kw["num_beams"]
        ref = _oracle_gen(rc, p2, px, 8, **kw)
        ref_seq = ref if K == 1 else ref.sequences
        got32 = model.generate(px.numpy(), **kw)
        assert np.array_equal(got32.sequences.cpu().numpy(), ref_seq)
        # how much of the rule free-running decoding reproduces is informational (the point is that decisions now have margins)
        print(f"   oracle captions follow the training rule on {(ref_seq[:, 1:9] == labels[:, 1:9].numpy()).mean():.2f} of the positions")
        got16 = m16.generate(px.numpy(), **kw)
        seq16 = got16.sequences.cpu().numpy()
        agree = float((seq16 == ref_seq).mean())
        whole = (seq16 == ref_seq).all(axis=1)
        print(f"trained reduced model, num_beams={K}: bf16 token agreement with the fp32 oracle {agree:.3f}, {int(whole.sum())}/8 captions identical")
        # free-running: one flipped near-tie changes the rest of that caption, so the floor is per caption (training itself is
        # not bit-reproducible — fp32 atomics — so which steps are near-ties varies from run to run)
        assert whole.sum() >= 6 and agree >= 0.75, (K, agree, seq16, ref_seq)
        if K > 1 and whole.any():
            d = np.abs(got16.scores.cpu().numpy() - ref.scores)[whole]
            print(f"   beam scores of identical hypotheses: max |difference| {d.max():.4f} (scores ~ {np.abs(ref.scores).mean():.3f})")
            # a hypothesis whose score carries the reference's -1e7 offset (gen:890, 910: unfinished candidates in the finished list)
            # has an fp32 ulp of 0.125 at |score| ~ 1.4e6: 2e-2 absolute on ordinary scores, 3 ulp on those
            lim = 2e-2 + 3 * np.spacing(np.abs(ref.scores[whole]).astype(np.float32))
            assert (d <= lim).all(), (d, ref.scores[whole])
        if K == 1:
            # teacher-forced on the oracle's own captions: wherever the fp32 decision has a margin, bf16 must make the same one
            ids = torch.from_numpy(ref_seq.astype(np.int64))
            with torch.no_grad():
                lo = M.forward_logits(rc, p2, px, ids, torch.ones_like(ids))
            l16 = m16(px.numpy(), ref_seq, np.ones_like(ref_seq))[0].float().cpu()
            top2 = lo.topk(2, dim=-1)
            margin = top2.values[..., 0] - top2.values[..., 1]
            err = (l16 - lo).abs().max().item()
            clear = margin > 4 * err
            same = l16.argmax(-1) == top2.indices[..., 0]
            print(f"   teacher-forced: max |bf16 - fp32| logit {err:.3f}; {int(clear.sum())}/{clear.numel()} positions have a margin > 4x that; "
                  f"top-1 agreement overall {same.float().mean():.3f}")
            assert bool(same[clear].all()) and clear.float().mean() > 0.5 and same.float().mean() > 0.9


def test_generate_api_errors(dev):
    rc, p, model = make_pair(torch.float32, dev)
    px, *_ = batch(rc, 1, 12, seed=1)
    with pytest.raises(NotImplementedError):
        model.generate(px.numpy(), do_sample=True, num_beams=4, max_length=5)
    with pytest.raises(NotImplementedError, match="num_beams > 32"):
        model.generate(px.numpy(), num_beams=33, max_length=5)
    model.config.mbart_config.decoder_start_token_id = None
    with pytest.raises(ValueError):
        model.generate(px.numpy(), max_length=5, num_beams=1)


# ---------------------------------------------------------------- sampling (gen:537-663)
def test_sample_rows_matches_jax_threefry_oracle(dev):
    """mic_sample_rows = argmax(logits + Gumbel(threefry2x32 stream)) against the oracle's restatement of
    jax.random.categorical: odd and even R*V (counter padding), bf16 and fp32 logits, temperature, EOS suppression."""
    from mic_amd import ops
    from oracle import generation_ref as G

    for (R, V, dt) in ((3, 1003, torch.float32), (4, 1000, torch.float32), (2, 5003, torch.bfloat16), (1, 7, torch.float32)):
        g = torch.Generator().manual_seed(R * V)
        x = (torch.randn(R, V, generator=g) * 3).to(dt)
        xd = x.to(dev)
        out = torch.empty(R, dtype=torch.int32, device=dev)
        key = G.prng_split(G.prng_key(R + V))[0]
        ops.sample_rows(xd, xd.stride(0), V, key, out, R)
        noisy = G.gumbel(key, (R, V)) + x.float().numpy()
        ref = noisy.argmax(-1)
        got = out.cpu().numpy()
        for r in range(R):  # identical draw, or an fp32 log-rounding near-tie
            assert got[r] == ref[r] or noisy[r, got[r]] >= noisy[r, ref[r]] - 1e-5, (R, V, r)
        assert (got == ref).all()
        ops.sample_rows(xd, xd.stride(0), V, key, out, R, temperature=0.7, suppress_eos=True, eos_token_id=2)
        y = (x.float().numpy() / np.float32(0.7)).astype(np.float32)
        y[:, 2] = -np.inf
        assert (out.cpu().numpy() == (G.gumbel(key, (R, V)) + y).argmax(-1)).all()
        ops.sample_rows(xd, xd.stride(0), V, key, out, R, forced_token=5)
        assert (out.cpu().numpy() == 5).all()


def test_generate_do_sample_matches_oracle(dev):
    """generate(do_sample=True): same PRNG key -> same token ids as the oracle's `sample` loop, in the reference's
    (raw-logits) mode and in the processed-logits mode; different keys differ; beam-sample is refused (gen:336)."""
    from oracle import generation_ref as G
    from oracle import model_ref as M

    rc, p, model = make_pair(torch.float32, dev)
    B, L = 3, 9
    px = batch(rc, B, 8, seed=4)[0]
    with torch.no_grad():
        ehs, _ = M.encode(rc, p, px, int32_cast=True)

    def stepper():
        return G.ModelStepper(rc, p, ehs, L)

    procs = G.get_logits_processor(0, L, rc.eos_token_id, rc.vocab_size - 7, rc.eos_token_id)
    for mode in (False, True):
        ref = G.sample(stepper(), B, 2, L, rc.pad_token_id, rc.eos_token_id, G.prng_key(123), procs,
                       G.get_logits_warper(0, 1.0, 0.8) if mode else [], sample_from_processed_logits=mode)
        out = model.generate(px, max_length=L, do_sample=True, num_beams=1, prng_key=123, top_k=0, top_p=1.0, temperature=0.8,
                             forced_bos_token_id=rc.vocab_size - 7, sample_from_processed_logits=mode)
        assert np.array_equal(out.sequences.cpu().numpy(), ref), mode
    o2 = model.generate(px, max_length=L, do_sample=True, num_beams=1, prng_key=124, top_k=0, top_p=1.0)
    assert not np.array_equal(o2.sequences.cpu().numpy(), ref)
    o3 = model.generate(px, max_length=L, do_sample=True, num_beams=1, prng_key=np.array([0, 124], dtype=np.uint32), top_k=0, top_p=1.0)
    assert torch.equal(o2.sequences, o3.sequences)  # PRNGKey(124) == [0, 124]
    with pytest.raises(NotImplementedError):
        model.generate(px, max_length=L, do_sample=True, num_beams=2)


def test_warp_thresholds_match_oracle_warpers(dev):
    """mic_warp_thresholds (threshold value + tie index limit) reproduces the keep-mask of FlaxTopK/TopP warpers as the oracle
    restates them (sort-based), including ties at the cut (bf16-valued logits tie heavily) and temperature / MinLength first."""
    from mic_amd import ops
    from oracle import generation_ref as G

    cases = [(3, 1003, 50, 1.0, 1.0, torch.float32), (3, 1003, 0, 0.9, 1.0, torch.float32), (2, 5003, 7, 0.5, 0.7, torch.bfloat16),
             (4, 2000, 1, 1.0, 1.0, torch.bfloat16), (2, 1003, 2000, 0.3, 1.3, torch.float32), (2, 4096, 100, 0.95, 2.0, torch.bfloat16),
             (1, 300, 5, 1e-4, 1.0, torch.float32)]
    for (R, V, k, p, temp, dt) in cases:
        g = torch.Generator().manual_seed(R * V + k)
        x = (torch.randn(R, V, generator=g) * 2).to(dt)
        xd = x.to(dev)
        thr = torch.empty(R, dtype=torch.float32, device=dev)
        lim = torch.empty(R, dtype=torch.int32, device=dev)
        ops.warp_thresholds(xd, xd.stride(0), V, thr, lim, R, temperature=temp, suppress_eos=True, eos_token_id=2, top_k=k, top_p=p)
        y = x.float().numpy()
        y[:, 2] = -np.inf                                     # MinLength processor
        ref = y
        for w in G.get_logits_warper(k, p, temp):
            ref = w(None, ref, 1)
        yt = (y / np.float32(temp)).astype(np.float32) if temp != 1.0 else y
        t, l = thr.cpu().numpy(), lim.cpu().numpy()
        idx = np.arange(V)[None, :]
        keep = (yt > t[:, None]) | ((yt == t[:, None]) & (idx < l[:, None]))
        keep &= np.isfinite(yt)
        ref_keep = np.isfinite(ref)
        diff = (keep != ref_keep).sum(-1)
        if p < 1.0:  # the nucleus boundary is decided by an fp32 cumulative sum in the oracle, fixed-point masses here
            assert (diff <= 1).all(), (R, V, k, p, diff)
        else:
            assert (diff == 0).all(), (R, V, k, p, diff)
        assert (keep.sum(-1) >= 1).all()


def test_generate_sample_with_topk_topp(dev):
    from oracle import generation_ref as G
    from oracle import model_ref as M

    rc, p, model = make_pair(torch.float32, dev)
    B, L = 3, 8
    px = batch(rc, B, 8, seed=9)[0]
    with torch.no_grad():
        ehs, _ = M.encode(rc, p, px, int32_cast=True)
    procs = G.get_logits_processor(3, L, rc.eos_token_id, None, rc.eos_token_id)
    ref = G.sample(G.ModelStepper(rc, p, ehs, L), B, 2, L, rc.pad_token_id, rc.eos_token_id, G.prng_key(77), procs,
                   G.get_logits_warper(20, 0.9, 0.7), sample_from_processed_logits=True)
    out = model.generate(px, max_length=L, do_sample=True, num_beams=1, prng_key=77, top_k=20, top_p=0.9, temperature=0.7,
                         min_length=3, sample_from_processed_logits=True)
    assert np.array_equal(out.sequences.cpu().numpy(), ref)
    # top_k = 1 is greedy on the processed logits whatever the key
    o1 = model.generate(px, max_length=L, do_sample=True, num_beams=1, prng_key=5, top_k=1, top_p=1.0, sample_from_processed_logits=True)
    og = model.generate(px, max_length=L, num_beams=1)
    assert torch.equal(o1.sequences, og.sequences)
