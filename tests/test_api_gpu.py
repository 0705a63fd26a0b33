"""Drop-in API surface of SURVEY §8(b) on the GPU: checkpoint round trips (config.json + flax_model.msgpack),
`from_clip_vision_mbart_pretrained` grafting (modeling:703-773), the params property/setter contract (utils:99-117) and the
`params=` keyword the reference's pmapped functions pass (main.py:692, 727; evaluation.py:81)."""
import json
import os

import numpy as np
import pytest
import torch

from util_small import batch, make_pair

pytestmark = pytest.mark.gpu


def test_save_and_from_pretrained_roundtrip(dev, tmp_path):
    from mic_amd import FlaxCLIPVisionMBartForConditionalGeneration as Model

    rc, p, model = make_pair(torch.float32, dev)
    px, labels, mask, dec_in = batch(rc, 2, 12, seed=31)
    ref = model(px.numpy(), dec_in.numpy(), mask.numpy())[0].cpu()
    d = str(tmp_path / "ckpt")
    model.save_pretrained(d)
    assert sorted(os.listdir(d)) == ["config.json", "flax_model.msgpack"]  # main.py:307-312 layout
    cfg = json.load(open(os.path.join(d, "config.json")))
    assert cfg["model_type"] == "clip-vision-mbart" and cfg["mbart_config"]["d_model"] == rc.d_model
    m2 = Model.from_pretrained(d, device=dev)
    assert m2.config.mbart_config.decoder_ln_eps == model.config.mbart_config.decoder_ln_eps
    got = m2(px.numpy(), dec_in.numpy(), mask.numpy())[0].cpu()
    assert torch.equal(got, ref)
    with pytest.raises(EnvironmentError):
        Model.from_pretrained(str(tmp_path / "nope"))
    with pytest.raises(NotImplementedError):
        model.save_pretrained(d, push_to_hub=True)


def test_from_clip_vision_mbart_pretrained_grafts_components(dev, tmp_path):
    from mic_amd import FlaxCLIPVisionMBartForConditionalGeneration as Model
    from mic_amd.checkpoint import save_flax_msgpack
    from mic_amd.params import flatten_tree, unflatten_tree

    rc, p, model = make_pair(torch.float32, dev)
    tree = unflatten_tree({k: v.numpy() for k, v in p.items()})
    clip_dir, mbart_dir = str(tmp_path / "clip"), str(tmp_path / "mbart")
    os.makedirs(clip_dir), os.makedirs(mbart_dir)
    save_flax_msgpack(os.path.join(clip_dir, "flax_model.msgpack"), tree["model"]["encoder"])  # FlaxCLIPVisionModel.params
    json.dump(model.config.clip_vision_config.to_dict(), open(os.path.join(clip_dir, "config.json"), "w"))
    save_flax_msgpack(os.path.join(mbart_dir, "flax_model.msgpack"),
                      {"shared": tree["model"]["shared"], "decoder": tree["model"]["decoder"], "encoder": {"ignored": np.zeros(3, np.float32)}})
    json.dump(model.config.mbart_config.to_dict(), open(os.path.join(mbart_dir, "config.json"), "w"))
    m = Model.from_clip_vision_mbart_pretrained(clip_dir, mbart_dir, seed=3, dtype="float32", mbart_from_pt=True, device=dev)  # main.py:421-427
    got = flatten_tree(m.params)
    for k, v in p.items():
        if k.startswith("model/encoder/") or k.startswith("model/decoder/") or k.startswith("model/shared/"):
            assert np.array_equal(got[k], v.numpy()), k  # modeling:768-770
    # visual_projection / final_logits_bias are NOT loaded: random / zero init (modeling:766-770)
    assert not np.array_equal(got["model/visual_projection/kernel"], p["model/visual_projection/kernel"].numpy())
    assert np.count_nonzero(got["final_logits_bias"]) == 0 and got["final_logits_bias"].shape == (1, rc.vocab_size)
    with pytest.raises(AssertionError):
        Model.from_clip_vision_mbart_pretrained(None, mbart_dir, device=dev)


def test_from_pretrained_components_head_model_layout_and_missing_leaves(dev, tmp_path):
    """A FlaxMBartForConditionalGeneration checkpoint keeps its weights under `model/` next to `final_logits_bias`; HF
    `from_pretrained` strips that base-model prefix (modeling:740-758 loads through FlaxMBartModel.from_pretrained).  The
    graft must find them there, and must refuse a component that does not supply every decoder/embedding leaf instead of
    silently keeping random-init weights."""
    from mic_amd import FlaxCLIPVisionMBartForConditionalGeneration as Model
    from mic_amd.checkpoint import save_flax_msgpack
    from mic_amd.params import flatten_tree, unflatten_tree

    rc, p, model = make_pair(torch.float32, dev)
    tree = unflatten_tree({k: v.numpy() for k, v in p.items()})
    clip_dir, mbart_dir, bad_dir = str(tmp_path / "clip"), str(tmp_path / "mbart"), str(tmp_path / "bad")
    os.makedirs(clip_dir), os.makedirs(mbart_dir), os.makedirs(bad_dir)
    save_flax_msgpack(os.path.join(clip_dir, "flax_model.msgpack"), tree["model"]["encoder"])
    json.dump(model.config.clip_vision_config.to_dict(), open(os.path.join(clip_dir, "config.json"), "w"))
    head_layout = {"model": {"shared": tree["model"]["shared"], "decoder": tree["model"]["decoder"],
                             "encoder": {"ignored": np.zeros(3, np.float32)}},
                   "final_logits_bias": np.ones((1, rc.vocab_size), np.float32)}
    save_flax_msgpack(os.path.join(mbart_dir, "flax_model.msgpack"), head_layout)
    json.dump(model.config.mbart_config.to_dict(), open(os.path.join(mbart_dir, "config.json"), "w"))
    m = Model.from_clip_vision_mbart_pretrained(clip_dir, mbart_dir, seed=3, dtype="float32", device=dev)
    got = flatten_tree(m.params)
    for k, v in p.items():
        if k.startswith(("model/encoder/", "model/decoder/", "model/shared/")):
            assert np.array_equal(got[k], v.numpy()), k
    assert np.count_nonzero(got["final_logits_bias"]) == 0  # the component's own bias is not grafted (modeling:766-770)
    # a component without one decoder layer's leaves
    dec = {k: v for k, v in tree["model"]["decoder"].items()}
    dec["layers"] = {k: v for k, v in dec["layers"].items() if k != "1"}
    save_flax_msgpack(os.path.join(bad_dir, "flax_model.msgpack"), {"shared": tree["model"]["shared"], "decoder": dec})
    json.dump(model.config.mbart_config.to_dict(), open(os.path.join(bad_dir, "config.json"), "w"))
    with pytest.raises(ValueError, match="mBART checkpoint supplies"):
        Model.from_clip_vision_mbart_pretrained(clip_dir, bad_dir, device=dev)
    with pytest.raises(ValueError, match="CLIP vision checkpoint supplies"):
        Model.from_clip_vision_mbart_pretrained(mbart_dir, mbart_dir, device=dev)


def test_step_metrics_do_not_alias(dev):
    """main.py:776 appends every step's metric dict and averages later: entries must keep their own values."""
    from mic_amd import Trainer, create_learning_rate_fn

    rc, p, model = make_pair(torch.float32, dev)
    tr = Trainer(model, create_learning_rate_fn(640, 2, 5, 3, 1e-2), seed=1)
    mk = lambda s: dict(zip(("pixel_values", "input_ids", "attention_mask", "decoder_input_ids"), (x.numpy() for x in batch(rc, 2, 10, seed=s))))
    kept = [tr.train_step(mk(40 + i)) for i in range(3)]
    ev = tr.eval_step(mk(50))
    torch.cuda.synchronize()
    lrs = [float(k["learning_rate"]) for k in kept]
    assert lrs[0] == 0.0 and lrs[1] > 0 and lrs[2] > lrs[1], lrs  # warm-up 0 -> lr: distinct values survive later steps
    losses = [float(k["loss"]) for k in kept]
    assert len({round(l, 6) for l in losses}) == 3 and float(ev["loss"]) not in losses


def test_params_property_contract_and_params_kwarg(dev):
    from mic_amd.params import flatten_tree, unflatten_tree

    rc, p, model = make_pair(torch.float32, dev)
    tree = model.params
    assert set(tree) == {"model", "final_logits_bias"} and set(tree["model"]) == {"encoder", "decoder", "shared", "visual_projection"}
    assert tree["model"]["encoder"]["vision_model"]["pre_layrnorm"]["scale"].shape == (rc.v_hidden,)  # upstream typo is part of the key
    assert tree["model"]["decoder"]["layers"]["0"]["self_attn"]["q_proj"]["kernel"].shape == (rc.d_model, rc.d_model)  # [in,out]
    bad = unflatten_tree({k: v for k, v in flatten_tree(tree).items() if "visual_projection" not in k})
    with pytest.raises(ValueError, match="Some parameters are missing"):
        model.params = bad  # utils:107-117
    # `params=` kwarg: generate_step / eval_step pass the (replicated) pytree explicitly (main.py:714, 727)
    px, labels, mask, dec_in = batch(rc, 2, 12, seed=33)
    base = model(px.numpy(), dec_in.numpy(), mask.numpy())[0].cpu()
    flat = dict(flatten_tree(tree))
    flat["final_logits_bias"] = flat["final_logits_bias"] + 1.5
    shifted = model(px.numpy(), dec_in.numpy(), mask.numpy(), params=unflatten_tree(flat))[0].cpu()
    assert torch.allclose(shifted, base + 1.5, atol=1e-5)
    out = model(px.numpy(), dec_in.numpy(), mask.numpy(), return_dict=False)
    assert isinstance(out, tuple) and out[0].shape == (2, 12, rc.vocab_size)


def test_prepare_and_update_inputs_for_generation(dev):
    rc, p, model = make_pair(torch.float32, dev)
    px, *_ = batch(rc, 2, 12, seed=34)
    enc = model.encode(px.numpy())
    kw = model.prepare_inputs_for_generation(np.full((2, 1), 2, np.int32), 9, encoder_outputs=enc)
    assert set(kw) == {"past_key_values", "encoder_outputs", "encoder_attention_mask", "decoder_attention_mask", "decoder_position_ids"}
    assert kw["decoder_attention_mask"].shape == (2, 9) and int(kw["decoder_attention_mask"].sum()) == 18  # modeling:669
    assert kw["decoder_position_ids"].tolist() == [[0], [0]]
    out = model.decode(np.full((2, 1), 2, np.int32), enc, decoder_position_ids=kw["decoder_position_ids"], past_key_values=kw["past_key_values"])
    kw = model.update_inputs_for_generation(out, kw)
    assert kw["decoder_position_ids"].tolist() == [[1], [1]] and kw["past_key_values"]["cache_index"] == 1  # modeling:688-693


def test_from_pt_directories(dev, tmp_path):
    """main.py:421-427 with `mbart_from_pt=True`: component directories holding PyTorch weights (pytorch_model.bin /
    model.safetensors in PyTorch layouts and names) graft to the same parameters as the Flax files."""
    from safetensors.numpy import save_file

    from mic_amd import FlaxCLIPVisionMBartForConditionalGeneration as Model
    from mic_amd.params import flatten_tree

    rc, p, model = make_pair(torch.float32, dev)

    def to_pt(prefix, strip):
        out = {}
        for k, v in p.items():
            if not k.startswith(prefix):
                continue
            a, parts = v.numpy(), k[len(strip):].split("/")
            if parts[-1] == "kernel":
                a, parts[-1] = (a.transpose(3, 2, 0, 1) if a.ndim == 4 else a.T), "weight"
            elif parts[-1] in ("scale", "embedding"):
                parts[-1] = "weight"
            out[".".join(parts)] = np.ascontiguousarray(a)
        return out

    clip_dir, mbart_dir = str(tmp_path / "clip"), str(tmp_path / "mbart")
    os.makedirs(clip_dir), os.makedirs(mbart_dir)
    torch.save({k: torch.from_numpy(v) for k, v in to_pt("model/encoder/", "model/encoder/").items()}, os.path.join(clip_dir, "pytorch_model.bin"))
    sd = {"model." + k: v for k, v in to_pt("model/decoder/", "model/").items()}
    sd["model.shared.weight"] = p["model/shared/embedding"].numpy()
    sd["lm_head.weight"] = p["model/shared/embedding"].numpy().copy()  # ForConditionalGeneration extras are dropped
    sd["final_logits_bias"] = np.ones((1, rc.vocab_size), np.float32)
    save_file(sd, os.path.join(mbart_dir, "model.safetensors"))
    json.dump(model.config.clip_vision_config.to_dict(), open(os.path.join(clip_dir, "config.json"), "w"))
    json.dump(model.config.mbart_config.to_dict(), open(os.path.join(mbart_dir, "config.json"), "w"))
    m = Model.from_clip_vision_mbart_pretrained(clip_dir, mbart_dir, seed=3, dtype="float32", mbart_from_pt=True, device=dev)
    got = flatten_tree(m.params)
    for k, v in p.items():
        if k.startswith(("model/encoder/", "model/decoder/", "model/shared/")):
            assert np.array_equal(got[k], v.numpy()), k
    assert np.count_nonzero(got["final_logits_bias"]) == 0


def test_trainer_checkpoint_resume(dev, tmp_path):
    """save_model_checkpoint(with_opt=True) / restore_model_checkpoint (main.py:299-345): a restored trainer continues
    where the saved one would have (parameters, both AdamW moments, step counter -> LR schedule, bias correction and dropout stream)."""
    from mic_amd import Trainer, create_learning_rate_fn

    lr_fn = create_learning_rate_fn(640, 2, 5, 3, 1e-3)
    rc, p, model = make_pair(torch.float32, dev, dropout=0.1)
    b0, b1 = (dict(zip(("pixel_values", "input_ids", "attention_mask", "decoder_input_ids"), (x.numpy() for x in batch(rc, 2, 10, seed=s))))
              for s in (5, 6))
    tr = Trainer(model, lr_fn, weight_decay=0.01, seed=7)
    tr.train_step(b0)
    tr.train_step(b1)
    ck = tr.save_checkpoint(str(tmp_path), with_opt=True)
    assert os.path.basename(ck) == "ckpt-1" and sorted(os.listdir(ck)) == ["config.json", "flax_model.msgpack", "opt_state.msgpack", "training_state.json"]
    assert json.load(open(os.path.join(ck, "training_state.json"))) == {"step": 2}
    ref = tr.train_step(b0)
    ref_loss, ref_params = float(ref["loss"]), model.store.master.clone()
    _, _, model2 = make_pair(torch.float32, dev, seed=99, dropout=0.1)  # different weights until restored
    tr2 = Trainer(model2, lr_fn, weight_decay=0.01, seed=7)
    assert tr2.restore_checkpoint(ck) == 2
    got = tr2.train_step(b0)
    assert float(got["loss"]) == ref_loss and float(got["learning_rate"]) == float(ref["learning_rate"])
    # the forward is bit-identical; backward sums biases / LayerNorm / embedding rows with fp32 atomics (order varies run to run)
    assert (model2.store.master - ref_params).abs().max().item() < 1e-4
    assert (model2.store.m - model.store.m).abs().max().item() < 1e-5 and (model2.store.v - model.store.v).abs().max().item() < 1e-6
    assert tr.save_checkpoint(str(tmp_path / "x"), with_opt=False).endswith("ckpt-2")
    assert sorted(os.listdir(tmp_path / "x" / "ckpt-2")) == ["config.json", "flax_model.msgpack"]


def test_evaluate_loop(dev):
    """The evaluation block of main.py:791-846: per-language eval loss + generate(decoder_start_token_id=lang) + BLEU-1..4."""
    from mic_amd import Trainer, create_learning_rate_fn
    from mic_amd.evaluation import evaluate

    rc, p, model = make_pair(torch.float32, dev)
    tr = Trainer(model, create_learning_rate_fn(640, 2, 5, 3, 1e-3))
    keys = ("pixel_values", "input_ids", "attention_mask", "decoder_input_ids")
    loaders = {"en_XX": [dict(zip(keys, (x.numpy() for x in batch(rc, 2, 10, seed=s)))) for s in (1, 2)],
               "fr_XX": [dict(zip(keys, (x.numpy() for x in batch(rc, 2, 10, seed=3))))]}
    codes = {"en_XX": rc.vocab_size - 10, "fr_XX": rc.vocab_size - 9}
    dec = lambda rows: [" ".join(f"w{int(t)}" for t in row if int(t) > 3 and int(t) < rc.vocab_size - 20) for row in rows]
    res = evaluate(tr, loaders, codes, dec, max_length=10, num_beams=2)
    assert set(res) == {"en_XX", "fr_XX", "loss"} and set(res["en_XX"]) == {"BLEU-1", "BLEU-2", "BLEU-3", "BLEU-4"}
    losses = [float(tr.eval_step(b)["loss"]) for l in loaders.values() for b in l]
    assert abs(res["loss"] - float(np.mean(losses))) < 1e-6
    assert all(0.0 <= v <= 1.0 for v in res["fr_XX"].values())
    # generation really starts from the language code (main.py:820)
    g = model.generate(loaders["fr_XX"][0]["pixel_values"], max_length=10, num_beams=2, decoder_start_token_id=codes["fr_XX"])
    assert (g.sequences[:, 0] == codes["fr_XX"]).all()


def test_output_hidden_states_match_oracle(dev):
    """`output_hidden_states=True` (modeling:499-510 forwards it to the modules): decoder and encoder per-layer states of the
    forward pass against the oracle's."""
    import numpy as np

    from oracle import model_ref as M
    from util_small import batch, make_pair

    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6)
    px, labels, mask, dec_in = batch(rc, 3, 12, seed=9)
    out = model(px.numpy(), dec_in.numpy(), mask.numpy(), output_hidden_states=True)
    plain = model(px.numpy(), dec_in.numpy(), mask.numpy())
    assert torch.equal(out.logits, plain.logits) and set(plain.keys()) == {"logits"}
    with torch.no_grad():
        x = M.layer_norm(M.vision_embeddings(rc, p, px), p[M.V + "pre_layrnorm/scale"], p[M.V + "pre_layrnorm/bias"], rc.v_ln_eps)
        enc = [x]
        for i in range(rc.v_layers):
            x = M.vit_layer(rc, p, x, i)
            enc.append(x)
        ehs = M.dense(x, p, "model/visual_projection")
        B, T = dec_in.shape
        pos = torch.arange(T)[None].expand(B, T)
        _, layers = M.decoder_forward(rc, p, dec_in, mask, pos, ehs, return_layers=True)
        dec = [M.decoder_embed(rc, p, dec_in, pos)] + layers
    assert len(out.decoder_hidden_states) == rc.d_layers + 1 and len(out.encoder_hidden_states) == rc.v_layers + 1
    valid = mask.bool()
    for got, ref in zip(out.encoder_hidden_states, enc):
        assert (got.cpu() - ref).abs().max().item() < 2e-4 * ref.abs().max().item()
    for got, ref in zip(out.decoder_hidden_states, dec):
        assert (got.cpu() - ref)[valid].abs().max().item() < 2e-4 * ref[valid].abs().max().item()
    assert (out.encoder_last_hidden_state.cpu() - ehs).abs().max().item() < 2e-4 * ehs.abs().max().item()
    assert len(model(px.numpy(), dec_in.numpy(), mask.numpy(), output_hidden_states=True, return_dict=False)) == 4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_output_attentions_match_oracle(dev, dtype):
    """`output_attentions=True` (modeling:499-510): decoder self-, cross- and encoder attention weights of every layer against the
    oracle's softmax weights (flax `dot_product_attention_weights`), with and without the hidden states, dict and tuple returns."""
    from oracle import model_ref as M
    from util_small import batch, make_pair

    rc, p, model = make_pair(dtype, dev, gelu="tanh", decoder_ln_eps=1e-6)
    px, labels, mask, dec_in = batch(rc, 3, 12, seed=19)
    out = model(px.numpy(), dec_in.numpy(), mask.numpy(), output_attentions=True)
    plain = model(px.numpy(), dec_in.numpy(), mask.numpy())
    assert torch.equal(out.logits, plain.logits)
    assert list(out.keys()) == ["logits", "decoder_attentions", "cross_attentions", "encoder_attentions"]
    M.ATTN_TAP = []
    try:
        with torch.no_grad():
            B, T = dec_in.shape
            M.forward_logits(rc, p, px, dec_in, mask, torch.arange(T)[None].expand(B, T))
        tap = M.ATTN_TAP
    finally:
        M.ATTN_TAP = None
    assert len(tap) == rc.v_layers + 2 * rc.d_layers
    enc, rest = tap[: rc.v_layers], tap[rc.v_layers:]
    dec, cross = rest[0::2], rest[1::2]
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    valid = mask.bool()[:, None, :, None]  # query rows of padded positions attend over garbage states: not compared
    for name, got_l, ref_l, qmask in (("encoder", out.encoder_attentions, enc, None), ("decoder", out.decoder_attentions, dec, valid),
                                      ("cross", out.cross_attentions, cross, valid)):
        assert len(got_l) == len(ref_l)
        for got, ref in zip(got_l, ref_l):
            assert got.dtype == dtype and tuple(got.shape) == tuple(ref.shape), (name, got.shape, ref.shape)
            diff = (got.float().cpu() - ref).abs()
            if qmask is not None:
                diff = diff * qmask
            assert diff.max().item() < tol, (name, diff.max().item())
            rows = got.float().sum(-1).cpu()
            assert (rows - 1).abs().max().item() < (1e-5 if dtype == torch.float32 else 2e-2)
    # causal + key padding: no weight above the diagonal or on padded keys
    for got in out.decoder_attentions:
        g = got.float().cpu()
        assert torch.triu(g, diagonal=1).abs().max().item() == 0.0
        assert (g * (~mask.bool())[:, None, None, :]).abs().max().item() == 0.0
    both = model(px.numpy(), dec_in.numpy(), mask.numpy(), output_attentions=True, output_hidden_states=True)
    assert list(both.keys()) == ["logits", "decoder_hidden_states", "decoder_attentions", "cross_attentions", "encoder_last_hidden_state",
                                 "encoder_hidden_states", "encoder_attentions"]  # FlaxSeq2SeqLMOutput order (past_key_values is None)
    assert len(model(px.numpy(), dec_in.numpy(), mask.numpy(), output_attentions=True, return_dict=False)) == 4
    assert len(both.to_tuple()) == 7


def test_dropping_a_model_frees_its_device_memory(dev):
    """`del model` (and its Trainer) must give the parameter / optimizer / activation buffers back without waiting for the cycle
    collector: the full-size state is ~16 GB, a sweep that builds models in a loop would otherwise hold several at once."""
    import gc

    from mic_amd import Trainer, create_learning_rate_fn
    from util_small import batch, make_pair

    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6)
        tr = Trainer(model, create_learning_rate_fn(64, 2, 4, 0, 1e-3))
        px, labels, mask, dec_in = batch(rc, 2, 8, seed=3)
        tr.train_step({"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()})
        model.generate(px.numpy(), max_length=6, num_beams=2)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() > base + (1 << 20)
        del tr, model, p
        torch.cuda.synchronize()
        left = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    assert left < (1 << 20), f"{left / 1e6:.1f} MB still allocated after the model and its Trainer were dropped (a reference cycle?)"
