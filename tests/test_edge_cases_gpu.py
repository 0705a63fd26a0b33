"""Edge cases the reference's data can produce (ragged / minimal / maximal inputs), HIP path vs oracle:
batch of one, captions of the minimal length (lang + eos), a full-length caption, max_length 2 and 3 generation (only
forced tokens), an image batch that is not a multiple of any tile, label smoothing, the last vocabulary id."""
import numpy as np
import pytest
import torch

from util_small import make_pair

pytestmark = pytest.mark.gpu


def _mk(rc, lens, T, seed):
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    px = torch.randn(B, rc.image_size, rc.image_size, 3, generator=g).clamp(-1.8, 2.2)
    labels = torch.full((B, T), rc.pad_token_id, dtype=torch.int64)
    mask = torch.zeros((B, T), dtype=torch.int64)
    for b, n in enumerate(lens):  # n content tokens between the language code and eos
        labels[b, 0] = rc.vocab_size - 10 + (b % 4)
        if n:
            labels[b, 1:1 + n] = torch.randint(4, rc.vocab_size - 20, (n,), generator=g)
        if 1 + n < T:
            labels[b, 1 + n] = rc.eos_token_id
        mask[b, : min(T, n + 2)] = 1
    dec_in = torch.full_like(labels, rc.pad_token_id)
    dec_in[:, 1:] = labels[:, :-1]
    return px, labels, mask, dec_in


@pytest.mark.parametrize("lens,T,ls", [((0,), 8, 0.0), ((0, 6, 0), 8, 0.0), ((14,), 16, 0.1), ((1, 0, 5, 13, 2), 16, 0.0), ((7, 7, 7, 7, 7, 7, 7), 9, 0.1)])
def test_loss_and_gradients_ragged_minimal_maximal(dev, lens, T, ls):
    from mic_amd import loss_rows
    from oracle import train_ref

    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0)
    px, labels, mask, dec_in = _mk(rc, lens, T, seed=sum(lens) + T)
    labels[0, 0] = rc.vocab_size - 1  # last vocabulary row (tail of the padded embedding table)
    ref_loss, ref_g = train_ref.loss_and_grads(rc, p, px, labels, mask, dec_in, label_smoothing_factor=ls)
    d = model._dev
    B = len(lens)
    pos = torch.arange(T, dtype=torch.int32, device=dev)[None].expand(B, T).contiguous()
    for compact in (False, True):
        kw = {}
        if compact:
            idx, rl = loss_rows(mask.numpy(), labels.numpy())
            if idx.size == B * T:
                continue  # nothing to compact (the trainer then runs the dense head)
            kw = dict(rows=(d(idx, torch.int32), len(idx)), row_labels=d(rl, torch.int32))
        loss = model.engine.loss_and_grads(d(px, torch.float32), d(dec_in, torch.int32).reshape(-1), pos.reshape(-1), d(mask, torch.int32),
                                           d(labels, torch.int32).reshape(-1), B, T, label_smoothing=ls, **kw)
        torch.cuda.synchronize()
        assert abs(loss.item() - ref_loss.item()) < 3e-5 * max(1.0, abs(ref_loss.item())), (compact, loss.item(), ref_loss.item())
        got = model.store.export_flat("grad")
        for k, rg in ref_g.items():
            sc = rg.abs().max().item()
            if sc > 1e-7:
                e = ((torch.from_numpy(got[k]).reshape(rg.shape) - rg).abs().max() / sc).item()
                assert e < 1e-3, (compact, k, e)


@pytest.mark.parametrize("B,kw", [(1, dict(max_length=2, num_beams=1)), (1, dict(max_length=2, num_beams=4)), (3, dict(max_length=3, num_beams=4, forced_bos_token_id=990)),
                                  (5, dict(max_length=7, num_beams=3)), (1, dict(max_length=6, num_beams=8)), (7, dict(max_length=5, num_beams=1, min_length=4))])
def test_generate_tiny_lengths_and_odd_batches(dev, B, kw):
    from oracle import generation_ref as G
    from oracle import model_ref as M

    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6)
    g = torch.Generator().manual_seed(B + kw["max_length"])
    px = torch.randn(B, rc.image_size, rc.image_size, 3, generator=g).clamp(-1.8, 2.2)
    with torch.no_grad():
        ehs, _ = M.encode(rc, p, px, int32_cast=True)
    K = kw["num_beams"]
    ref = G.generate(lambda rows: G.ModelStepper(rc, p, ehs.repeat_interleave(K, 0) if K > 1 else ehs, kw["max_length"]), B, G.GenDefaults(), **kw)
    out = model.generate(px.numpy(), **kw)
    ref_seq = ref if K == 1 else ref.sequences
    assert np.array_equal(out.sequences.cpu().numpy(), ref_seq)
    if K > 1:
        assert np.allclose(out.scores.cpu().numpy(), ref.scores, rtol=1e-4, atol=1e-4)


def test_train_step_batch_of_one_and_eval(dev):
    from mic_amd import Trainer, create_learning_rate_fn

    rc, p, model = make_pair(torch.float32, dev, dropout=0.1)
    tr = Trainer(model, create_learning_rate_fn(64, 1, 2, 2, 1e-3), label_smoothing_factor=0.1)
    px, labels, mask, dec_in = _mk(rc, (5,), 12, seed=3)
    b = {"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()}
    l0 = float(tr.eval_step(b)["loss"])
    for _ in range(4):
        out = tr.train_step(b)
    assert np.isfinite(float(out["loss"])) and float(tr.eval_step(b)["loss"]) < l0  # it learns its single example


# ---------------------------------------------------------------- sequences longer than one 64x64 attention tile
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_train_longer_than_64_tokens(dev, dtype):
    """seq_len 100 and a ViT of 101 tokens (image 160 / patch 16): every attention shape takes the tiled
    kernels (online-softmax forward, two-pass backward).  The reference README names 64 -> 128 tokens as its next step."""
    from oracle import train_ref

    rc, p, model = make_pair(dtype, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0, max_position_embeddings=256, image_size=160)
    assert rc.v_seq == 101
    T = 100
    px, labels, mask, dec_in = _mk(rc, (98, 3, 70), T, seed=77)
    ref_loss, ref_g = train_ref.loss_and_grads(rc, p, px, labels, mask, dec_in)
    d = model._dev
    B = 3
    pos = torch.arange(T, dtype=torch.int32, device=dev)[None].expand(B, T).contiguous()
    loss = model.engine.loss_and_grads(d(px, torch.float32), d(dec_in, torch.int32).reshape(-1), pos.reshape(-1), d(mask, torch.int32),
                                       d(labels, torch.int32).reshape(-1), B, T)
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    assert abs(loss.item() - ref_loss.item()) < (3e-5 if f32 else 2e-2) * max(1.0, abs(ref_loss.item())), (loss.item(), ref_loss.item())
    got = model.store.export_flat("grad")
    for k, rg in ref_g.items():
        sc = rg.abs().max().item()
        if sc > 1e-6:
            e = ((torch.from_numpy(got[k]).reshape(rg.shape) - rg).abs().max() / sc).item()
            assert e < (1e-3 if f32 else 1e-1), (k, e)


@pytest.mark.parametrize("kw", [dict(num_beams=1), dict(num_beams=3, max_length=90, forced_bos_token_id=990), dict(num_beams=4, max_length=130)])
def test_generate_longer_than_64_tokens(dev, kw):
    """`model.generate(pixels)` without max_length runs to the config default of 200 (gen:205-209, mbart-large-50 config):
    the KV cache is walked in 64-slot chunks.  Token ids exact against the oracle in float32."""
    from oracle import generation_ref as G
    from oracle import model_ref as M

    rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6, max_position_embeddings=256)
    assert model.config.mbart_config.max_length == 200
    B, K = 2, kw["num_beams"]
    L = kw.get("max_length", 200)
    g = torch.Generator().manual_seed(5 + K)
    px = torch.randn(B, rc.image_size, rc.image_size, 3, generator=g).clamp(-1.8, 2.2)
    with torch.no_grad():
        ehs, _ = M.encode(rc, p, px, int32_cast=True)
    okw = dict(kw)
    okw["max_length"] = L
    ref = G.generate(lambda rows: G.ModelStepper(rc, p, ehs.repeat_interleave(K, 0) if K > 1 else ehs, L), B, G.GenDefaults(), **okw)
    out = model.generate(px.numpy(), **kw)
    ref_seq = ref if K == 1 else ref.sequences
    got = out.sequences.cpu().numpy()
    assert got.shape == (B, L)
    assert np.array_equal(got, ref_seq), np.argwhere(got != ref_seq)[:4]
    if K > 1:
        assert np.allclose(out.scores.cpu().numpy(), ref.scores, rtol=1e-4, atol=1e-4)


def test_generate_rejects_lengths_beyond_the_position_table(dev):
    rc, p, model = make_pair(torch.float32, dev)  # max_position_embeddings = 64
    px = np.zeros((1, rc.image_size, rc.image_size, 3), np.float32)
    with pytest.raises(ValueError, match="max_position_embeddings"):
        model.generate(px)  # config default max_length 200 > 64 positions
