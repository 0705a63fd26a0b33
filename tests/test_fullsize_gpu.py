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


def test_fullsize_loss_and_gradients_f32(full):
    from mic_amd import loss_rows
    from oracle import train_ref

    rc, p, models, (px, labels, mask, dec_in), _ = full
    model = models[torch.float32]
    ref_loss, ref_g = train_ref.loss_and_grads(rc, p, px, labels, mask, dec_in)
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
