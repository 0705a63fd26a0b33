"""Per-kernel parity: every HIP op of libmic_hip.so (through the C ABI) against a plain torch fp32 reference of
the same op on the same seeded inputs.  Tolerances: f32 path 2e-5 relative-to-scale (exact-fp32 MFMA, different
summation order); bf16 path = inputs rounded to bf16 on both sides, fp32 accumulate, output rounded to bf16:
<= 1 bf16 ulp of the output scale (2^-8 relative) + accumulation-order noise."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 1.2e-2


def relerr(got, ref):
    ref = ref.float()
    return ((got.float().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


def rnd(shape, g, dtype, scale=1.0):
    x = torch.randn(shape, generator=g) * scale
    return x.to(dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 192), (384, 1000, 128), (37, 72, 64)])
def test_gemm_layouts(dev, dtype, akm, bkm, M, N, K):
    from mic_amd import ops

    if dtype == torch.bfloat16 and ((akm and M % 8) or (bkm and N % 8)):
        pytest.skip("k-major bf16 operands need the contiguous dim % 8 == 0")
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + akm * 2 + bkm)
    A = rnd((K, M) if akm else (M, K), g, dtype)
    B = rnd((K, N) if bkm else (N, K), g, dtype)
    ref = (A.float().T if akm else A.float()) @ (B.float() if bkm else B.float().T)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=dev)
    ops.gemm(A.to(dev), B.to(dev), out, M, N, K, a_kmajor=akm, b_kmajor=bkm)
    torch.cuda.synchronize()
    assert relerr(out, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DT)
def test_gemm_epilogue_full(dev, dtype):
    """bias + save pre-activation + activation + dropout + residual, then the backward-side options."""
    from mic_amd import ops

    M, N, K = 192, 256, 128
    g = torch.Generator().manual_seed(5)
    A, B = rnd((M, K), g, dtype), rnd((N, K), g, dtype, 0.1)
    bias = torch.randn(N, generator=g)
    R = rnd((M, N), g, dtype)
    for act_id, act in ((1, lambda x: torch.nn.functional.gelu(x)), (2, lambda x: torch.nn.functional.gelu(x, approximate="tanh")),
                        (3, lambda x: x * torch.sigmoid(1.702 * x))):
        out = torch.empty((M, N), dtype=dtype, device=dev)
        z = torch.empty((M, N), dtype=dtype, device=dev)
        ops.gemm(A.to(dev), B.to(dev), out, M, N, K, bias=bias.to(dev), act=act_id, zout=z, residual=R.to(dev), dropout_p=0.25,
                 dropout_seed=77)
        keep = ops.dropout_mask(M * N, 0.25, 77, dev).cpu().reshape(M, N).float()
        torch.cuda.synchronize()
        assert abs(keep.mean().item() - 0.75) < 0.02
        zr = A.float() @ B.float().T + bias
        zr_s = zr.to(dtype).float()
        ref = act(zr_s) * keep / 0.75 + R.float()
        assert relerr(z, zr) < tol(dtype)
        assert relerr(out, ref) < tol(dtype)
        # backward: dz = (dy @ W) * act'(z)
        dY = rnd((M, K), g, dtype)
        W = rnd((K, N), g, dtype, 0.1)  # stored [out=K][in=N]; dX = dY @ W  -> b_kmajor
        dz = torch.empty((M, N), dtype=dtype, device=dev)
        ops.gemm(dY.to(dev), W.to(dev), dz, M, N, K, b_kmajor=True, zin=z, dact=act_id)
        zz = z.float().cpu().requires_grad_(True)
        act(zz).backward(torch.ones_like(zz))
        refdz = (dY.float() @ W.float()) * zz.grad
        torch.cuda.synchronize()
        assert relerr(dz, refdz) < tol(dtype) * 1.5
    # fp32 output with accumulate (weight gradients), bf16/f32 inputs
    dW = torch.ones((N, K), dtype=torch.float32, device=dev)
    dYm = rnd((M, N), g, dtype)
    ops.gemm(dYm.to(dev), A.to(dev), dW, N, K, M if M % 64 == 0 else M, a_kmajor=True, b_kmajor=True, accumulate=True)
    torch.cuda.synchronize()
    assert relerr(dW, dYm.float().T @ A.float() + 1.0) < (2e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, True)])
def test_gemm_big_tiles_split_k_and_grouped(dev, akm, bkm):
    """256x256 configuration (chosen when a launch has >= 200 big tiles), split-K via fp32 atomics, grouped launches."""
    from mic_amd import ops

    g = torch.Generator().manual_seed(11 + akm + 2 * bkm)
    dt = torch.bfloat16

    def mk(M, N, K):
        A = rnd((K, M) if akm else (M, K), g, dt)
        B = rnd((K, N) if bkm else (N, K), g, dt)
        ref = (A.float().T if akm else A.float()) @ (B.float() if bkm else B.float().T)
        return A.to(dev), B.to(dev), ref

    # big tiles with ragged edges in both dims: 15 x 14 = 210 tiles of 256
    M, N, K = 3720, 3400, 128
    A, B, ref = mk(M, N, K)
    bias = torch.randn(N, generator=g)
    out = torch.full((M, N), float("nan"), dtype=dt, device=dev)
    ops.gemm(A, B, out, M, N, K, a_kmajor=akm, b_kmajor=bkm, bias=bias.to(dev))
    torch.cuda.synchronize()
    assert relerr(out, ref + bias) < tol(dt)
    # split-K: tiny output, long reduction, fp32 atomics into a zeroed C (small and big tile configs)
    for (M, N, K, sk) in ((256, 136, 4096, 8), (1024, 1024, 8192, 16)):
        A, B, ref = mk(M, N, K)
        c = torch.zeros((M, N), dtype=torch.float32, device=dev)
        ops.gemm(A, B, c, M, N, K, a_kmajor=akm, b_kmajor=bkm, split_k=sk)
        torch.cuda.synchronize()
        assert relerr(c, ref) < 2e-3
    # grouped: 5 problems of different shapes in one launch
    probs, refs, outs = [], [], []
    for (M, N, K) in ((768, 768, 256), (1024, 264, 128), (136, 1024, 192), (2304, 768, 128), (384, 384, 64)):
        A, B, ref = mk(M, N, K)
        c = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
        probs.append(ops.gemm_args(A, B, c, M, N, K, a_kmajor=akm, b_kmajor=bkm))
        refs.append(ref)
        outs.append((A, B, c))
    ops.gemm_grouped(probs)
    torch.cuda.synchronize()
    for (A, B, c), ref in zip(outs, refs):
        assert relerr(c, ref) < 2e-3


def test_gemm_rejects_bad_args(dev):
    from mic_amd import _lib, ops

    a = torch.zeros(64, 40, dtype=torch.bfloat16, device=dev)
    with pytest.raises(_lib.MicError):
        ops.gemm(a, a, torch.zeros(64, 64, dtype=torch.bfloat16, device=dev), 64, 64, 40)  # K % 64 != 0
    with pytest.raises(_lib.MicError):
        ops.gemm(a.cpu(), a, torch.zeros(64, 64, dtype=torch.bfloat16, device=dev), 64, 64, 64)  # host pointer


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,width", [(50, 768), (64, 1024), (7, 128)])
def test_layernorm_fwd_bwd(dev, dtype, rows, width):
    from mic_amd import ops

    g = torch.Generator().manual_seed(rows + width)
    x = rnd((rows, width), g, dtype, 2.0)
    gamma, beta = 1 + 0.1 * torch.randn(width, generator=g), 0.1 * torch.randn(width, generator=g)
    dy, dres = rnd((rows, width), g, dtype), rnd((rows, width), g, dtype)
    y = torch.empty_like(x, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.layernorm_fwd(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, y, mean, rstd)
    xr = x.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (width,), gr, br, 1e-5)
    assert relerr(y, yr.detach()) < tol(dtype)
    yr.backward(dy.float())
    dx, dxm = torch.empty_like(x, device=dev), torch.empty_like(x, device=dev)
    dg, db = torch.zeros(width, device=dev), torch.zeros(width, device=dev)
    ops.layernorm_bwd(x.to(dev), gamma.to(dev), mean, rstd, dy.to(dev), dx, dg, db, dres=dres.to(dev), dxm=dxm, dropout_p=0.1,
                      dropout_seed=9)
    keep = ops.dropout_mask(rows * width, 0.1, 9, dev).cpu().reshape(rows, width).float()
    torch.cuda.synchronize()
    refdx = xr.grad + dres.float()
    assert relerr(dx, refdx) < tol(dtype)
    assert relerr(dxm, refdx.to(dtype).float() * keep / 0.9) < tol(dtype)
    assert relerr(dg, gr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert relerr(db, br.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    # dropout after LN (embedding LN) and its backward
    y2 = torch.empty_like(x, device=dev)
    ops.layernorm_fwd(x.to(dev), gamma.to(dev), beta.to(dev), 1e-5, y2, mean, rstd, dropout_p=0.1, dropout_seed=3)
    keep2 = ops.dropout_mask(rows * width, 0.1, 3, dev).cpu().reshape(rows, width).float()
    assert relerr(y2, yr.detach().to(dtype).float() * keep2 / 0.9) < tol(dtype)
    dx2 = torch.empty_like(x, device=dev)
    ops.layernorm_bwd(x.to(dev), gamma.to(dev), mean, rstd, dy.to(dev), dx2, None, None, in_dropout_p=0.1, in_dropout_seed=3)
    xr2 = x.float().clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr2, (width,), gamma, beta, 1e-5).backward(dy.float() * keep2 / 0.9)
    torch.cuda.synchronize()
    assert relerr(dx2, xr2.grad) < tol(dtype)


def _attn_ref(q, k, v, causal, key_mask):
    # q [B,Tq,H,64] k,v [B,Tk,H,64]
    s = torch.einsum("bthd,bshd->bhts", q / 8.0, k)
    Tq, Tk = q.shape[1], k.shape[1]
    allowed = torch.ones(q.shape[0], 1, Tq, Tk, dtype=torch.bool)
    if causal:
        allowed = allowed & torch.tril(torch.ones(Tq, Tk, dtype=torch.bool))[None, None]
    if key_mask is not None:
        allowed = allowed & key_mask.bool()[:, None, None, :]
    s = s.masked_fill(~allowed, float("-inf"))
    return torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), v)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Tq,Tk,H,causal,masked", [(50, 50, 12, False, False), (64, 64, 16, True, True), (64, 50, 16, False, False),
                                                   (12, 12, 2, True, True), (12, 10, 2, False, False), (33, 64, 3, False, True),
                                                   # longer than one 64x64 tile: online-softmax forward, two-pass backward
                                                   (128, 128, 4, True, True), (200, 200, 2, True, True), (197, 197, 3, False, False),
                                                   (130, 50, 2, False, False), (64, 197, 2, False, True), (65, 65, 1, True, False)])
def test_attention_fwd_bwd(dev, dtype, Tq, Tk, H, causal, masked):
    from mic_amd import ops

    B, D = 3, 64
    g = torch.Generator().manual_seed(Tq * 100 + Tk)
    fused = Tq == Tk  # self-attention reads q,k,v out of one fused [rows][3*H*D] projection
    if fused:
        qkv = rnd((B * Tq, 3 * H * D), g, dtype)
        q, k, v = qkv[:, : H * D], qkv[:, H * D: 2 * H * D], qkv[:, 2 * H * D:]
        ldq = ldk = ldv = 3 * H * D
        qkv_d = qkv.to(dev)
        qd, kd, vd = qkv_d[:, : H * D], qkv_d[:, H * D: 2 * H * D], qkv_d[:, 2 * H * D:]
    else:
        q = rnd((B * Tq, H * D), g, dtype)
        kv = rnd((B * Tk, 2 * H * D), g, dtype)
        k, v = kv[:, : H * D], kv[:, H * D:]
        ldq, ldk, ldv = H * D, 2 * H * D, 2 * H * D
        qd, kv_d = q.to(dev), kv.to(dev)
        kd, vd = kv_d[:, : H * D], kv_d[:, H * D:]
    key_mask = None
    if masked:
        key_mask = torch.ones(B, Tk, dtype=torch.int32)
        key_mask[1, Tk - 3:] = 0
        key_mask[2, Tk // 2:] = 0
    out = torch.empty((B * Tq, H * D), dtype=dtype, device=dev)
    lse = torch.empty((B, H, Tq), device=dev)
    km_d = key_mask.to(dev) if key_mask is not None else None
    ops.attn_fwd(qd, kd, vd, out, B, H, Tq, Tk, ldq=ldq, ldk=ldk, ldv=ldv, ldo=H * D, key_mask=km_d, causal=causal, lse=lse)
    qr = q.float().reshape(B, Tq, H, D).clone().requires_grad_(True)
    kr = k.float().reshape(B, Tk, H, D).clone().requires_grad_(True)
    vr = v.float().reshape(B, Tk, H, D).clone().requires_grad_(True)
    ref = _attn_ref(qr, kr, vr, causal, key_mask)
    torch.cuda.synchronize()
    assert relerr(out, ref.detach().reshape(B * Tq, H * D)) < tol(dtype)
    do = rnd((B * Tq, H * D), g, dtype)
    ref.backward(do.float().reshape(B, Tq, H, D))
    dq = torch.empty((B * Tq, H * D), dtype=dtype, device=dev)
    dkv = torch.empty((B * Tk, 2 * H * D), dtype=dtype, device=dev)
    ops.attn_bwd(qd, kd, vd, out, do.to(dev), lse, dq, dkv[:, : H * D], dkv[:, H * D:], B, H, Tq, Tk, ldq=ldq, ldk=ldk, ldv=ldv,
                 ldo=H * D, lddo=H * D, lddq=H * D, lddk=2 * H * D, lddv=2 * H * D, key_mask=km_d, causal=causal)
    torch.cuda.synchronize()
    t = tol(dtype) * (1 if dtype == torch.float32 else 2.5)
    assert relerr(dq, qr.grad.reshape(B * Tq, H * D)) < t
    assert relerr(dkv[:, : H * D], kr.grad.reshape(B * Tk, H * D)) < t
    assert relerr(dkv[:, H * D:], vr.grad.reshape(B * Tk, H * D)) < t


@pytest.mark.parametrize("dtype", DT)
def test_attn_decode_and_kv_append(dev, dtype):
    from mic_amd import ops

    R, H, D, L = 8, 4, 64, 16
    g = torch.Generator().manual_seed(3)
    kc = torch.zeros((R, L, H * D), dtype=dtype, device=dev)
    vc = torch.zeros_like(kc)
    hist_k = rnd((R, L, H * D), g, dtype)
    hist_v = rnd((R, L, H * D), g, dtype)
    for t in range(6):
        ops.kv_append(hist_k[:, t].contiguous().to(dev), hist_v[:, t].contiguous().to(dev), kc, vc, R, H * D, L, t, ldk=H * D, ldv=H * D)
    src = torch.randint(0, R, (R, L), generator=g, dtype=torch.int32)
    q = rnd((R, H * D), g, dtype)
    out = torch.empty((R, H * D), dtype=dtype, device=dev)
    cur = 5
    ops.attn_decode(q.to(dev), kc, vc, out, R, H, L, cur, ldq=H * D, ldo=H * D, src_row=src.to(dev))
    torch.cuda.synchronize()
    kk = torch.stack([hist_k[src[r, : cur + 1].long(), torch.arange(cur + 1)] for r in range(R)]).float().reshape(R, cur + 1, H, D)
    vv = torch.stack([hist_v[src[r, : cur + 1].long(), torch.arange(cur + 1)] for r in range(R)]).float().reshape(R, cur + 1, H, D)
    ref = _attn_ref(q.float().reshape(R, 1, H, D), kk, vv, False, None).reshape(R, H * D)
    assert relerr(out, ref) < tol(dtype)
    # cross-attention form: rows share cache row r // row_div, all S slots valid
    ops.attn_decode(q.to(dev), kc, vc, out, R, H, L, 5, ldq=H * D, ldo=H * D, row_div=4)
    torch.cuda.synchronize()
    idx = torch.arange(R) // 4
    ref = _attn_ref(q.float().reshape(R, 1, H, D), hist_k[idx, :6].float().reshape(R, 6, H, D), hist_v[idx, :6].float().reshape(R, 6, H, D),
                    False, None).reshape(R, H * D)
    assert relerr(out, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("L,cur", [(200, 199), (200, 64), (130, 70), (64, 63)])
def test_attn_decode_long_cache(dev, dtype, L, cur):
    """Caches longer than 64 slots (generate()'s config default max_length is 200, gen:205-209): chunked walk with a running
    (max, sum, output) triple; beam-parent slot indirection included."""
    from mic_amd import ops

    R, H, D = 6, 3, 64
    g = torch.Generator().manual_seed(L * 1000 + cur)
    hist_k, hist_v = rnd((R, L, H * D), g, dtype), rnd((R, L, H * D), g, dtype)
    hist_k[2, 100 % L] *= 6.0  # a late dominant key: the running max must move in a later chunk
    src = torch.randint(0, R, (R, L), generator=g, dtype=torch.int32)
    q = rnd((R, H * D), g, dtype)
    out = torch.empty((R, H * D), dtype=dtype, device=dev)
    ops.attn_decode(q.to(dev), hist_k.to(dev), hist_v.to(dev), out, R, H, L, cur, ldq=H * D, ldo=H * D, src_row=src.to(dev))
    torch.cuda.synchronize()
    n = cur + 1
    kk = torch.stack([hist_k[src[r, :n].long(), torch.arange(n)] for r in range(R)]).float().reshape(R, n, H, D)
    vv = torch.stack([hist_v[src[r, :n].long(), torch.arange(n)] for r in range(R)]).float().reshape(R, n, H, D)
    ref = _attn_ref(q.float().reshape(R, 1, H, D), kk, vv, False, None).reshape(R, H * D)
    assert relerr(out, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DT)
def test_vit_embed_ops(dev, dtype):
    from mic_amd import ops

    B, img, ps, W = 2, 64, 32, 128
    g = torch.Generator().manual_seed(0)
    px = torch.randn(B, img, img, 3, generator=g) * 1.5
    gg = img // ps
    patches = torch.empty((B * gg * gg, ps * ps * 3), dtype=dtype, device=dev)
    for trunc in (False, True):
        ops.im2col(px.to(dev), patches, B, img, ps, ps * ps * 3, trunc_int32=trunc)
        src = torch.trunc(px) if trunc else px
        ref = src.reshape(B, gg, ps, gg, ps, 3).permute(0, 1, 3, 2, 4, 5).reshape(B * gg * gg, -1)
        torch.cuda.synchronize()
        assert torch.equal(patches.float().cpu(), ref.to(dtype).float())
    S = gg * gg + 1
    po = rnd((B * gg * gg, W), g, dtype)
    cls, pos = torch.randn(W, generator=g), torch.randn(S, W, generator=g)
    x = torch.empty((B * S, W), dtype=dtype, device=dev)
    ops.vit_assemble(po.to(dev), cls.to(dev), pos.to(dev), x, B, S, W, W)
    ref = torch.cat([cls.expand(B, 1, W), po.float().reshape(B, S - 1, W)], 1) + pos
    torch.cuda.synchronize()
    assert relerr(x, ref.reshape(B * S, W)) < tol(dtype)
    dx = rnd((B * S, W), g, dtype)
    dpatch = torch.empty((B * (S - 1), W), dtype=dtype, device=dev)
    dcls, dpos = torch.zeros(W, device=dev), torch.zeros(S, W, device=dev)
    ops.vit_assemble_bwd(dx.to(dev), dpatch, dcls, dpos, B, S, W, W)
    torch.cuda.synchronize()
    d3 = dx.float().reshape(B, S, W)
    assert torch.equal(dpatch.float().cpu(), d3[:, 1:].reshape(-1, W))
    assert relerr(dpos, d3.sum(0)) < 1e-5 and relerr(dcls, d3[:, 0].sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B", [9, 64])
def test_vit_assemble_bwd_batch_slices(dev, dtype, B):
    """the ViT embedding backward at the train step's shape (50 tokens x 768): bf16 storage cuts the batch into slices that meet by fp32
    atomics (B = 9: four slices with a ragged last one; 64: eight), fp32 keeps one add per element"""
    from mic_amd import ops

    S, W = 50, 768
    g = torch.Generator().manual_seed(B)
    dx = rnd((B * S, W), g, dtype)
    dpatch = torch.full((B * (S - 1) + 3, W), 7.0, dtype=dtype, device=dev)
    dcls, dpos = torch.zeros(W, device=dev), torch.zeros(S, W, device=dev)
    ops.vit_assemble_bwd(dx.to(dev), dpatch, dcls, dpos, B, S, W, W)
    torch.cuda.synchronize()
    d3 = dx.float().reshape(B, S, W)
    assert torch.equal(dpatch[: B * (S - 1)].float().cpu(), d3[:, 1:].reshape(-1, W)) and float((dpatch[B * (S - 1):].float() - 7.0).abs().max()) == 0.0
    assert relerr(dpos, d3.sum(0)) < 1e-5 and relerr(dcls, d3[:, 0].sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", DT)
def test_token_embed_fwd_bwd(dev, dtype):
    from mic_amd import ops

    V, W, rows = 300, 128, 40
    g = torch.Generator().manual_seed(1)
    table = rnd((V, W), g, dtype)
    pos_table = torch.randn(66, W, generator=g)
    ids = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32)
    ids[5] = ids[6] = ids[7]  # repeated rows must accumulate in backward
    pos = torch.randint(0, 64, (rows,), generator=g, dtype=torch.int32)
    h = torch.empty((rows, W), dtype=dtype, device=dev)
    ops.embed_fwd(ids.to(dev), pos.to(dev), table.to(dev), pos_table.to(dev), math.sqrt(W), h, rows, W)
    ref = table.float()[ids.long()] * math.sqrt(W) + pos_table[pos.long() + 2]
    torch.cuda.synchronize()
    assert relerr(h, ref) < tol(dtype)
    dh = rnd((rows, W), g, dtype)
    dt, dp = torch.zeros((V, W), device=dev), torch.zeros((66, W), device=dev)
    ops.embed_bwd(ids.to(dev), pos.to(dev), dh.to(dev), math.sqrt(W), dt, dp, rows, W)
    rt, rp = torch.zeros(V, W), torch.zeros(66, W)
    rt.index_add_(0, ids.long(), dh.float() * math.sqrt(W))
    rp.index_add_(0, pos.long() + 2, dh.float())
    torch.cuda.synchronize()
    assert relerr(dt, rt) < 1e-5 and relerr(dp, rp) < 1e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("ls", [0.0, 0.1])
def test_cross_entropy(dev, dtype, ls):
    from mic_amd import ops
    from oracle import train_ref

    rows, V, Vpad = 24, 1003, 1024
    g = torch.Generator().manual_seed(2)
    logits = torch.zeros((rows, Vpad), dtype=dtype)
    logits[:, :V] = rnd((rows, V), g, dtype, 3.0)
    labels = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32)
    mask = (torch.rand(rows, generator=g) > 0.3).to(torch.int32)
    ld = logits.to(dev)
    lse, rl = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    loss, denom = torch.empty(1, device=dev), torch.empty(1, device=dev)
    ops.ce_rows(ld, Vpad, V, labels.to(dev), mask.to(dev), ls, lse, rl, rows)
    ops.ce_reduce(rl, mask.to(dev), loss, denom, rows)
    ops.ce_bwd(ld, Vpad, V, Vpad, labels.to(dev), mask.to(dev), ls, lse, denom, rows)
    lr = logits[:, :V].float().clone().requires_grad_(True)
    ref = train_ref.loss_fn(lr[None], labels[None], mask[None], ls)
    ref.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    assert denom.item() == mask.sum().item()
    assert relerr(ld[:, :V], lr.grad) < (1e-4 if dtype == torch.float32 else 1.2e-2)
    assert ld[:, V:].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", DT)
def test_colsum_cast(dev, dtype):
    from mic_amd import ops

    g = torch.Generator().manual_seed(4)
    x = rnd((300, 200), g, dtype)
    out = torch.empty(200, device=dev)
    ops.colsum(x.to(dev), out, 300, 200, 200)
    torch.cuda.synchronize()
    assert relerr(out, x.float().sum(0)) < 1e-5
    y = torch.empty((300, 200), dtype=torch.bfloat16, device=dev)
    ops.cast(x.float().to(dev), y)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x.float().to(torch.bfloat16))


def test_adamw_matches_oracle(dev):
    from mic_amd import ops
    from oracle import train_ref

    n = 4096
    g = torch.Generator().manual_seed(6)
    p, m, v, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.01, torch.randn(n, generator=g)
    pd, md, vd = p.to(dev), m.to(dev), v.to(dev)
    plp = torch.empty(n, dtype=torch.bfloat16, device=dev)
    count, lr = 6, 3e-4
    hyper = torch.tensor([lr, float(count + 1)], device=dev)
    ops.adamw(pd, md, vd, gr.to(dev), plp, hyper, 0.9, 0.999, 1e-8, 0.01)
    rp, rm, rv = train_ref.adamw_update(p, gr, m, v, count, lr, 0.9, 0.999, 1e-8, 0.01)
    torch.cuda.synchronize()
    assert relerr(pd, rp) < 1e-6 and relerr(md, rm) < 1e-6 and relerr(vd, rv) < 1e-6, (relerr(pd, rp), relerr(md, rm), relerr(vd, rv))
    assert torch.equal(plp.cpu(), pd.cpu().to(torch.bfloat16))


def test_cu_masked_stream_runs_kernels_and_rejects_bad_ranges(dev):
    """mic_stream_create_cu_masked: a stream over a proper subset of the CUs runs this library's launches (same results as the
    default stream); empty, oversized and whole-device ranges are refused with an error, not a crash."""
    from mic_amd import ops
    from mic_amd._lib import MicError

    total = torch.cuda.get_device_properties(dev).multi_processor_count
    n = 1 << 16
    g = torch.Generator().manual_seed(3)
    p, m, v, gr = (torch.randn(n, generator=g).to(dev) for _ in range(4))
    v = v.abs()
    hyper = torch.tensor([1e-3, 2.0], device=dev)
    ref = [t.clone() for t in (p, m, v)]
    ops.adamw(ref[0], ref[1], ref[2], gr, None, hyper, 0.9, 0.999, 1e-8, 0.0)
    torch.cuda.synchronize()
    for first, cnt in ((0, 8), (total - 32, 32), (8, total - 8)):
        st = ops.cu_masked_stream(first, cnt, dev)
        got = [t.clone() for t in (p, m, v)]
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            ops.adamw(got[0], got[1], got[2], gr, None, hyper, 0.9, 0.999, 1e-8, 0.0)
        st.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    for first, cnt in ((0, 0), (0, total), (total - 8, 16), (-1, 8)):
        with pytest.raises(MicError):
            ops.cu_masked_stream(first, cnt, dev)


@pytest.mark.parametrize("rows,width", [(37, 256), (5, 1024), (1003, 64), (64, 12)])
def test_adamw_rows_split_is_exact(dev, rows, width):
    """mic_adamw over a [rows][width] slice == mic_adamw_rows on the unflagged rows followed by mic_adamw_rows on the flagged ones,
    bit for bit (AdamW is elementwise), with the flags built by mic_row_flags from ids that repeat and that fall outside the range."""
    from mic_amd import ops

    n = rows * width
    g = torch.Generator().manual_seed(61)
    p, m, v, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.01, torch.randn(n, generator=g)
    hyper = torch.tensor([3e-4, 7.0], device=dev)
    ids = torch.randint(0, rows, (max(rows // 3, 2),), generator=g, dtype=torch.int32)
    ids = torch.cat([ids, ids[:2], torch.tensor([-1, rows, rows + 5], dtype=torch.int32)])  # repeats; out-of-range ids are ignored
    flags = torch.full((rows,), 7, dtype=torch.uint8, device=dev)
    ops.row_flags(ids.to(dev), ids.numel(), flags)
    want = torch.zeros(rows, dtype=torch.uint8)
    want[ids[(ids >= 0) & (ids < rows)].long()] = 1
    assert torch.equal(flags.cpu(), want) and 0 < int(want.sum()) < rows

    def state():
        return [t.clone().to(dev) for t in (p, m, v)] + [torch.zeros(n, dtype=torch.bfloat16, device=dev)]

    a, b = state(), state()
    grd = gr.to(dev)
    ops.adamw(a[0], a[1], a[2], grd, a[3], hyper, 0.9, 0.999, 1e-8, 0.01, grad_scale=0.5)
    ops.adamw_rows(rows, width, flags, 0, b[0], b[1], b[2], grd, b[3], hyper, 0.9, 0.999, 1e-8, 0.01, grad_scale=0.5)
    mid = [t.clone() for t in b]
    ops.adamw_rows(rows, width, flags, 1, b[0], b[1], b[2], grd, b[3], hyper, 0.9, 0.999, 1e-8, 0.01, grad_scale=0.5)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # the first pass left the flagged rows alone
    fl = want.bool().to(dev)
    assert torch.equal(mid[0].reshape(rows, width)[fl], p.to(dev).reshape(rows, width)[fl])
    assert torch.equal(mid[0].reshape(rows, width)[~fl], a[0].reshape(rows, width)[~fl])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("V,Vpad,k", [(5003, 5008, 8), (1003, 1024, 8), (250054, 250112, 8), (5003, 5008, 24), (250054, 250112, 40), (1003, 1024, 64)])
def test_row_lse_topk(dev, dtype, V, Vpad, k):
    """k = 2 * num_beams: the 8-, 16- (not here: beam tests), 32- and 64-wide builds of the streaming kernel"""
    from mic_amd import ops
    from oracle import generation_ref as G

    R = 6
    g = torch.Generator().manual_seed(8)
    logits = torch.zeros((R, Vpad), dtype=dtype)
    logits[:, :V] = rnd((R, V), g, dtype, 2.0)
    logits[2, 100:140] = logits[2, :V].max() + 1  # exact ties at the top: lowest index first
    bias = torch.tensor([0.0, -1e7, -3.5, 0.0, -1e7, 2.0])
    tv, ti = torch.empty((R, k), device=dev), torch.empty((R, k), dtype=torch.int32, device=dev)
    x = logits[:, :V].float().numpy()
    for forced, sup in ((-1, False), (17, False), (-1, True)):
        ops.row_lse_topk(logits.to(dev), Vpad, V, k, tv, ti, R, forced_token=forced, suppress_eos=sup, eos_token_id=2, row_bias=bias.to(dev))
        lp = G.log_softmax(x)
        if sup:
            lp[:, 2] = -np.inf
        if forced >= 0:
            lp = G.ForcedBOS(forced)(None, lp, 1)
        lp = lp + bias.numpy()[:, None]
        rv, ri = G.top_k(lp, k)
        torch.cuda.synchronize()
        assert np.array_equal(ti.cpu().numpy(), ri.astype(np.int32)), (forced, sup)
        got = tv.cpu().numpy()
        fin = np.isfinite(rv)
        assert np.array_equal(np.isfinite(got), fin)
        assert np.allclose(got[fin], rv[fin], rtol=0, atol=2e-5 * max(1.0, np.abs(rv[fin]).max()))
    # greedy form: raw logits, k = 1 == first-max argmax
    tv1, ti1 = torch.empty((R, 1), device=dev), torch.empty((R, 1), dtype=torch.int32, device=dev)
    ops.row_lse_topk(logits.to(dev), Vpad, V, 1, tv1, ti1, R, raw_logits=True)
    torch.cuda.synchronize()
    assert np.array_equal(ti1.cpu().numpy()[:, 0], np.argmax(x, axis=-1))


@pytest.mark.parametrize("Mo,No,K,kvalid", [(3072, 1024, 4096, 4096), (768, 768, 3200, 3137), (1024, 768, 3200, 3200), (248, 136, 256, 200),
                                           (5003 // 8 * 8, 256, 192, 130)])
def test_gemm_a_rowsum_is_the_bias_gradient(dev, Mo, No, K, kvalid):
    """mic_gemm a_rowsum: with A = dy^T (k-major) the weight-gradient GEMM dW = dy^T x also yields colsum(dy[:kvalid]) — every
    tile configuration (256/128/64 tiles, K-groups) and a partially valid last K-tile; C itself is unchanged."""
    from mic_amd import ops

    g = torch.Generator().manual_seed(Mo + K)
    dy = (torch.randn(K, Mo, generator=g)).to(torch.bfloat16).to(dev)
    x = (torch.randn(K, No, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    x[kvalid:] = 0  # reduction padding rows of x are zero (dW unaffected), dy's may hold anything
    dW = torch.empty(Mo, No, dtype=torch.float32, device=dev)
    db = torch.zeros(Mo, dtype=torch.float32, device=dev)
    ops.gemm(dy, x, dW, Mo, No, K, a_kmajor=True, b_kmajor=True, a_rowsum=db, rowsum_k=kvalid)
    ref_db = dy[:kvalid].float().sum(0)
    ref_dW = dy.float().T @ x.float()
    assert (db - ref_db).abs().max().item() < 2e-3 * max(1.0, ref_db.abs().max().item())
    assert (dW - ref_dW).abs().max().item() < 2e-3 * ref_dW.abs().max().item()
    # grouped launch: each problem carries its own vector
    db2 = torch.zeros(Mo, dtype=torch.float32, device=dev)
    db3 = torch.zeros(Mo, dtype=torch.float32, device=dev)
    dW2 = torch.empty_like(dW)
    ops.gemm_grouped([ops.gemm_args(dy, x, dW, Mo, No, K, a_kmajor=True, b_kmajor=True, a_rowsum=db2, rowsum_k=kvalid),
                      ops.gemm_args(dy, x, dW2, Mo, No, K, a_kmajor=True, b_kmajor=True, a_rowsum=db3)])
    assert (db2 - ref_db).abs().max().item() < 2e-3 * max(1.0, ref_db.abs().max().item())
    assert (db3 - dy.float().sum(0)).abs().max().item() < 2e-3 * max(1.0, dy.float().sum(0).abs().max().item())


def test_gemm_a_rowsum_needs_a_kmajor(dev):
    """the row-sum code exists only in the A-k-major kernels (include/mic_hip.h: a_rowsum): any other layout is refused, not ignored"""
    from mic_amd import ops
    from mic_amd._lib import MicError

    a = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    c = torch.empty(256, 256, dtype=torch.float32, device=dev)
    rs = torch.zeros(256, dtype=torch.float32, device=dev)
    with pytest.raises(MicError):
        ops.gemm(a, w, c, 256, 256, 128, a_rowsum=rs)


def test_gemm_split_k_slabs_and_sum(dev):
    """split_k with split_stride: every split stores its partial tile into its own fp32 slab (no atomics, no zero-fill);
    mic_sum_slabs adds the slabs and rounds once.  Same result as the un-split GEMM; deterministic run to run."""
    from mic_amd import ops

    g = torch.Generator().manual_seed(9)
    M, N, K = 520, 1024, 64 * 96
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w = (torch.randn(K, N, generator=g) * 0.05).to(torch.bfloat16).to(dev)  # k-major B, like the tied embedding in the head dX
    ref = a.float() @ w.float()
    for nsp in (8, 16, 6):
        slab = 576 * N
        ws = torch.full((nsp * 576, N), float("nan"), dtype=torch.float32, device=dev)  # garbage in: slabs are fully overwritten
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(a, w, ws, M, N, K, b_kmajor=True, split_k=nsp, split_stride=slab)
        ops.sum_slabs(ws, nsp, slab, out, M, N, ws.stride(0), out.stride(0))
        assert torch.isfinite(out.float()).all()
        assert (out.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item()
        out2 = torch.empty_like(out)
        ops.gemm(a, w, ws, M, N, K, b_kmajor=True, split_k=nsp, split_stride=slab)
        ops.sum_slabs(ws, nsp, slab, out2, M, N, ws.stride(0), out2.stride(0))
        assert torch.equal(out, out2)
    o32 = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.sum_slabs(ws, 6, slab, o32, M, N, ws.stride(0), o32.stride(0))
    assert (o32 - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    from mic_amd._lib import MicError

    with pytest.raises(MicError):
        ops.gemm(a, w, ws, M, N, K, b_kmajor=True, split_k=8, split_stride=100)


# ---------------------------------------------------------------- head GEMM softmax partials and their consumers
@pytest.mark.parametrize("M,V,Vpad", [(300, 1003, 1024), (70, 5003, 5056), (1024, 250054, 250112)])
def test_head_rowstat_topk_and_ce_from_tile_partials(dev, M, V, Vpad):
    """mic_gemm's per-tile (max, sum exp) by-product against the stored logits, then its two consumers against the kernels
    that stream the whole row: mic_row_topk_tiles == mic_row_lse_topk (indices identical, values to fp32 rounding — both are
    checked against the oracle's log_softmax + top_k) and mic_ce_rows_tiles == mic_ce_rows."""
    from mic_amd import ops
    from oracle import generation_ref as G

    K = 256
    g = torch.Generator().manual_seed(M + V)
    x = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    w = torch.zeros(Vpad, K, dtype=torch.bfloat16)
    w[:V] = (torch.randn(V, K, generator=g) * 0.2).to(torch.bfloat16)
    bias = torch.zeros(Vpad)
    bias[:V] = torch.randn(V, generator=g)
    bias[2] = 25.0  # EOS dominates: suppress_eos has to look past a tile maximum
    nt = Vpad // 64
    logits = torch.zeros((M, Vpad), dtype=torch.bfloat16, device=dev)
    stat = torch.full((M, nt, 2), float("nan"), device=dev)
    ops.gemm(x.to(dev), w.to(dev), logits, M, Vpad, K, bias=bias.to(dev), rowstat=stat, rowstat_nvalid=V)
    torch.cuda.synchronize()
    lg = logits.float().cpu()
    # partials: exactly the stored values' maximum, sum exp to fp32 rounding
    pad = torch.full((M, nt * 64), float("-inf"))
    pad[:, :V] = lg[:, :V]
    tiles = pad.reshape(M, nt, 64)
    mx = tiles.max(-1).values
    live = mx > float("-inf")  # granules entirely beyond V carry (-inf, 0)
    assert torch.equal(stat[:, :, 0].cpu()[live], mx[live]) and (stat[:, :, 0].cpu()[~live] == float("-inf")).all()
    sm = torch.exp(tiles - torch.where(live, mx, torch.zeros_like(mx))[..., None]).sum(-1)
    assert ((stat[:, :, 1].cpu() - sm).abs()[live] / sm[live]).max().item() < 2e-5 and (stat[:, :, 1].cpu()[~live] == 0).all()
    # top-k consumer vs the full-row kernel and the oracle
    R, k = min(M, 48), 8
    rb = torch.linspace(-3.0, 0.0, R)
    tv, ti = torch.empty((R, k), device=dev), torch.empty((R, k), dtype=torch.int32, device=dev)
    tv2, ti2 = torch.empty((R, k), device=dev), torch.empty((R, k), dtype=torch.int32, device=dev)
    for sup in (False, True):
        ops.row_topk_tiles(logits, Vpad, V, stat, k, tv, ti, R, suppress_eos=sup, eos_token_id=2, row_bias=rb.to(dev))
        ops.row_lse_topk(logits, Vpad, V, k, tv2, ti2, R, suppress_eos=sup, eos_token_id=2, row_bias=rb.to(dev))
        torch.cuda.synchronize()
        assert torch.equal(ti.cpu(), ti2.cpu()), sup
        assert (tv.cpu() - tv2.cpu()).abs().max().item() < 2e-5
        lp = G.log_softmax(lg[:R, :V].numpy())
        if sup:
            lp[:, 2] = -np.inf
        rv, ri = G.top_k(lp + rb.numpy()[:, None], k)
        assert np.array_equal(ti.cpu().numpy(), ri.astype(np.int32))
        assert np.allclose(tv.cpu().numpy(), rv, rtol=1e-5, atol=2e-5)
    # k = 16 (num_beams 8) and raw mode (no log-softmax)
    tv, ti = torch.empty((R, 16), device=dev), torch.empty((R, 16), dtype=torch.int32, device=dev)
    ops.row_topk_tiles(logits, Vpad, V, stat, 16, tv, ti, R, raw_logits=True)
    torch.cuda.synchronize()
    rv, ri = G.top_k(lg[:R, :V].numpy(), 16)
    assert np.array_equal(ti.cpu().numpy(), ri.astype(np.int32)) and np.array_equal(tv.cpu().numpy(), rv)
    # cross-entropy consumer
    labels = torch.randint(0, V, (M,), generator=g).to(torch.int32)
    ones = torch.ones(M, dtype=torch.int32)
    l1, r1 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    l2, r2 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.ce_rows_tiles(logits, Vpad, V, stat, labels.to(dev), l1, r1, M)
    ops.ce_rows(logits, Vpad, V, labels.to(dev), ones.to(dev), 0.0, l2, r2, M)
    torch.cuda.synchronize()
    assert (l1 - l2).abs().max().item() < 2e-5 and (r1 - r2).abs().max().item() < 2e-5
    ref = torch.logsumexp(lg[:, :V], -1)
    assert (l1.cpu() - ref).abs().max().item() < 2e-5


def test_row_topk_tiles_ties_across_many_tiles(dev):
    """All logits equal (every granule at the maximum): index-stable ordering and the > 256 candidate-granule fallback."""
    from mic_amd import ops

    R, V = 3, 100_000
    Vpad = (V + 63) // 64 * 64
    logits = torch.zeros((R, Vpad), dtype=torch.bfloat16, device=dev)
    logits[1, 77_777] = 1.0
    nt = Vpad // 64
    stat = torch.zeros((R, nt, 2), device=dev)
    stat[:, :, 1] = 64.0
    stat[:, nt - 1, 1] = V - (nt - 1) * 64
    stat[1, 77_777 // 64, 0] = 1.0
    stat[1, 77_777 // 64, 1] = 1.0 + 63.0 * float(np.exp(-1.0))
    tv, ti = torch.empty((R, 8), device=dev), torch.empty((R, 8), dtype=torch.int32, device=dev)
    ops.row_topk_tiles(logits, Vpad, V, stat, 8, tv, ti, R)
    torch.cuda.synchronize()
    assert ti[0].tolist() == list(range(8)) and ti[2].tolist() == list(range(8))
    assert ti[1].tolist() == [77_777] + list(range(7))


@pytest.mark.parametrize("M,N,K,act", [(1024, 1024, 1024, 0), (1000, 3072, 1024, 0), (4096, 1024, 1024, 0), (1024, 4096, 1024, 2),
                                        (2304, 12288, 256, 0), (200, 128, 128, 3)])
def test_gemm_layernorm_fold_and_rowsum2(dev, M, N, K, act):
    """LayerNorm folded around the GEMM (decode path): a producer GEMM stores x = a0 b0^T + r and accumulates (sum, sum of
    squares) of every stored row (rowsum2); the consumer runs on the RAW x with gamma folded into its weight
    (mic_ln_fold_weight) and finishes LN in its epilogue.  Reference: LayerNorm (fp32 math on the stored bf16 x, output rounded
    to bf16 like mic_layernorm_fwd) followed by the plain Linear.  The fold skips that rounding, so the two agree to bf16
    rounding of the activations, not bit for bit: 1.2e-2 of the output scale."""
    from mic_amd import ops
    from mic_amd import _lib as L

    g = torch.Generator().manual_seed(M + N + K)
    dt = torch.bfloat16
    a0, b0, r = rnd((M, 64), g, dt), rnd((K, 64), g, dt, 0.3), rnd((M, K), g, dt, 1.0)
    r += 0.7   # a common-mode offset: the epilogue's  acc - mu g  has something to cancel
    x = torch.zeros((M, K), dtype=dt, device=dev)
    st = torch.zeros((M, 2), dtype=torch.int64, device=dev)   # (sum, sum of squares) x 2^20
    ops.gemm(a0.to(dev), b0.to(dev), x, M, K, 64, residual=r.to(dev), rowsum2=st)
    torch.cuda.synchronize()
    xr = (a0.float() @ b0.float().T + r.float()).to(dt)
    assert relerr(x, xr) < tol(dt)
    xs = x.float().cpu()
    sf = st.cpu().double() / 2 ** 20
    assert relerr(sf[:, 0], xs.double().sum(1)) < 1e-5 and relerr(sf[:, 1], (xs.double() ** 2).sum(1)) < 1e-5
    st2 = torch.zeros_like(st)   # integer atomics: the same bits whatever order the column tiles finish in
    ops.gemm(a0.to(dev), b0.to(dev), x, M, K, 64, residual=r.to(dev), rowsum2=st2)
    torch.cuda.synchronize()
    assert torch.equal(st, st2)
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    w, bias = rnd((N, K), g, dt, 0.05), 0.1 * torch.randn(N, generator=g)
    wf = torch.empty((N, K), dtype=dt, device=dev)
    cs, bf = torch.empty(N, device=dev), torch.empty(N, device=dev)
    ops.ln_fold_weight(w.to(dev), gamma.to(dev), beta.to(dev), bias.to(dev), wf, cs, bf)
    torch.cuda.synchronize()
    wfr = (w.float() * gamma[None]).to(dt)
    assert torch.equal(wf.cpu(), wfr)
    assert relerr(cs, wfr.float().sum(1)) < 1e-5 and relerr(bf, bias + w.float() @ beta) < 1e-5
    out = torch.empty((M, N), dtype=dt, device=dev)
    eps = 1e-5
    ops.gemm(x, wf, out, M, N, K, bias=bf, act=act, ln_stats=st, ln_colsum=cs, ln_width=K, ln_eps=eps)
    torch.cuda.synchronize()
    ln = torch.nn.functional.layer_norm(xs, (K,), gamma, beta, eps).to(dt).float()
    ref = ln @ w.float().T + bias
    if act == 2:
        ref = torch.nn.functional.gelu(ref.to(dt).float(), approximate="tanh")
    elif act == 3:
        z = ref.to(dt).float()
        ref = z * torch.sigmoid(1.702 * z)
    assert relerr(out, ref) < 1.2e-2, relerr(out, ref)
    # grouped: two folded problems sharing A (the q / k of a decoder layer)
    o1, o2 = torch.empty((M, N // 2), dtype=dt, device=dev), torch.empty((M, N // 2), dtype=dt, device=dev)
    h = N // 2
    lk = dict(ln_stats=st, ln_width=K, ln_eps=eps)
    ops.gemm_grouped([ops.gemm_args(x, wf[:h], o1, M, h, K, bias=bf[:h], act=act, ln_colsum=cs[:h], **lk),
                      ops.gemm_args(x, wf[h:], o2, M, h, K, bias=bf[h:], act=act, ln_colsum=cs[h:], **lk)])
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([o1, o2], 1), out)
    # misuse
    with pytest.raises(L.MicError, match="rowsum2"):
        ops.gemm(a0.to(dev), b0.to(dev), x, M, K, 64, act=2, rowsum2=st)
    with pytest.raises(L.MicError, match="folded LayerNorm"):
        ops.gemm(x, wf, out, M, N, K, ln_stats=st, ln_colsum=cs, ln_width=K, ln_eps=eps)   # bias' missing



@pytest.mark.parametrize("M,N,K", [(2404, 4096, 1024), (3200, 3072, 768), (2049, 4096, 128), (3072, 4096, 64)])
def test_gemm_192_row_tiles(dev, M, N, K):
    """192 x 128 tiles (wave tile 96 x 32, three 64-row A images, three 32-row epilogue passes): taken for the single-problem NT / NN
    launches whose 128-row tiles would spill into a second round.  NT with bias + GELU + saved pre-activation (FFN-in forward), NN
    with act'(z) (the dGELU dX), NN bare with a residual; a last row tile that is partial (2404 = 12 x 192 + 100) or absent."""
    from mic_amd import _lib as L
    from mic_amd import ops

    for bkm in (False, True):
        assert ops.gemm_plan([(M, N, K)], b_kmajor=bkm)["tile_m"] == 192
    assert ops.gemm_plan([(M, N, K)], a_kmajor=True, b_kmajor=True)["tile_m"] != 192  # k-major A: not built
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    A, B = rnd((M, K), g, dt), rnd((N, K), g, dt, 0.1)
    bias = torch.randn(N, generator=g)
    out = torch.full((M + 64, N), float("nan"), dtype=dt, device=dev)  # rows behind M must stay untouched
    z = torch.full((M + 64, N), float("nan"), dtype=dt, device=dev)
    ops.gemm(A.to(dev), B.to(dev), out, M, N, K, bias=bias.to(dev), act=L.ACT_GELU_TANH, zout=z)
    zr = A.float() @ B.float().T + bias
    ref = torch.nn.functional.gelu(zr.to(dt).float(), approximate="tanh")
    torch.cuda.synchronize()
    assert relerr(z[:M], zr) < tol(dt) and relerr(out[:M], ref) < tol(dt)
    assert torch.isnan(out[M:].float()).all() and torch.isnan(z[M:].float()).all()
    # NN: dz = (dy @ W) * act'(z);  W stored [K][N] (k-major B)
    dY, W = rnd((M, K), g, dt), rnd((K, N), g, dt, 0.1)
    dz = torch.full((M + 64, N), float("nan"), dtype=dt, device=dev)
    ops.gemm(dY.to(dev), W.to(dev), dz, M, N, K, b_kmajor=True, zin=z, dact=L.ACT_GELU_TANH)
    zz = z[:M].float().cpu().requires_grad_(True)
    torch.nn.functional.gelu(zz, approximate="tanh").backward(torch.ones_like(zz))
    torch.cuda.synchronize()
    assert relerr(dz[:M], (dY.float() @ W.float()) * zz.grad) < tol(dt) and torch.isnan(dz[M:].float()).all()
    # NN, PLAIN epilogue with a residual
    R = rnd((M, N), g, dt)
    o2 = torch.full((M + 64, N), float("nan"), dtype=dt, device=dev)
    ops.gemm(dY.to(dev), W.to(dev), o2, M, N, K, b_kmajor=True, residual=R.to(dev))
    torch.cuda.synchronize()
    assert relerr(o2[:M], dY.float() @ W.float() + R.float()) < tol(dt) and torch.isnan(o2[M:].float()).all()



@pytest.mark.parametrize("M,N,K", [(2048, 16384, 192), (1000, 66048, 64), (1024, 131072, 1024), (3200, 24576, 1024), (600, 98304, 128)])
def test_gemm_phased_many_tiles(dev, M, N, K):
    """the LDS-DMA 256 x 256 kernel on single NT problems with several tiles per CU: odd and even K-tile counts, a single K-tile,
    partial row tiles, the bias / activation epilogues and the LM head's softmax partials; run-to-run identical bits.  (Written for
    the persistent variant with next-tile prefetch that round 4 measured slower and took out again; kept as coverage of the kernel.)"""
    from mic_amd import _lib as L
    from mic_amd import ops

    plan = ops.gemm_plan([(M, N, K)])
    w4 = os.environ.get("MIC_GEMM_W4", "1") != "0" and K >= 256 and K % 128 == 0  # (the four-wave kernel takes these shapes by default)
    assert plan["tile"] == 256 and plan["phased"] == (2 if w4 else 1) and plan["blocks"] > 256
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    A, B = rnd((M, K), g, dt), rnd((N, K), g, dt, 0.1)
    bias = torch.randn(N, generator=g)
    ref = A.float() @ B.float().T + bias
    out = torch.full((M + 8, N), float("nan"), dtype=dt, device=dev)
    ops.gemm(A.to(dev), B.to(dev), out, M, N, K, bias=bias.to(dev))
    torch.cuda.synchronize()
    assert relerr(out[:M], ref) < tol(dt) and torch.isnan(out[M:].float()).all()
    base = out[:M].clone()
    for _ in range(3):  # run to run: the same bits (no dependence on which tile a block reaches when)
        ops.gemm(A.to(dev), B.to(dev), out, M, N, K, bias=bias.to(dev))
        torch.cuda.synchronize()
        assert torch.equal(out[:M], base)
    # non-PLAIN epilogue: GELU + saved pre-activation
    z = torch.empty((M, N), dtype=dt, device=dev)
    o2 = torch.empty((M, N), dtype=dt, device=dev)
    ops.gemm(A.to(dev), B.to(dev), o2, M, N, K, bias=bias.to(dev), act=L.ACT_GELU_TANH, zout=z)
    torch.cuda.synchronize()
    assert relerr(z, ref) < tol(dt) and relerr(o2, torch.nn.functional.gelu(ref.to(dt).float(), approximate="tanh")) < tol(dt)
    # the LM head's by-product: (max, sum exp) per 64-column granule of the values as stored
    stat = torch.zeros((M, 2 * (N // 64)), dtype=torch.float32, device=dev)
    o3 = torch.empty((M, N), dtype=dt, device=dev)
    ops.gemm(A.to(dev), B.to(dev), o3, M, N, K, bias=bias.to(dev), rowstat=stat, rowstat_nvalid=N - 37)
    torch.cuda.synchronize()
    assert torch.equal(o3, base)
    x = o3.float().cpu().reshape(M, N // 64, 64).clone()
    x.reshape(M, N)[:, N - 37:] = float("-inf")
    mx = x.max(dim=2).values
    sm = torch.exp(x - mx[..., None]).sum(dim=2)
    st = stat.cpu().reshape(M, N // 64, 2)
    assert torch.equal(st[..., 0], mx) and torch.allclose(st[..., 1], sm, rtol=2e-5, atol=1e-6)


def test_gemm_layernorm_fold_with_softmax_partials(dev):
    """the decode-time LM head in one launch: LayerNorm folded around the GEMM AND the (max, sum exp) partials per 64-column granule
    — the logits equal the same launch without the partials bit for bit, the partials describe the values as stored"""
    from mic_amd import ops

    M, N, K = 600, 66048, 256
    assert ops.gemm_plan([(M, N, K)])["tile"] == 256
    g = torch.Generator().manual_seed(7)
    dt = torch.bfloat16
    x = (rnd((M, K), g, dt, 1.0) + 0.5).to(dev)
    xs = x.float().cpu().double()
    st = torch.stack([(xs.sum(1) * 2 ** 20).round(), ((xs ** 2).sum(1) * 2 ** 20).round()], 1).to(torch.int64).to(dev)
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    w, bias = rnd((N, K), g, dt, 0.05), 0.1 * torch.randn(N, generator=g)
    wf = torch.empty((N, K), dtype=dt, device=dev)
    cs, bf = torch.empty(N, device=dev), torch.empty(N, device=dev)
    ops.ln_fold_weight(w.to(dev), gamma.to(dev), beta.to(dev), bias.to(dev), wf, cs, bf)
    lk = dict(bias=bf, ln_stats=st, ln_colsum=cs, ln_width=K, ln_eps=1e-5)
    o1, o2 = torch.empty((M, N), dtype=dt, device=dev), torch.empty((M, N), dtype=dt, device=dev)
    stat = torch.zeros((M, 2 * (N // 64)), dtype=torch.float32, device=dev)
    ops.gemm(x, wf, o1, M, N, K, **lk)
    ops.gemm(x, wf, o2, M, N, K, rowstat=stat, rowstat_nvalid=N - 5, **lk)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.float().cpu(), (K,), gamma, beta, 1e-5).to(dt).float() @ w.float().T + bias
    assert relerr(o1, ref) < 1.2e-2 and torch.equal(o1, o2)
    v = o2.float().cpu().reshape(M, N // 64, 64).clone()
    v.reshape(M, N)[:, N - 5:] = float("-inf")
    mx = v.max(dim=2).values
    sm = torch.exp(v - mx[..., None]).sum(dim=2)
    got = stat.cpu().reshape(M, N // 64, 2)
    assert torch.equal(got[..., 0], mx) and torch.allclose(got[..., 1], sm, rtol=2e-5, atol=1e-6)


def test_gemm_four_phase_kernel_behind_its_switch(dev):
    """The single-problem NT 256 x 256 launches with a bare epilogue (bias, the folded LayerNorm, the LM head's softmax partials) run on
    gemm_w4.hip by default and on gemm_phased.hip under MIC_GEMM_W4=0.  The library reads the switch once, so the second configuration
    runs in a child process: the many-tiles cases with K = 1024 (bias, bit-identical reruns, partial row tiles, the softmax partials)
    and the folded-LayerNorm / softmax-partial tests of this file — the same tests this process runs on the four-wave kernel."""
    import subprocess
    import sys

    from mic_amd import ops

    assert ops.gemm_plan([(1024, 131072, 1024)])["phased"] == (2 if os.environ.get("MIC_GEMM_W4", "1") != "0" else 1)
    env = dict(os.environ, MIC_GEMM_W4="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "(phased_many_tiles and 1024) or head_rowstat or layernorm_fold"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,img,ps", [(3, 224, 32), (2, 64, 32), (2, 24, 4), (1, 48, 16)])
def test_im2col_matches_patch_gather(dev, dtype, B, img, ps):
    """patches[(b, pi, pj)][(u, v, c)] = pixels[b][pi ps + u][pj ps + v][c] (the ViT's patch embedding as a GEMM, modeling T1) — the
    8-elements-per-thread kernel (patch rows of ps * 3 = 96 floats) and the element-wise fallback (ps = 4), with and without the
    reference's int32 truncation of the pixels (modeling:330)"""
    from mic_amd import ops

    g = torch.Generator().manual_seed(B * img + ps)
    px = (torch.randn(B, img, img, 3, generator=g) * 3).to(dev)
    G, pk = img // ps, ps * ps * 3
    for trunc in (False, True):
        out = torch.zeros((B * G * G + 5, pk), dtype=dtype, device=dev)
        ops.im2col(px, out, B, img, ps, pk, trunc)
        torch.cuda.synchronize()
        src = px.trunc() if trunc else px
        ref = src.reshape(B, G, ps, G, ps, 3).permute(0, 1, 3, 2, 4, 5).reshape(B * G * G, pk).to(dtype)
        assert torch.equal(out[: B * G * G], ref) and float(out[B * G * G:].float().abs().max()) == 0.0


@pytest.mark.parametrize("n,width,V", [(3000, 256, 500), (32768, 1024, 250054), (13, 64, 5), (32767, 1024, 250054)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_rows_add_deterministic(dev, dtype, n, width, V):
    """the data-parallel embedding-row scatter: dtable[ids[i]] += scale * dh[i] with heavy duplicates (every sequence starts with the
    same few ids), ids < 0 skipped; equal to an fp64 index_add, and the SAME BITS from run to run (a fixed reduction tree: what
    keeps the replicas of a data-parallel job identical) — unlike the atomic scatter of the single-process path"""
    from mic_amd import ops

    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V, (n,), generator=g, dtype=torch.int32)
    if n == 32767:  # what 8 ranks x 4096 rows look like: sequences of 64 that start with (eos, language id), a third of the rows padding
        ids[::64] = 2
        ids[1::64] = 250004 + (torch.arange(len(ids[1::64])) % 4).to(torch.int32)
        pad = torch.rand(n, generator=g) < 0.33
        pad[::64] = pad[1::64] = False
        ids[pad] = -1
    else:
        ids[::5] = min(7, V - 1)   # n / 5 occurrences of one id
        ids[1::7] = -1             # skipped rows (their dh may hold anything)
    dh = rnd((n, width), g, dtype, 1.0)
    dh[ids < 0] = float("nan")
    touched = torch.unique(ids[ids >= 0]).long()
    base = torch.randn(len(touched), width, generator=g)   # (the 8 ranks x 4096 rows case touches ~20 k rows of a 250 054-row table)
    outs = []
    ids_d, dh_d = ids.to(dev), dh.to(dev)
    ws = ops.embed_rows_add_det_workspace(V, dev)
    ws0 = ws.clone()
    for rep in range(3):
        t = torch.zeros((V, width), device=dev)
        t[touched.to(dev)] = base.to(dev)
        ops.embed_rows_add_det(ids_d, dh_d, 0.5, t, n, width, ws)
        torch.cuda.synchronize()
        outs.append(t[touched.to(dev)].cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(ws[: 3 * V + 1], ws0[: 3 * V + 1])  # owner / count / last / list length are left as found (the list's entries are scratch)
    keep = ids >= 0
    pos = torch.searchsorted(touched, ids[keep].long())
    ref = base.double().index_add(0, pos, dh[keep].double() * 0.5)
    assert ((outs[0].double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
    if n > 30000:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.embed_rows_add_det(ids_d, dh_d, 0.5, t, n, width, ws)
        e1.record()
        torch.cuda.synchronize()
        print(f"[embed_rows_add_det] {n} rows x {width}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us per call (the data-parallel step's tail at 8 ranks)")


@pytest.mark.parametrize("d2", ["0", "2"])
def test_gemm_two_blocks_per_cu_kernel_behind_its_switch(dev, d2):
    """gemm_d2.hip takes, by default (MIC_GEMM_D2=3), only the launches with softmax partials.  Its general epilogue (activation, saved
    pre-activation, dGELU, dropout, residual, fp32 C, accumulate, split-K slabs) is reachable under MIC_GEMM_D2=2 (every single-problem
    NT 256-tile launch it covers), and with MIC_GEMM_D2=0 the softmax-partials launches fall back to the four-wave kernel's STATS
    instantiations: both configurations run the many-tile, softmax-partial, folded-LayerNorm and bit-for-bit tests of this file in a
    child process (the library latches the switch at its first GEMM)."""
    import subprocess
    import sys

    env = dict(os.environ, MIC_GEMM_D2=d2)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "(phased and not behind_its_switch) or head_rowstat or layernorm_fold or repeats_bit_for_bit"], env=env, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("M,N,K", [(1024, 131072, 256), (2300, 66048, 384)])
def test_gemm_four_wave_kernel_repeats_bit_for_bit(dev, M, N, K):
    """the four-wave kernel on many-tile launches with the LM head's epilogue: the same bits from run to run, also when two launches
    overlap on two streams and when the last row tile is partial (waves whose rows lie past M idle through the K loop)"""
    from mic_amd import ops

    assert ops.gemm_plan([(M, N, K)])["tile"] == 256 and ops.gemm_plan([(M, N, K)])["blocks"] > 256
    g = torch.Generator().manual_seed(M + K)
    a, b = rnd((M, K), g, torch.bfloat16).to(dev), rnd((N, K), g, torch.bfloat16, 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)

    def run(stream=None):
        c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        st = torch.zeros((M, 2 * (N // 64)), dtype=torch.float32, device=dev)
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())  # (the fills above run on the current stream)
        with torch.cuda.stream(stream or torch.cuda.current_stream()):
            ops.gemm(a, b, c, M, N, K, bias=bias, rowstat=st, rowstat_nvalid=N - 7)
        return c, st

    c0, s0 = run()
    torch.cuda.synchronize()
    assert relerr(c0, a.float().cpu() @ b.float().cpu().T + bias.cpu()) < tol(torch.bfloat16)
    side = torch.cuda.Stream()
    for it in range(12):
        if it % 3 == 2:
            cb, sb = run(side)
            c, st = run()   # overlaps the launch on the side stream
            torch.cuda.synchronize()
            assert torch.equal(cb, c0) and torch.equal(sb, s0)
        else:
            c, st = run()
            torch.cuda.synchronize()
        assert torch.equal(c, c0) and torch.equal(st, s0)


# ---------------------------------------------------------------- round 5: the LM head's backward as NT launches
@pytest.mark.parametrize("rows,cols,pad", [(64, 512, 0), (100, 1024, 128), (37, 72, 64), (2300, 1032, 2304)])
def test_transpose_bf16(dev, rows, cols, pad):
    """mic_transpose_bf16: dst[c][r] = src[r][c], bit for bit; columns rows .. rows_pad of dst zero, nothing written beyond"""
    from mic_amd import ops

    g = torch.Generator().manual_seed(rows + cols)
    src = rnd((rows + 3, cols + 8), g, torch.bfloat16).to(dev)   # (a view with a leading dimension larger than `cols`)
    rp = pad if pad else (rows + 63) // 64 * 64
    dst = torch.full((cols, rp + 8), 7.0, dtype=torch.bfloat16, device=dev)
    ops.transpose_bf16(src, dst, rows, cols, rows_pad=pad)
    torch.cuda.synchronize()
    assert torch.equal(dst[:, :rows], src[:rows, :cols].T)
    assert (dst[:, rows:rp] == 0).all() and (dst[:, rp:] == 7.0).all()


@pytest.mark.parametrize("ls", [0.0, 0.1])
@pytest.mark.parametrize("rows,V,Vpad", [(24, 1003, 1024), (150, 5003, 5056), (70, 600, 640)])
def test_cross_entropy_backward_with_transposed_copy(dev, ls, rows, V, Vpad):
    """mic_ce_bwd_t = mic_ce_bwd in place (same bits), plus dlogits^T (zero reduction padding) and the column sums of the stored
    gradient added to `colsum` (the final_logits_bias gradient)"""
    from mic_amd import ops

    g = torch.Generator().manual_seed(rows + V)
    logits = torch.zeros((rows + 5, Vpad), dtype=torch.bfloat16)
    logits[:rows, :V] = rnd((rows, V), g, torch.bfloat16, 3.0)
    labels = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32).to(dev)
    mask = (torch.rand(rows, generator=g) > 0.3).to(torch.int32).to(dev)
    a, b = logits.to(dev), logits.to(dev)
    lse, rl = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    loss, denom = torch.empty(1, device=dev), torch.empty(1, device=dev)
    ops.ce_rows(a, Vpad, V, labels, mask, ls, lse, rl, rows)
    ops.ce_reduce(rl, mask, loss, denom, rows)
    ops.ce_bwd(a, Vpad, V, Vpad, labels, mask, ls, lse, denom, rows)
    rp = (rows + 127) // 128 * 128
    dT = torch.full((Vpad, rp + 64), 3.0, dtype=torch.bfloat16, device=dev)
    cs = torch.full((Vpad,), 0.5, dtype=torch.float32, device=dev)
    ops.ce_bwd_t(b, Vpad, V, Vpad, labels, mask, ls, lse, denom, rows, dT, rows_pad=rp, colsum=cs)
    torch.cuda.synchronize()
    assert torch.equal(a, b)                                   # the in-place gradient: the same bits as mic_ce_bwd
    assert torch.equal(dT[:, :rows], b[:rows].T)               # ... and transposed
    assert (dT[:, rows:rp] == 0).all() and (dT[:, rp:] == 3.0).all()
    ref = b[:rows].float().sum(0).cpu()
    assert (cs.cpu() - 0.5 - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()) + 2e-7
    assert (b[rows:] == logits[rows:].to(dev)).all()           # rows behind `rows` untouched


@pytest.mark.parametrize("M,N,K,nsp", [(4096, 4096, 256, 1), (2300, 6000, 384, 1), (2432, 1024, 64 * 96, 6), (2176, 1024, 64 * 50, 7),
                                      (520, 1024, 64 * 128, 24)])
def test_gemm_four_wave_kernel_fp32_and_split_k_slabs(dev, M, N, K, nsp):
    """the four-wave kernel's fp32 epilogue: C32 = A B^T (NT) straight into fp32, and as split-K slabs (one per split, fully
    overwritten, empty K ranges as zeros) summed by mic_sum_slabs — the LM head's dE and dX launches; bit-identical run to run"""
    from mic_amd import ops

    plan = ops.gemm_plan([(M, N, K)], split_k=nsp)
    assert plan["tile"] == 256 and plan["phased"] == 2, plan
    g = torch.Generator().manual_seed(M + N + K)
    a = rnd((M, K), g, torch.bfloat16, 0.5).to(dev)
    b = rnd((N, K), g, torch.bfloat16, 0.1).to(dev)
    ref = a.float() @ b.float().T
    Mp = (M + 63) // 64 * 64
    slab = Mp * N
    outs = []
    for _ in range(2):
        ws = torch.full((max(nsp, 1) * Mp, N), float("nan"), dtype=torch.float32, device=dev)
        ops.gemm(a, b, ws, M, N, K, split_k=nsp if nsp > 1 else 0, split_stride=slab if nsp > 1 else 0)
        if nsp > 1:
            o = torch.empty((M, N), dtype=torch.float32, device=dev)
            ops.sum_slabs(ws, nsp, slab, o, M, N, ws.stride(0), o.stride(0))
        else:
            o = ws[:M]
        torch.cuda.synchronize()
        assert torch.isfinite(o).all()
        assert (o - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
        outs.append(o.clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,width", [(2432, 1024), (3200, 768), (37, 72), (5, 2048)])
def test_layernorm_bwd_partials_and_param_grads(dev, dtype, rows, width):
    """mic_layernorm_bwd_partials + mic_ln_param_grads = mic_layernorm_bwd: the same dx / dxm bits, gamma / beta gradients equal to
    the atomics' to fp32 summation order — and the same bits from run to run (block partials added in block order), overwrite and
    accumulate modes, several LayerNorms in one grouped launch"""
    from mic_amd import ops

    g = torch.Generator().manual_seed(rows * 3 + width)
    x, dy, dres = (rnd((rows, width), g, dtype, 2.0).to(dev) for _ in range(3))
    gamma = (1 + 0.1 * torch.randn(width, generator=g)).to(dev)
    mean, rstd = x.float().mean(1), (x.float().var(1, unbiased=False) + 1e-5).rsqrt()
    kw = dict(dres=dres, dropout_p=0.1, dropout_seed=9)
    dx0, dxm0 = torch.empty_like(x), torch.empty_like(x)
    dg0, db0 = torch.zeros(width, device=dev), torch.zeros(width, device=dev)
    ops.layernorm_bwd(x, gamma, mean, rstd, dy, dx0, dg0, db0, dxm=dxm0, **kw)
    nblk = ops.layernorm_bwd_blocks(rows)
    assert 1 <= nblk <= 512
    outs = []
    for rep in range(2):
        part = torch.full((2 * nblk, width), float("nan"), device=dev)  # fully overwritten
        dx1, dxm1 = torch.empty_like(x), torch.empty_like(x)
        ops.layernorm_bwd_partials(x, gamma, mean, rstd, dy, dx1, part, dxm=dxm1, **kw)
        dg1, db1 = torch.full((width,), 7.0, device=dev), torch.full((width,), 7.0, device=dev)
        dg2, db2 = torch.full((width,), 1.0, device=dev), torch.full((width,), 1.0, device=dev)
        ops.ln_param_grads([(part, nblk, width, dg1, db1, False), (part, nblk, width, dg2, None, True), (part, nblk, width, None, db2, True)])
        torch.cuda.synchronize()
        assert torch.equal(dx1, dx0) and torch.equal(dxm1, dxm0)
        sc = max(dg0.abs().max().item(), 1e-6)
        assert (dg1 - dg0).abs().max().item() < 1e-5 * sc + 1e-6 and (db1 - db0).abs().max().item() < 1e-5 * max(db0.abs().max().item(), 1e-6) + 1e-6
        assert torch.equal(dg2, dg1 + 1.0) or (dg2 - dg1 - 1.0).abs().max().item() < 1e-6 * sc + 1e-6  # accumulate
        assert (db2 - db1 - 1.0).abs().max().item() < 1e-6 * max(db1.abs().max().item(), 1.0)
        outs.append((dg1.clone(), db1.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])  # deterministic
