"""CPU tests of the product's host logic (no GPU compute): C-ABI exports, config/param-tree plumbing, checkpoint wire
format, schedule/shift helpers, bucket planning, and the N>1 gradient reducer on gloo (world_size 2)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    import mic_amd  # noqa: F401
    from mic_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "mic_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(mic_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    l = _lib.lib()  # loads without a GPU
    for name in declared:
        assert hasattr(l, name), f"{name} declared in include/mic_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert l.mic_version() == 1


def test_header_lists_every_environment_switch_the_library_latches():
    """include/mic_hip.h promises that the only state libmic_hip.so keeps is host-side planning state and NAMES the environment switches
    it latches: the list must be the set of getenv() names in csrc/"""
    csrc = os.path.join(ROOT, "multilingual-image-captioning_amd", "csrc")
    names = set()
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".inc")):
            names |= set(re.findall(r'getenv\("(MIC_[A-Z0-9_]+)"\)', open(os.path.join(csrc, f)).read()))
    hdr = open(os.path.join(ROOT, "include", "mic_hip.h")).read()
    top = hdr[: hdr.index("#ifndef MIC_HIP_H")]
    listed = set(re.findall(r"MIC_[A-Z0-9_]+", top)) - {"MIC_BF16", "MIC_F32", "MIC_E", "MIC_HIP_H"}
    assert names and names == listed, (sorted(names - listed), sorted(listed - names))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "multilingual-image-captioning_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_ops_fail_loudly_without_device_tensors():
    import mic_amd  # noqa: F401
    from mic_amd import _lib, ops

    a = torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises(_lib.MicError):
        ops.gemm(a, a, a.clone(), 64, 64, 64)


def _small_cfg():
    from mic_amd import CLIPVisionMBartConfig

    return CLIPVisionMBartConfig(mbart_config=dict(vocab_size=1003, d_model=128, decoder_layers=2, decoder_attention_heads=2, decoder_ffn_dim=256,
                                                   max_position_embeddings=64),
                                 clip_vision_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                                         image_size=48, patch_size=16))


def test_config_mirror():
    from mic_amd import CLIPVisionMBartConfig

    with pytest.raises(ValueError):
        CLIPVisionMBartConfig(mbart_config={})
    c = _small_cfg()
    assert c.is_encoder_decoder and c.mbart_config.hidden_size == 128 and c.clip_vision_config.image_size == 48
    assert c.mbart_config.pad_token_id == 1 and c.mbart_config.decoder_start_token_id == 2 and c.mbart_config.num_beams == 5
    d = c.to_dict()
    assert d["model_type"] == "clip-vision-mbart" and d["mbart_config"]["vocab_size"] == 1003
    c2 = CLIPVisionMBartConfig.from_clip_vision_mbart_configs(c.clip_vision_config, c.mbart_config)
    assert c2.to_dict() == d


def test_param_store_flax_tree_roundtrip_cpu():
    """Flax pytree (Dense [in,out], HWIO conv, separate q/k/v) <-> fused/transposed device layout, both directions."""
    from mic_amd.params import ParamStore, flatten_tree, unflatten_tree
    from oracle import model_ref as M

    from util_small import ref_config

    rc = ref_config()
    st = ParamStore(_small_cfg(), torch.float32, "cpu")
    assert {k: tuple(v) for k, v in st.flax_shapes().items()} == {k: tuple(v) for k, v in M.param_shapes(rc).items()}
    p = M.init_params(rc, seed=5, perturb_ln=True)
    st.load_flat({k: v.numpy() for k, v in p.items()})
    out = st.export_flat("master")
    for k, v in p.items():
        assert np.array_equal(out[k], v.numpy()), k
    # fused qkv rows are [q; k; v] of the transposed kernels; shared embedding zero-padded to a multiple of 128 rows
    L0 = "model/decoder/layers/0/self_attn/"
    w = st.f32("dec0.qkv.w")
    assert torch.equal(w[:128], p[L0 + "q_proj/kernel"].T) and torch.equal(w[256:], p[L0 + "v_proj/kernel"].T)
    assert st.Vpad == 1024 and torch.count_nonzero(st.f32("shared")[1003:]) == 0
    assert st.numel % 256 == 0 and all(s.offset % 64 == 0 for s in st.segs.values())
    tree = unflatten_tree(out)
    assert set(flatten_tree(tree)) == set(out)


def test_checkpoint_msgpack_roundtrip(tmp_path):
    from mic_amd.checkpoint import load_flax_msgpack, save_flax_msgpack

    tree = {"model": {"a": {"kernel": np.arange(12, dtype=np.float32).reshape(3, 4)}, "b": np.array([1.5, -2.0], dtype=np.float32)},
            "final_logits_bias": np.zeros((1, 5), dtype=np.float32)}
    path = str(tmp_path / "flax_model.msgpack")
    save_flax_msgpack(path, tree)
    back = load_flax_msgpack(path)
    assert np.array_equal(back["model"]["a"]["kernel"], tree["model"]["a"]["kernel"]) and back["model"]["b"].dtype == np.float32
    # wire format = flax.serialization: ext type 1 holding (shape, dtype name, bytes)
    import msgpack

    raw = msgpack.unpackb(open(path, "rb").read(), raw=False, strict_map_key=False)
    ext = raw["final_logits_bias"]
    assert isinstance(ext, msgpack.ExtType) and ext.code == 1
    shape, dtype_name, buf = msgpack.unpackb(ext.data, raw=False)
    assert shape == [1, 5] and dtype_name == "float32" and len(buf) == 20


def test_checkpoint_flax_wire_format_known_answer():
    """byte-level known answer, hand-assembled from the msgpack specification and flax.serialization's published rules (NOT a
    round trip through this build's writer): a map with an ndarray (ext 1), a numpy scalar (ext 3), a Python complex (ext 2), a
    bfloat16 leaf and a chunked leaf (`__msgpack_chunked_array__`), read by `from_bytes`; then the writer reproduces the bytes of
    the part it can produce."""
    import struct

    from mic_amd.checkpoint import from_bytes, to_bytes

    def fixstr(t):
        assert len(t) < 32
        return bytes([0xA0 | len(t)]) + t.encode()

    def nd_payload(shape, dtype_name, raw):  # fixarray(3): [fixarray(shape) of positive fixints, fixstr dtype, bin8 data]
        assert len(raw) < 256 and all(0 <= x < 128 for x in shape)
        return bytes([0x93, 0x90 | len(shape)]) + bytes(shape) + fixstr(dtype_name) + bytes([0xC4, len(raw)]) + raw

    def ext8(code, payload):  # ext 8: 0xc7, length, type, data
        assert len(payload) < 256 and len(payload) not in (1, 2, 4, 8, 16)
        return bytes([0xC7, len(payload), code]) + payload

    a_raw = struct.pack("<4h", 1, 2, 3, 4)
    kernel = ext8(1, nd_payload([2, 2], "int16", a_raw))
    scal_p = nd_payload([], "float32", struct.pack("<f", 1.5))
    assert len(scal_p) == 16
    scalar = bytes([0xD8, 3]) + scal_p                                    # fixext 16, type 3 (numpy scalar)
    cplx_p = bytes([0x92, 0xCB]) + struct.pack(">d", 1.0) + bytes([0xCB]) + struct.pack(">d", -2.0)  # [float64 1.0, float64 -2.0]
    cplx = ext8(2, cplx_p)
    bf = ext8(1, nd_payload([3], "bfloat16", struct.pack("<3H", 0x3F80, 0xC000, 0x3DCD)))  # 1.0, -2.0, 0.10009765625
    ch0 = ext8(1, nd_payload([3], "float32", struct.pack("<3f", 0.0, 1.0, 2.0)))
    ch1 = ext8(1, nd_payload([3], "float32", struct.pack("<3f", 3.0, 4.0, 5.0)))
    chunked = (bytes([0x83]) + fixstr("__msgpack_chunked_array__") + bytes([0xC3])            # fixmap(3): flag true
               + fixstr("shape") + bytes([0x82]) + fixstr("0") + bytes([2]) + fixstr("1") + bytes([3])
               + fixstr("chunks") + bytes([0x82]) + fixstr("0") + ch0 + fixstr("1") + ch1)
    blob = (bytes([0x85]) + fixstr("kernel") + kernel + fixstr("count") + scalar + fixstr("z") + cplx + fixstr("h") + bf
            + fixstr("big") + chunked)
    t = from_bytes(blob)
    assert t["kernel"].dtype == np.int16 and np.array_equal(t["kernel"], [[1, 2], [3, 4]])
    assert isinstance(t["count"], np.float32) and t["count"] == np.float32(1.5)  # unwrapped with ar[()], not a 0-d array
    assert t["z"] == complex(1.0, -2.0)
    assert t["h"].dtype == np.float32 and np.array_equal(t["h"], np.array([1.0, -2.0, 0.10009765625], dtype=np.float32))
    assert t["big"].shape == (2, 3) and np.array_equal(t["big"], np.arange(6, dtype=np.float32).reshape(2, 3))
    # the writer: same bytes for the leaves numpy can hold (no bfloat16), chunking at a 12-byte limit like flax's at 2^30
    tree = {"kernel": np.array([[1, 2], [3, 4]], dtype=np.int16), "count": np.float32(1.5), "z": complex(1.0, -2.0),
            "big": np.arange(6, dtype=np.float32).reshape(2, 3)}
    want = (bytes([0x84]) + fixstr("kernel") + kernel + fixstr("count") + scalar + fixstr("z") + cplx + fixstr("big") + chunked)
    assert to_bytes(tree, max_chunk_bytes=12) == want
    # under the limit a leaf stays one ext-1 record; a 0-d ndarray (optax `count` after device_get) is ext 1 with an empty shape
    one = to_bytes({"big": tree["big"], "c": np.asarray(7, dtype=np.int32)})
    assert one == bytes([0x82]) + fixstr("big") + ext8(1, nd_payload([2, 3], "float32", struct.pack("<6f", *range(6)))) + fixstr("c") + \
        ext8(1, nd_payload([], "int32", struct.pack("<i", 7)))
    back = from_bytes(one)
    assert back["c"].shape == () and int(back["c"]) == 7


def test_host_helpers_equal_oracle():
    from mic_amd import create_learning_rate_fn, shift_tokens_right
    from oracle import train_ref

    ids = np.array([[250004, 5, 6, 2, 1, 1], [250008, 9, 2, 1, 1, 1]])
    assert np.array_equal(shift_tokens_right(ids, 1), train_ref.shift_tokens_right(ids, 1))
    f = create_learning_rate_fn(train_ds_size=6400, train_batch_size=64, num_train_epochs=7, num_warmup_steps=100, learning_rate=5e-5)
    for s in (0, 1, 50, 100, 101, 400, 699, 700, 900):
        assert abs(f(s) - train_ref.linear_warmup_decay(s, 5e-5, 100, 700)) < 1e-15
    g = create_learning_rate_fn(train_ds_size=64, train_batch_size=64, num_train_epochs=3, num_warmup_steps=5, learning_rate=1e-3)
    for s in (0, 2, 5, 9):  # 3 total steps < 5 warmup steps: no decay phase -> constant lr after warmup (optax)
        assert g(s) == train_ref.linear_warmup_decay(s, 1e-3, 5, 3)
    assert g(9) == 1e-3


def test_plan_buckets():
    from mic_amd.train import plan_buckets

    b = plan_buckets(1000, 300, [0, 100, 250, 400, 420, 800, 990])
    assert b[0][0] == 0 and b[-1][1] == 1000 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert all(e - s >= 300 for s, e in b[:-1])
    assert plan_buckets(10, 100, [0, 5]) == [(0, 10)]
    # a slice above max_elems is cut further, inside the segments declared splittable only, at multiples of their row width
    g = plan_buckets(100_000, 20_000, [0, 500, 60_500, 80_000], max_elems=16_000, splittable=((500, 60_500, 1000),))
    assert g == [(0, 16_500), (16_500, 32_500), (32_500, 48_500), (48_500, 60_500), (60_500, 100_000)]
    assert plan_buckets(100_000, 20_000, [0, 500, 60_500, 80_000], max_elems=16_000) == plan_buckets(100_000, 20_000, [0, 500, 60_500, 80_000])


def test_plan_buckets_invariants_random():
    """whatever the layout: the buckets tile [0, numel) in order, every cut sits on a segment boundary or (inside a splittable segment)
    on one of its row boundaries, no bucket but the last is below the minimum, and no piece of a splittable segment exceeds the cap by
    more than the segment's head that rides with its first piece"""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from mic_amd.train import plan_buckets

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.integers(1, 5000), min_size=1, max_size=40), st.integers(1, 4000), st.integers(1, 6000), st.integers(0, 39), st.integers(1, 64))
    def check(sizes, min_elems, max_elems, big, align):
        sizes = list(sizes)
        big %= len(sizes)
        sizes[big] = sizes[big] * align  # the splittable segment is a whole number of rows
        bounds = [0]
        for x in sizes:
            bounds.append(bounds[-1] + x)
        numel = bounds[-1]
        seg = (bounds[big], bounds[big + 1], align)
        b = plan_buckets(numel, min_elems, bounds[:-1], max_elems=max_elems, splittable=(seg,))
        assert b[0][0] == 0 and b[-1][1] == numel and all(x[1] == y[0] and x[1] > x[0] for x, y in zip(b, b[1:]))
        base = plan_buckets(numel, min_elems, bounds[:-1])
        assert all(e - s >= min_elems for s, e in base[:-1])
        cuts, base_cuts = {e for _, e in b[:-1]}, {e for _, e in base[:-1]}
        assert base_cuts <= cuts and base_cuts <= set(bounds)
        step = max(align, (max_elems // align) * align)
        for c in cuts - base_cuts:   # the extra cuts: inside the splittable segment, on its row grid, `step` apart
            assert seg[0] < c < seg[1] and (c - seg[0]) % step == 0
        for s_, e in b:
            if s_ >= seg[0] and e <= seg[1] and e - s_ > step:
                raise AssertionError((s_, e, step))

    check()


def test_gemm_planner_invariants_random():
    """for any single bf16 NT / NN problem and any CU budget: the plan's blocks are the tile count of its tile shape, two / four
    K-groups are only chosen for launches that fit one round, a persistent grid never exceeds the budget"""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from mic_amd import ops

    @settings(max_examples=300, deadline=None)
    @given(st.integers(1, 6000), st.integers(1, 40), st.integers(1, 64), st.sampled_from([0, 256, 224, 192, 160, 96]), st.booleans())
    def check(M, n8, k64, cus, bkm):
        N, K = n8 * 128, k64 * 64
        ops.set_cu_budget(cus)
        p = ops.gemm_plan([(M, N, K)], b_kmajor=bkm)
        c = p["cu_budget"]
        assert c == (cus or 256) and p["tile"] in (64, 128, 256) and p["tile_m"] in (p["tile"], 192)
        assert p["blocks"] == -(-M // p["tile_m"]) * -(-N // p["tile"])
        if p["kgroups"] > 1:
            assert p["blocks"] <= c * p["blocks_per_cu"] and p["tile"] != 256
        assert p["grid"] <= max(p["blocks"], 1) and (p["grid"] == p["blocks"] or p["grid"] == c)
        if p["tile_m"] == 192:
            assert p["blocks"] <= 2 * c < -(-M // 128) * -(-N // 128)

    try:
        check()
    finally:
        ops.set_cu_budget(0)


def test_packed_rows_helper():
    from mic_amd import loss_rows, packed_rows

    mask = np.array([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [1, 0, 0, 0, 0]])
    dec_in = np.arange(15).reshape(3, 5) + 100
    q_off, q_len, ids, pos = packed_rows(mask, dec_in)
    assert q_off.tolist() == [0, 3, 8] and q_len.tolist() == [3, 5, 1] and ids.shape == (15,) and ids.dtype == np.int32
    assert ids[:9].tolist() == [100, 101, 102, 105, 106, 107, 108, 109, 110] and ids[9:].tolist() == [0] * 6
    assert pos[:9].tolist() == [0, 1, 2, 0, 1, 2, 3, 4, 0]
    idx, _ = loss_rows(mask, dec_in)
    assert np.array_equal(dec_in.reshape(-1)[idx], ids[: len(idx)])      # same order as the compacted LM head's rows
    assert packed_rows(np.array([[1, 0, 1]]), np.zeros((1, 3))) is None  # not a prefix of ones: the padded path is used


def test_bucket_plan_full_size_world8():
    """configs[2] (8 ranks, 547 M parameters): the exchange buckets of the real layout, computed without allocating it.  They
    follow backward: the first bucket is the dense LM-head half of the tied embedding (+ the logits bias) — complete right
    after the head's weight-gradient GEMM, 1 GB in fp32, on the wire for the whole of backward — then the decoder layers top
    down, the projection, the ViT layers top down; the region accumulated by atomics (biases, LayerNorm, position tables) is last."""
    from mic_amd import CLIPVisionMBartConfig
    from mic_amd.params import ParamStore
    from mic_amd.train import bucket_plan, describe_buckets

    st = ParamStore(CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={}), torch.bfloat16, "cpu", allocate=False)
    assert st.numel == 547_223_040 and st.master is None
    b = bucket_plan(st, 64.0)
    d = describe_buckets(st, b)
    assert b[0][0] == 0 and b[-1][1] == st.numel and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    # >= 64 MB each, except the TAIL of the plan: what backward completes last is what the next forward pass reads first, and its
    # exchange + optimizer pass have nothing left to hide under — the atomically accumulated region (5.8 MB), in front of it the
    # patch embedding alone (9.4 MB), in front of that <= 36 MB (the ViT's first layers)
    assert all(x["elements"] * 4 >= 64 * 1024 * 1024 for x in d[:-4])
    assert d[-2]["first"] == d[-2]["last"] == "patch.w" and d[-3]["MB"] <= 38 and d[-1]["MB"] < 12
    assert bucket_plan(st, 64.0, tail_mb=0.0)[-2:] == [(b[-4][0], b[-2][1]), b[-1]]  # without the tail cuts: rounds 1-5's plan
    # the first buckets = flb + the tied embedding in pieces of <= 128 MB (whole rows): the first gradients backward completes —
    # all pieces at once (one weight-gradient GEMM), but piece k's optimizer pass can start behind piece k's all-reduce
    sh = st.segs["shared"]
    pieces = [x for x in d if x["last"] == "shared"]
    assert d[0]["first"] == "flb" and len(pieces) == 8 and pieces == d[: len(pieces)]
    assert all(x["MB"] <= 128 * 1.048576 + 1.1 for x in pieces) and b[len(pieces) - 1][1] == sh.offset + sh.numel
    assert all((e - sh.offset) % st.d == 0 for (_, e) in b[: len(pieces)])
    assert 1.0e3 < sum(x["MB"] for x in pieces) < 1.1e3
    whole = bucket_plan(st, 64.0, max_mb=1e9)  # without the cap: one 1 GB bucket, as in rounds 1-3
    assert len(whole) == len(b) - 7 and whole[1:] == b[8:]
    # then decoder layers 11 -> 0 (14.7 M elements each without the cross k/v projections: a bucket spans a little more than
    # one layer), the cross k/v projections of all layers (complete at the end of decoder backward), projection + ViT; the
    # region accumulated by atomics last
    firsts = [x["first"] for x in d]
    dec = [f for f in firsts if f.startswith("dec") and f.endswith(".w") and ".ckv." not in f]
    layer = lambda n: int(n[3:].split(".")[0])
    ls = [layer(f) for f in dec]
    assert ls == sorted(ls, reverse=True) and ls[0] >= 10 and ls[-1] <= 1 and 8 <= len(dec) <= 12, ls
    segs = sorted(st.segs.values(), key=lambda s: s.offset)
    names = [s.name for s in segs]
    assert names.index("dec0.qkv.w") < names.index("dec0.ckv.w") < names.index("dec11.ckv.w") < names.index("vp.w")
    assert st.segs["dec11.ckv.w"].offset - st.segs["dec0.ckv.w"].offset == 11 * 2 * 1024 * 1024  # contiguous: one [L*2d][d] matrix
    assert d[-1]["last"] == "vit.cls"
    vit_first = next(i for i, f in enumerate(firsts) if f.startswith("vit") or f == "vp.w")
    assert all(not f.startswith("dec") or not f.endswith(".w") for f in firsts[vit_first:])
    assert 21 <= len(b) <= 31, len(b)


def test_comm_dtype_follows_the_byte_counts():
    """`grad_comm_dtype="auto"`: fp32 (the reference's pmean, main.py:698) where the projected ring all-reduce of the 2.19 GB of
    fp32 gradients hides under ~13 ms of backward, bf16 where it would not"""
    from mic_amd.train import allreduce_ms, choose_comm_dtype

    n = 547_223_040
    assert [round(allreduce_ms(4.0 * n, w), 1) for w in (1, 2, 4, 8)] == [0.0, 34.2, 17.1, 8.6]
    assert choose_comm_dtype(1, n) is None and choose_comm_dtype(8, n) is None
    assert choose_comm_dtype(2, n) is torch.bfloat16 and choose_comm_dtype(4, n) is torch.bfloat16
    assert choose_comm_dtype(2, 1_000_000) is None  # a small model's exchange hides at any world size


def test_exchange_defaults_are_the_references_and_the_cu_budget_is_enforced_not_assumed(monkeypatch):
    """Trainer defaults: fp32 gradient exchange (`lax.pmean` of fp32 gradients, main.py:698; bf16 / "auto" are opt-in), and the CUs
    the planner leaves to the collectives = the channel cap exported to RCCL (`configure_rccl` -> NCCL_MAX_NCHANNELS), 0 without one."""
    import inspect

    from mic_amd import Trainer
    from mic_amd.train import COMM_CUS_DEFAULT, configure_rccl, rccl_channel_cap

    sig = inspect.signature(Trainer.__init__).parameters
    assert sig["grad_comm_dtype"].default is None and sig["comm_cus"].default is None
    for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS"):
        monkeypatch.delenv(k, raising=False)
    assert rccl_channel_cap() == 0                      # nothing exported: no cap, the planner keeps all CUs
    assert configure_rccl() == COMM_CUS_DEFAULT == 32 and os.environ["NCCL_MAX_NCHANNELS"] == "32" and rccl_channel_cap() == 32
    assert configure_rccl(16) == 32                     # an exported value (here: ours, in general the user's) wins
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "24")
    monkeypatch.setenv("NCCL_MIN_NCHANNELS", "40")
    assert configure_rccl() == 24 and os.environ["NCCL_MIN_NCHANNELS"] == "24"  # min may not exceed the cap
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "junk")
    assert rccl_channel_cap() == 0


def test_bench_reads_rccl_channel_count_from_its_init_log(tmp_path):
    import bench

    p = tmp_path / "rccl.log"
    p.write_text("host:1:1 [0] NCCL INFO Channel 00/32 :    0   1\nhost:1:1 [0] NCCL INFO Channel 31/32 :    0   1\n"
                 "host:1:1 [0] NCCL INFO 32 coll channels, 32 collnet channels, 0 nvls channels, 32 p2p channels, 2 p2p channels per peer\n")
    assert bench.rccl_channels_from_log(str(p)) == 32
    p.write_text("host:1:1 [0] NCCL INFO Channel 00/16 :    0   1\n")
    assert bench.rccl_channels_from_log(str(p)) == 16
    p.write_text("nothing here\n")
    assert bench.rccl_channels_from_log(str(p)) is None and bench.rccl_channels_from_log(str(tmp_path / "missing")) is None


def test_gemm_planner_follows_the_cu_budget():
    """The tile planner under a reduced CU budget (collectives hold CUs beside backward): a launch that was ONE round of blocks on
    256 CUs stays one round on 224 / 192 — it re-plans (fewer K-groups = more blocks per CU, or smaller tiles) instead of spilling
    a few blocks into a second round.  Host arithmetic of libmic_hip.so (mic_gemm_plan): no GPU needed."""
    from mic_amd import ops

    step = {  # the one-round launches of the batch-64 train step and of the decoder step (M, N, K)
        "dec so/cq/co dense": [(4096, 1024, 1024)], "dec so/cq/co packed": [(2404, 1024, 1024)], "dec fc2 packed": [(2404, 1024, 4096)],
        "vit out": [(3200, 768, 768)], "vit fc2": [(3200, 768, 3072)], "gen d x d": [(1024, 1024, 1024)], "gen fc2": [(1024, 1024, 4096)],
        "gen qkv": [(1024, 1024, 1024)] * 3, "gen fc1": [(1024, 4096, 1024)],
    }
    try:
        ops.set_cu_budget(0)
        assert ops.get_cu_budget() == 256
        base = {k: ops.gemm_plan(v) for k, v in step.items()}
        assert all(p["blocks"] <= 256 * p["blocks_per_cu"] for p in base.values()), base  # one round today
        assert base["dec so/cq/co dense"]["kgroups"] == 2 and base["gen d x d"] == dict(tile=64, tile_m=64, kgroups=4, blocks=256, grid=256, blocks_per_cu=1, phased=0, cu_budget=256)
        for cus in (224, 192):
            ops.set_cu_budget(cus)
            for k, v in step.items():
                p = ops.gemm_plan(v)
                assert p["cu_budget"] == cus and p["blocks"] <= cus * p["blocks_per_cu"], (cus, k, p)
            # the dense 256-tile projection gives up its second K-group (two 8-wave blocks per CU instead of one 16-wave block)
            assert ops.gemm_plan(step["dec so/cq/co dense"])["kgroups"] == 1 and ops.gemm_plan(step["gen d x d"])["kgroups"] == 2
            # 256 tiles of 256 x 256 would be two rounds on fewer than 256 CUs: quarter tiles instead
            assert ops.gemm_plan([(4096, 4096, 1024)])["tile"] == 128
            # persistent grids shrink with the budget
            head = ops.gemm_plan([(3072, 1024, 4096), (1024, 1024, 4096), (1024, 1024, 4096), (1024, 1024, 4096), (4096, 1024, 4096), (1024, 4096, 4096)],
                                 a_kmajor=True, b_kmajor=True)
            assert head["grid"] <= max(cus, head["blocks"] if head["tile"] != 256 else 0) or head["tile"] != 256
        ops.set_cu_budget(0)
        assert {k: ops.gemm_plan(v) for k, v in step.items()} == base  # the default plan is what it was
    finally:
        ops.set_cu_budget(0)


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import mic_amd  # noqa: F401
    from mic_amd.train import GradReducer, plan_buckets

    n = 5000
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    buckets = plan_buckets(n, 1200, list(range(0, n, 500)))
    red = GradReducer(g, buckets)
    # opt-in bf16 exchange: sum of the ranks' bf16-rounded buckets, widened back to fp32
    g16 = torch.arange(n, dtype=torch.float32) * 0.37 * (rank + 1)
    r16 = GradReducer(g16, buckets, comm_dtype=torch.bfloat16)
    r16.start_step()
    r16.progress(n)
    r16.finish()
    exp16 = sum((torch.arange(n, dtype=torch.float32) * 0.37 * (r + 1)).to(torch.bfloat16) for r in range(world)).float()
    ok16 = torch.equal(g16, exp16)
    red.start_step()
    fired = []
    for off in range(500, n + 1, 500):  # backward reports progress segment by segment
        before = red.next
        red.progress(off)
        fired.append(red.next - before)
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(g, expect) and red.next == len(buckets) and sum(fired) == len(buckets) and ok16
    # the host side of one step's exchange at the bucket count of the real layout (25): what GradReducer.progress / finish cost the
    # issuing thread (bench.py puts the figure next to host_issue_ms_per_step; over RCCL the enqueue is asynchronous like this one)
    n25 = 25 * 4096
    g25 = torch.ones(n25) * (rank + 1)
    r25 = GradReducer(g25, [(i * 4096, (i + 1) * 4096) for i in range(25)])
    issue, total = [], []
    for _ in range(5):
        g25.fill_(float(rank + 1))
        r25.start_step()
        for i in range(25):
            r25.progress((i + 1) * 4096)
        issue.append(r25.host_s * 1e3)   # 25 asynchronous collectives issued
        r25.finish()                      # (CPU tensors: finish() also WAITS for the collectives; on the GPU it only enqueues)
        total.append(r25.host_s * 1e3)
    ok = ok and torch.equal(g25, torch.ones(n25) * sum(r + 1 for r in range(world))) and 0.0 < min(issue) <= min(total)  # (wall-clock figures are printed, not bounded: they depend on the host's load)
    if rank == 0:
        print(f"[reducer host side, gloo world {world}, 25 buckets] issue {min(issue):.3f} ms per step, with the waits {min(total):.3f} ms "
              "(best of 5)", flush=True)
    # pmean of per-rank masked means (main.py:679, 698): mean of means, not a token-weighted global mean
    m = torch.tensor([float(rank + 1), 0.0])
    dist.all_reduce(m)
    m /= world
    q.put((rank, bool(ok), float(m[0])))
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [True, True]
    assert all(abs(r[2] - 1.5) < 1e-9 for r in res)


def test_optimizer_state_files_roundtrip(tmp_path):
    """opt_state.msgpack / training_state.json (main.py:313-317, 330-343): wire layout of the optax adamw chain state and a
    bit-exact round trip of parameters + both moments through the Flax layouts."""
    import msgpack

    from mic_amd.checkpoint import load_flax_msgpack, load_train_state, save_flax_msgpack, save_train_state
    from mic_amd.params import ParamStore, unflatten_tree

    st = ParamStore(_small_cfg(), torch.float32, "cpu")
    st.init_random(seed=3)
    st.ensure_opt_state()
    g = torch.Generator().manual_seed(1)
    st.m.copy_(torch.randn(st.numel, generator=g))
    st.v.copy_(torch.rand(st.numel, generator=g))
    ref = {k: st.export_flat(k) for k in ("master", "m", "v")}
    d = str(tmp_path)
    save_flax_msgpack(os.path.join(d, "flax_model.msgpack"), unflatten_tree(ref["master"]))
    save_train_state(d, st, step=17)
    opt = load_flax_msgpack(os.path.join(d, "opt_state.msgpack"))
    assert set(opt) == {"0", "1", "2"} and set(opt["0"]) == {"count", "mu", "nu"} and opt["1"] == {} and set(opt["2"]) == {"count"}
    assert int(opt["0"]["count"]) == 17 and opt["0"]["count"].dtype == np.int32
    assert set(opt["0"]["mu"]["model"]) == {"encoder", "decoder", "shared", "visual_projection"}
    st2 = ParamStore(_small_cfg(), torch.float32, "cpu")
    assert load_train_state(d, st2) == 17
    for k in ("master", "m", "v"):
        back = st2.export_flat(k)
        assert all(np.array_equal(back[n], ref[k][n]) for n in ref[k]), k


def test_pt_checkpoint_ingestion_matches_twin(tmp_path):
    """`mbart_from_pt=True` (main.py:426): PyTorch state dicts of the PT twins -> Flax leaves; the oracle run on the converted
    leaves reproduces the twin's own forward (names, Dense/conv transposes, LayerNorm/Embed renames)."""
    transformers = pytest.importorskip("transformers")
    from mic_amd.checkpoint import convert_pt_state_dict, load_pt_state_dict
    from oracle import model_ref as M

    from util_small import ref_config

    rc = ref_config("erf", 1e-5)
    torch.manual_seed(0)
    vcfg = transformers.CLIPVisionConfig(hidden_size=rc.v_hidden, intermediate_size=rc.v_ffn, num_hidden_layers=rc.v_layers,
                                         num_attention_heads=rc.v_heads, image_size=rc.image_size, patch_size=rc.patch_size)
    mcfg = transformers.MBartConfig(vocab_size=rc.vocab_size, d_model=rc.d_model, decoder_layers=rc.d_layers, encoder_layers=1,
                                    decoder_attention_heads=rc.d_heads, encoder_attention_heads=rc.d_heads, decoder_ffn_dim=rc.d_ffn,
                                    encoder_ffn_dim=rc.d_ffn, max_position_embeddings=rc.max_position_embeddings, scale_embedding=True, dropout=0.0)
    clip = transformers.CLIPVisionModel(vcfg).eval()
    mbart = transformers.MBartForConditionalGeneration(mcfg).eval()
    cdir, mdir = tmp_path / "clip", tmp_path / "mbart"
    cdir.mkdir()
    mdir.mkdir()
    torch.save(clip.state_dict(), cdir / "pytorch_model.bin")
    from safetensors.torch import save_file

    save_file({k: v.contiguous() for k, v in mbart.state_dict().items() if k != "lm_head.weight" and "embed_tokens" not in k}, str(mdir / "model.safetensors"))
    shapes = M.param_shapes(rc)
    p = M.init_params(rc, seed=9, perturb_ln=True)
    enc = convert_pt_state_dict(load_pt_state_dict(str(cdir)), {k[len("model/encoder/"):] for k in shapes if k.startswith("model/encoder/")})
    dec = convert_pt_state_dict(load_pt_state_dict(str(mdir)), {k[len("model/"):] for k in shapes if k.startswith("model/")})
    n = 0
    for k, v in enc.items():
        assert tuple(v.shape) == tuple(shapes["model/encoder/" + k]), k
        p["model/encoder/" + k] = torch.from_numpy(np.array(v))
        n += 1
    for k, v in dec.items():
        if k.startswith("decoder/") or k.startswith("shared/"):
            assert tuple(v.shape) == tuple(shapes["model/" + k]), k
            p["model/" + k] = torch.from_numpy(np.array(v))
            n += 1
    assert n == sum(1 for k in shapes if k.startswith(("model/encoder/", "model/decoder/", "model/shared/")))  # every leaf was found
    g = torch.Generator().manual_seed(2)
    px = torch.randn(2, rc.image_size, rc.image_size, 3, generator=g)
    ids = torch.randint(4, rc.vocab_size, (2, 6), generator=g)
    with torch.no_grad():
        last, _ = M.vit_encoder(rc, p, px)
        twin_last = clip(pixel_values=px.permute(0, 3, 1, 2)).last_hidden_state
        assert (last - twin_last).abs().max().item() < 2e-5
        ehs = torch.randn(2, rc.v_seq, rc.d_model, generator=g)
        h = M.decoder_forward(rc, p, ids, torch.ones_like(ids), torch.arange(6)[None].expand(2, 6), ehs)
        twin_h = mbart.model.decoder(input_ids=ids, encoder_hidden_states=ehs).last_hidden_state
        assert (h - twin_h).abs().max().item() < 2e-5


def test_bleu_known_answers():
    """compute_bleu (main.py:573, 596-598) against hand-computed corpus BLEU values."""
    import math

    from mic_amd.evaluation import compute_bleu, compute_metrics, simple_word_tokenize

    ref = "the cat is on the mat".split()
    # identical -> 1; disjoint -> 0
    assert compute_bleu([ref], [[ref]], 4)["bleu"] == 1.0
    assert compute_bleu(["a b c d".split()], [[ref]], 4)["bleu"] == 0.0
    # clipping: "the the the the the the the" vs reference with 2 x "the": p1 = 2/7
    r = compute_bleu(["the the the the the the the".split()], [[ref]], 1)
    assert abs(r["precisions"][0] - 2 / 7) < 1e-12 and r["brevity_penalty"] == 1.0 and abs(r["bleu"] - 2 / 7) < 1e-12
    # hyp "the cat sat on the mat": p1 = 5/6, p2 = 3/5, p3 = 1/4, p4 = 0/3 -> BLEU-4 = 0 (no smoothing), BLEU-2 = sqrt(5/6*3/5)
    hyp = "the cat sat on the mat".split()
    r = compute_bleu([hyp], [[ref]], 4)
    assert r["precisions"] == [5 / 6, 3 / 5, 1 / 4, 0.0] and r["bleu"] == 0.0
    assert abs(compute_bleu([hyp], [[ref]], 2)["bleu"] - math.sqrt(5 / 6 * 3 / 5)) < 1e-12
    # brevity penalty with the shortest reference: hyp len 4, refs len 6 and 5 -> bp = exp(1 - 5/4)
    r = compute_bleu(["the cat is on".split()], [[ref, "the cat is on mat".split()]], 1)
    assert abs(r["brevity_penalty"] - math.exp(1 - 5 / 4)) < 1e-12 and r["precisions"][0] == 1.0
    # corpus-level pooling (not a mean of sentence scores)
    r = compute_bleu([hyp, ref], [[ref], [ref]], 4)
    p = [(5 + 6) / 12, (3 + 5) / 10, (1 + 4) / 8, (0 + 3) / 6]
    assert abs(r["bleu"] - math.exp(sum(math.log(x) for x in p) / 4)) < 1e-12
    assert simple_word_tokenize("Ein Hund, der läuft!") == ["Ein", "Hund", ",", "der", "läuft", "!"]
    vocab = {5: "a", 6: "dog", 7: "runs", 8: "."}
    dec = lambda rows: [" ".join(vocab[t] for t in row if t in vocab) for row in rows]
    m = compute_metrics([[2, 5, 6, 7, 8, 2, 1]], [[250004, 5, 6, 7, 8, 2, 1]], dec)
    assert m == {"BLEU-1": 1.0, "BLEU-2": 1.0, "BLEU-3": 1.0, "BLEU-4": 1.0}


def test_tool_scripts_parse():
    """the shell scripts under tools/ (profiling passes, A/B drivers, the first-multi-GPU-lease script nobody has been able to run yet)
    at least parse, and the Python tools compile"""
    import glob
    import py_compile
    import subprocess

    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))):
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))):
        py_compile.compile(f, doraise=True)
