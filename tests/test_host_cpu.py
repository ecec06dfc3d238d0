"""CPU-side checks: the C-ABI library loads and exports every symbol include/lsam.h declares,
the drop-in modules reproduce the reference's state_dict keys / geometry, the plan geometry
in the library matches the oracle's restatement of Hiera.__init__, and the product path
refuses to run without the GPU (no fallback).  No compute calls are made here."""
import ctypes as C
import os
import re
import types

import pytest
import torch

from conftest import BPLUS, REPO, TINY


def test_library_exports_every_declared_symbol():
    from image_segmentation_amd import _C
    hdr = open(os.path.join(REPO, "include", "lsam.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lsam_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = _C.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in lsam.h but not exported"
    assert declared == set(_C.EXPORTED)
    assert L.lsam_version() >= 1


@pytest.mark.parametrize("cfg", [TINY, BPLUS, dict(embed_dim=144, num_heads=2, stages=(2, 6, 36, 4),
                                                   global_att_blocks=(23, 33, 43), window_spec=(8, 4, 16, 8))])
def test_module_tree_matches_oracle_enumeration(cfg):
    import image_segmentation_amd as m
    from oracle import hiera_oracle as O
    from oracle.params import encoder_param_shapes
    spec = O.block_specs(**cfg)
    enc = m.build_image_encoder(dict(cfg))
    want = dict(encoder_param_shapes(spec, rank=0))
    got = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    assert got == want
    assert enc.trunk.stage_ends == spec["stage_ends"]
    assert enc.trunk.q_pool_blocks == spec["q_pool_blocks"]
    assert enc.trunk.channel_list == spec["channel_list"]
    for blk, s in zip(enc.trunk.blocks, spec["blocks"]):
        assert (blk.dim, blk.dim_out, blk.attn.num_heads, blk.window_size, blk.q_stride) == \
               (s["dim"], s["dim_out"], s["num_heads"], s["window_size"], s["q_stride"])
    # wrapping renames the base Linear and adds the four factors on blocks[-2:] only
    m.wrap_image_encoder_lora(types.SimpleNamespace(image_encoder=enc), 4)
    got = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    want = dict(encoder_param_shapes(spec, rank=4))
    n = len(spec["blocks"])
    for i in (n - 2, n - 1):
        for leaf in ("weight", "bias"):
            want[f"trunk.blocks.{i}.attn.qkv.qkv.{leaf}"] = want.pop(f"trunk.blocks.{i}.attn.qkv.{leaf}")
    assert got == want
    trainable = [k for k, p in enc.named_parameters() if p.requires_grad]
    assert len(trainable) == 8 and all("linear_" in k for k in trainable)
    for k, p in enc.named_parameters():
        if "linear_b_" in k:
            assert float(p.abs().max()) == 0.0   # B = 0 init (lora_qkv.py:191-192)


def test_library_geometry_matches_oracle():
    from image_segmentation_amd import _C
    from oracle import hiera_oracle as O
    for cfgd, size in ((TINY, 256), (BPLUS, 1024)):
        spec = O.block_specs(**cfgd)
        full = dict(stages=(2, 3, 16, 3), window_spec=(8, 4, 14, 7), global_att_blocks=(12, 16, 20))
        full.update(cfgd)
        cfg = _C.HieraConfig()
        cfg.dtype, cfg.img_size = _C.BF16, size
        cfg.embed_dim, cfg.num_heads, cfg.num_stages = full["embed_dim"], full["num_heads"], 4
        for i in range(4):
            cfg.stages[i], cfg.window_spec[i] = full["stages"][i], full["window_spec"][i]
        cfg.n_global = len(full["global_att_blocks"])
        for i, b in enumerate(full["global_att_blocks"]):
            cfg.global_att_blocks[i] = b
        cfg.q_pool, cfg.lora_rank, cfg.n_lora_blocks = 3, 4, 2
        h = C.c_void_p()
        _C.check(_C.lib().lsam_hiera_create(C.byref(cfg), C.byref(h)))
        assert _C.lib().lsam_hiera_num_blocks(h) == len(spec["blocks"])
        grid = size // 4
        for i, s in enumerate(spec["blocks"]):
            v = [C.c_int() for _ in range(6)]
            _C.check(_C.lib().lsam_hiera_block_info(h, i, *[C.byref(x) for x in v]))
            assert [x.value for x in v] == [s["dim"], s["dim_out"], s["num_heads"], s["window_size"],
                                            1 if s["q_stride"] else 0, grid]
            if s["q_stride"]:
                grid //= 2
        assert _C.lib().lsam_hiera_arena_bytes(h) > 0
        ws1 = _C.lib().lsam_hiera_workspace_bytes(h, 1, 1)
        assert _C.lib().lsam_hiera_workspace_bytes(h, 2, 1) > ws1 > _C.lib().lsam_hiera_workspace_bytes(h, 1, 0) > 0
        _C.lib().lsam_hiera_destroy(h)


def test_library_rejects_unsupported_configs():
    from image_segmentation_amd import _C
    cfg = _C.HieraConfig()
    cfg.dtype, cfg.img_size, cfg.embed_dim, cfg.num_heads, cfg.num_stages = 1, 256, 144, 1, 4   # head_dim 144 > 128
    for i, (s, w) in enumerate(zip((1, 2, 7, 2), (8, 4, 14, 7))):
        cfg.stages[i], cfg.window_spec[i] = s, w
    cfg.q_pool = 3
    h = C.c_void_p()
    rc = _C.lib().lsam_hiera_create(C.byref(cfg), C.byref(h))
    assert rc == -2 and b"head_dim" in _C.lib().lsam_last_error()
    with pytest.raises(_C.LsamError):
        _C.check(_C.lib().lsam_hiera_create(None, C.byref(h)))


def test_no_cpu_fallback():
    import image_segmentation_amd as m
    from image_segmentation_amd._C import LsamError
    enc = m.build_image_encoder(dict(TINY))
    with pytest.raises(LsamError):
        enc(torch.zeros(1, 3, 256, 256))
    with pytest.raises(LsamError):
        enc.trunk.blocks[0](torch.zeros(1, 64, 64, 56))


def test_lora_save_format_matches_reference_filter(tmp_path):
    """custom_save keeps the reference's q-only filter (lora_qkv.py:216) unless include_v."""
    import image_segmentation_amd as m
    enc = m.build_image_encoder(dict(TINY))
    holder = types.SimpleNamespace(image_encoder=enc)
    m.wrap_image_encoder_lora(holder, 4)
    f = tmp_path / "lora.pt"
    m.custom_save_lora_parameters(enc, f)
    sd = torch.load(f, weights_only=True)
    assert len(sd) == 4 and all(("_a_q" in k or "_b_q" in k) for k in sd)
    m.custom_save_lora_parameters(enc, f, include_v=True)
    assert len(torch.load(f, weights_only=True)) == 8
    m.custom_load_lora_parameters(enc, f)


def test_lora_bank_switches_sets_in_memory(tmp_path):
    """LoRABank: the in-memory replacement of the save -> load -> forward -> load sequence of the reference's
    distillation teacher pass (train_cl.py:209-228).  Host logic only: values, identity of the parameters,
    independence of the stored sets, and interchange with the reference's q-only file format."""
    import image_segmentation_amd as m
    enc = m.build_image_encoder(dict(TINY))
    m.wrap_image_encoder_lora(types.SimpleNamespace(image_encoder=enc), 4)
    lora = {n: p for n, p in enc.named_parameters() if "linear_a_" in n or "linear_b_" in n}
    assert len(lora) == 8
    bank = m.LoRABank(enc)
    with torch.no_grad():
        for i, p in enumerate(lora.values()):
            p.fill_(1.0 + i)
    bank.store("domain0")
    with torch.no_grad():
        for p in lora.values():
            p.mul_(-2.0)
    bank.store("domain1")
    ids = {n: id(p) for n, p in lora.items()}
    bank.activate("domain0")
    assert all(torch.all(p == 1.0 + i) for i, p in enumerate(lora.values()))
    bank.activate("domain1")
    assert all(torch.all(p == -2.0 * (1.0 + i)) for i, p in enumerate(lora.values()))
    assert {n: id(p) for n, p in enc.named_parameters() if n in ids} == ids      # same Parameter objects
    with torch.no_grad():
        next(iter(lora.values())).zero_()                                          # training moves on ...
    bank.activate("domain1")
    assert torch.all(next(iter(lora.values())) == -2.0)                            # ... the stored set did not
    assert sorted(bank.keys()) == ["domain0", "domain1"] and "domain0" in bank
    with pytest.raises(KeyError):
        bank.activate("nope")
    # a reference-format file (q factors only) becomes a set; v factors keep the model's current values
    f = tmp_path / "lora_q.pt"
    bank.activate("domain0")
    m.custom_save_lora_parameters(enc, f)
    bank.activate("domain1")
    bank.load_state_dict("from_file", torch.load(f, weights_only=True))
    bank.activate("from_file")
    for i, (n, p) in enumerate(lora.items()):
        want = 1.0 + i if ("_a_q" in n or "_b_q" in n) else -2.0 * (1.0 + i)
        assert torch.all(p == want), n
    with pytest.raises(ValueError):
        m.LoRABank(m.build_image_encoder(dict(TINY)))
