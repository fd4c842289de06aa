"""CPU-only checks of the host layer: C-ABI symbols, state-dict compatibility with the reference
layout, loud failure without a GPU, and the host-side mean-shift pieces."""
import ctypes
import os

import numpy as np
import pytest
import torch

from unseenobjectswithmeanshift_amd import _lib
from unseenobjectswithmeanshift_amd import synthetic as syn


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _lib.declared_symbols()
    assert len(declared) >= 20
    for s in declared:
        assert hasattr(L, s), s
    assert set(declared) == set(_lib._SIGNATURES), set(declared) ^ set(_lib._SIGNATURES)
    import re
    with open(_lib.HEADER_PATH) as f:
        header_abi = int(re.search(r"#define\s+MSM_ABI_VERSION\s+(\d+)", f.read()).group(1))
    assert _lib.lib().msm_abi_version() == _lib.ABI_VERSION == header_abi


def test_argument_errors_are_reported_without_a_gpu():
    L = _lib.lib()
    rc = L.msm_gemm_f32(None, None, None, None, None, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, None)
    assert rc == -1 and b"null pointer" in L.msm_last_error_string()
    rc = L.msm_mask_logits_fwd(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None, None,
                               1, 100, 250, 120, 160, 0, 0, 0, 0, None, 0, None)
    assert rc == -1 and b"multiple of 32" in L.msm_last_error_string()
    assert L.msm_hypersphere_attn_workspace(8, 100, 4800, 8) > 0
    # the f32_split hill climb keeps X as three bf16 planes in its workspace: 96 floats per (padded) row more than the fp32 form
    n, S = 307200, 100
    assert L.msm_ms_hill_climb_split_workspace(n, S) == L.msm_ms_hill_climb_workspace(n, S) + 96 * n + 4
    assert L.msm_ms_hill_climb_split_workspace(33, 1) == L.msm_ms_hill_climb_workspace(33, 1) + 96 * 64 + 4
    rc = L.msm_ms_hill_climb_split(ctypes.c_void_p(16), 100, 64, ctypes.c_void_p(16), 5, 20.0, 1, ctypes.c_void_p(16), 8, None)
    assert rc == -3 and b"workspace too small" in L.msm_last_error_string()
    rc = L.msm_conv3x3_c64_split(None, ctypes.c_void_p(16), ctypes.c_void_p(16), None, 0, 1, 4, 4, None)
    assert rc == -1 and b"null pointer" in L.msm_last_error_string()
    rc = L.msm_topk_class_scores_gather(ctypes.c_void_p(16), 1, 10, 3, 4, ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16),
                                        ctypes.c_void_p(16), 2, 4, ctypes.c_void_p(16), None)
    assert rc == -1 and b"bad gather arguments" in L.msm_last_error_string()


def test_ops_refuse_cpu_tensors():
    from unseenobjectswithmeanshift_amd import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops.gemm(torch.zeros(4, 32), torch.zeros(8, 32))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.mask_logits(torch.zeros(1, 4, 32), torch.zeros(1, 32, 4, 4))
    with pytest.raises(ValueError, match="precision"):
        ops.ms_hill_climb(torch.zeros(8, 64), torch.zeros(2, 64), 20.0, 1, precision="fp8")
    with pytest.raises(RuntimeError, match="GPU"):
        ops.ms_hill_climb(torch.zeros(8, 64), torch.zeros(2, 64), 20.0, 1, precision="f32_split")


def test_state_dict_layout_matches_reference():
    """synthetic.*_param_shapes is asserted equal to the reference modules' state_dict() in
    tests/golden/make_golden.py; the HIP-backed modules must expose exactly the same keys."""
    from unseenobjectswithmeanshift_amd.meta_arch import build_resnet50_head
    head = build_resnet50_head()
    dec = {k: tuple(v.shape) for k, v in head.predictor.state_dict().items()}
    ref = {k: tuple(v) for k, v in syn.decoder_param_shapes().items()}
    assert dec == ref and list(dec) == list(ref)
    pd = {k: tuple(v.shape) for k, v in head.pixel_decoder.state_dict().items()}
    ref = {k: tuple(v) for k, v in syn.pixel_decoder_param_shapes().items()}
    assert pd == ref
    keys = list(head.state_dict())
    assert all(k.startswith(("pixel_decoder.", "predictor.")) for k in keys)
    # v1 checkpoints used "static_query" (meanshiftformer_transformer_decoder.py:348-369)
    sd = syn.synth_state_dict(syn.decoder_param_shapes())
    sd["static_query.weight"] = sd.pop("query_feat.weight")
    head.predictor.load_state_dict(sd, strict=True)


def test_unsupported_configurations_raise():
    from unseenobjectswithmeanshift_amd.modeling import MeanShiftTransformerDecoder
    kw = dict(in_channels=64, mask_classification=True, num_classes=2, hidden_dim=256, num_queries=100, nheads=8,
              dim_feedforward=2048, dec_layers=9, pre_norm=False, mask_dim=256, enforce_input_project=False)
    with pytest.raises(NotImplementedError):
        MeanShiftTransformerDecoder(**{**kw, "pre_norm": True})
    with pytest.raises(NotImplementedError):
        MeanShiftTransformerDecoder(**{**kw, "disable_attention_mask": True})


def test_connected_components_host(golden):
    from unseenobjectswithmeanshift_amd import mean_shift as ms
    g = golden("mean_shift")
    assert torch.equal(ms.connected_components(torch.from_numpy(g["s_Z"]), 0.04), torch.from_numpy(g["s_cc"]))
    assert torch.equal(ms.connected_components(torch.from_numpy(g["chain"]), 0.04), torch.from_numpy(g["cc_chain"]))
    with pytest.raises(NotImplementedError):
        ms.connected_components(torch.from_numpy(g["chain"]), 0.04, metric="euclidean")


def test_instances_container():
    from unseenobjectswithmeanshift_amd.meta_arch import Instances, combine_masks, get_confident_instances
    m = torch.zeros(3, 4, 5)
    m[0, :2] = 1
    m[1, 1:3] = 1
    m[2, 3] = 1
    inst = Instances((4, 5), pred_masks=m, scores=torch.tensor([0.9, 0.5, 0.8]), pred_classes=torch.tensor([1, 1, 0]))
    conf = get_confident_instances({"instances": inst}, score=0.6)
    assert len(conf) == 2
    lab = combine_masks(conf)
    assert lab[0, 0] == 2 and lab[3, 0] == 3 and lab[2, 0] == 0
    top = get_confident_instances({"instances": inst}, topk=True, low_threshold=0.4)
    assert len(top) == 2 and bool((top.pred_classes == 1).all())
    lab = combine_masks(top)
    assert lab[1, 0] == 3 and lab[0, 0] == 2          # later instances overwrite earlier ones


def test_library_options_are_explicit_and_default_to_auto():
    """Kernel-selection overrides go through msm_set_option (no environment variable is read by the library); host-only calls."""
    import subprocess
    L = _lib.lib()
    for i, name in enumerate(_lib.OPTIONS):
        assert L.msm_get_option(i) == _lib.OPT_AUTO, name
    assert _lib.set_option("MASK_NC", 2) == _lib.OPT_AUTO and L.msm_get_option(_lib.OPTIONS.index("MASK_NC")) == 2
    with _lib.option("ATTN_KERNEL", 3):
        assert L.msm_get_option(_lib.OPTIONS.index("ATTN_KERNEL")) == 3
    assert L.msm_get_option(_lib.OPTIONS.index("ATTN_KERNEL")) == _lib.OPT_AUTO
    _lib.set_option("MASK_NC")
    assert L.msm_set_option(len(_lib.OPTIONS), 1) != 0 and b"unknown key" in L.msm_last_error_string()
    assert L.msm_set_option(len(_lib.OPTIONS) - 1, 1) == 0 and L.msm_set_option(len(_lib.OPTIONS) - 1, -1) == 0     # enum and OPTIONS agree in length
    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in out


def test_graph_stale_check_is_cheap_and_sees_updates():
    """graphs.StaleCheck (the per-replay staleness test of captured graphs): moves on in-place parameter updates, on
    plan-attribute assignments with a NEW value and on invalidate(); does not move otherwise; costs tens of microseconds
    where the exhaustive signature costs a millisecond."""
    import time
    from unseenobjectswithmeanshift_amd import graphs
    from unseenobjectswithmeanshift_amd.meta_arch import build_resnet50_head
    head = build_resnet50_head()
    chk = graphs.StaleCheck(head)
    s0 = chk()
    assert chk() == s0
    with torch.no_grad():
        head.predictor.class_embed.bias.add_(1.0)
    s1 = chk()
    assert s1 != s0
    head.predictor.aux_outputs = False                 # same value: no plan change
    assert chk() == s1
    head.predictor.aux_outputs = True
    s2 = chk()
    assert s2 != s1
    head.predictor.ffn_parts = 4                       # every plan attribute is covered, also the tuning ones
    s3 = chk()
    assert s3 != s2
    chk.invalidate()
    assert chk() != s3
    strict = graphs.StaleCheck(head, strict=True)
    assert strict() == graphs.param_signature(head)
    t0 = time.perf_counter()
    for _ in range(200):
        chk()
    fast = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(20):
        strict()
    slow = (time.perf_counter() - t0) / 20
    assert fast < 0.2 * slow, (fast, slow)


def test_stale_check_sees_replaced_and_repointed_middle_parameters():
    """Round-3 advisor finding: the cheap signatures compared the tensor count, the version sum and the FIRST / LAST address, so
    ``p.data = new`` on a middle tensor, ``m.weight = nn.Parameter(...)`` or ``load_state_dict(assign=True)`` kept stale packed
    weights and graphs.  Now: every address enters the key (a sum), and the cached tensor lists are rebuilt when a Parameter /
    buffer object is (re)registered anywhere (``_plan.TensorList`` on torch's registration hooks)."""
    from torch import nn
    from unseenobjectswithmeanshift_amd import _plan, graphs
    from unseenobjectswithmeanshift_amd.meta_arch import build_resnet50_head
    head = build_resnet50_head()
    chk = graphs.StaleCheck(head)
    mid = head.pixel_decoder.transformer.encoder.layers[2].linear1          # neither first nor last
    s0 = chk()
    mid.weight.data = mid.weight.data.clone()                                # same version, new storage
    s1 = chk()
    assert s1 != s0
    mid.weight = nn.Parameter(mid.weight.detach().clone())                   # a new Parameter object at version 0
    s2 = chk()
    assert s2 != s1
    sd = {k: v.clone() for k, v in mid.state_dict().items()}
    mid.load_state_dict(sd, assign=True)
    assert chk() != s2
    # the module-level lists behind the packed-weight caches follow the same rule
    tl = _plan.TensorList.of(head.pixel_decoder, "transformer.encoder")
    k0 = _plan.version_key(tl())
    mid.weight = nn.Parameter(mid.weight.detach().clone())
    k1 = _plan.version_key(tl())
    assert k1 != k0 and any(t is mid.weight for t in tl())
    mid.bias.data = mid.bias.data.clone()
    assert _plan.version_key(tl()) != k1
    # the inference-plan attributes of the meta-arch are plan attributes too (advisor: K selects kernels inside a captured graph)
    assert {"test_topk_per_image", "topk_before_masks", "hm_activations"} <= _plan.PLAN_ATTRS


def test_tensor_lists_follow_deepcopy_and_pickle():
    """Round-4 advisor finding: the tensor lists behind the packed-weight caches / graph staleness keys were built from lambdas
    closing over the module; deepcopy copies functions atomically, so a copied model kept computing its key from the ORIGINAL's
    tensors (an in-place update of the copy went unseen once the parameter epoch had moved) and pickle refused the lambda.  Now
    the builder is an owner reference (re-bound by deepcopy, pickled with the module) and plain functions are refused."""
    import copy
    import pickle
    from torch import nn
    from unseenobjectswithmeanshift_amd import _plan, graphs
    from unseenobjectswithmeanshift_amd.meta_arch import build_resnet50_head
    head = build_resnet50_head()
    pd = head.pixel_decoder
    pd._enc_params = _plan.TensorList.of(pd, "transformer.encoder")          # what _encoder_stream() installs on first use
    k_orig = _plan.version_key(pd._enc_params())                             # list cached on the original
    chk = graphs.StaleCheck(head)
    chk()
    twin = copy.deepcopy(head)
    chk2 = copy.deepcopy(chk)
    assert twin.pixel_decoder._enc_params._owner is twin.pixel_decoder       # re-bound to the copy, cache dropped
    assert twin.pixel_decoder._enc_params._list is None
    _ = nn.Linear(2, 2)                                                      # any module construction moves the parameter epoch
    k_twin = _plan.version_key(twin.pixel_decoder._enc_params())
    with torch.no_grad():
        twin.pixel_decoder.transformer.encoder.layers[1].linear1.weight.add_(1.0)
    assert _plan.version_key(twin.pixel_decoder._enc_params()) != k_twin     # the copy sees its own update ...
    assert _plan.version_key(pd._enc_params()) == k_orig                     # ... and the original did not change
    assert all(a is not b for a, b in zip(pd._enc_params(), twin.pixel_decoder._enc_params()))
    # a deep-copied StaleCheck follows ITS model copy (graphs.py keeps model + check together in GraphedInference)
    assert chk2._tensors._owner is not head
    # bound methods are re-bound by deepcopy as well
    tl = _plan.TensorList(head.predictor.parameters)
    tl()
    assert copy.deepcopy(tl)._build.__self__ is not head.predictor
    with pytest.raises(TypeError):
        _plan.TensorList(lambda: head.parameters())
    # pickle: a module that has used its list round-trips (the cached list is not part of the state)
    blob = pickle.dumps(pd)
    back = pickle.loads(blob)
    assert back._enc_params._owner is back and back._enc_params._list is None
    assert len(back._enc_params()) == len(pd._enc_params())


class _NotATensor:
    pass


def test_checkpoint_loader_is_tensors_only_and_says_how_to_opt_out(tmp_path):
    """load_checkpoint_file: numpy payloads of either numpy major version's module path are allow-listed; anything else fails
    with a message that names unsafe=True (round-3 advisor finding: a bare UnpicklingError with no hint)."""
    import pickle
    from unseenobjectswithmeanshift_amd import checkpoint as c
    p = str(tmp_path / "a.pth")
    torch.save({"model": {"w": np.arange(4, dtype=np.float32), "t": torch.ones(2)}, "iteration": np.float64(3.0)}, p)
    got = c.load_checkpoint_file(p)
    assert got["model"]["w"].tolist() == [0.0, 1.0, 2.0, 3.0] and float(got["iteration"]) == 3.0

    torch.save({"model": {"w": _NotATensor()}}, p)
    with pytest.raises(pickle.UnpicklingError, match="unsafe=True"):
        c.load_checkpoint_file(p)


def test_round4_weight_layouts_match_the_header_formulas():
    """Host-side packing of the round-4 kernels, checked against the index formulas include/msm_hip.h documents (pure tensor code:
    runs without a GPU): the hi + lo fragment order of msm_conv1x1_in_lp, the separable K/V constant, and the sizes the library
    reports for the bf16 plan's prologue blocks."""
    from unseenobjectswithmeanshift_amd import ops
    from unseenobjectswithmeanshift_amd._lib import lib
    g = torch.Generator().manual_seed(3)
    Cin = 512
    w = torch.randn(64, Cin, generator=g) * Cin ** -0.5
    wp = ops.pack_conv_in_weight_lp(w)
    assert wp.dtype == torch.bfloat16 and wp.numel() == 2 * 64 * Cin
    hi = w.to(torch.bfloat16)
    planes = torch.stack([hi, (w - hi.float()).to(torch.bfloat16)])
    k, o = torch.meshgrid(torch.arange(Cin), torch.arange(64), indexing="ij")
    for pl in range(2):
        idx = ((((k // 32) * 4 + o // 16) * 2 + pl) * 64 + ((k % 32) // 8) * 16 + o % 16) * 8 + k % 8
        assert torch.equal(wp[idx], planes[pl].t())
    # hi + lo carries the weight to 2^-16 relative
    assert float(((planes[0].float() + planes[1].float()) - w).abs().max()) <= float(w.abs().max()) * 2.0 ** -15
    with pytest.raises(RuntimeError):
        ops.pack_conv_in_weight_lp(torch.zeros(64, 128))
    # separable K/V constant: H row vectors then W column vectors; token (y, x) gets row[y] + col[x]
    H, W, N = 5, 7, 256
    rc = torch.randn(H + W, N, generator=g)
    dense = ops.dense_kv_constant(rc, W)
    assert tuple(dense.shape) == (H * W, N) and torch.equal(dense.view(H, W, N)[3, 4], rc[3] + rc[H + 4])
    assert ops.dense_kv_constant(dense, 0) is dense
    # the prologue blocks of the bf16 plan: value (16 KiB) + projection (72 KiB) as [row block][k-group][hi, lo] 1-KiB fragments
    assert lib().msm_encoder_prologue_hm_weight_bytes() == 16384 + 18 * 4096
    blocks, small = ops.pack_encoder_prologue_hm(torch.randn(64, 64, generator=g), torch.randn(288, 64, generator=g),
                                                 torch.randn(64, generator=g), torch.randn(288, generator=g))
    assert blocks.dtype == torch.int16 and blocks.numel() * 2 == 16384 + 18 * 4096 and small.numel() == 352


def test_bench_line_fits_the_drivers_record():
    """bench.compact_line: the ONE stdout line of bench.py stays within the 8 KB tail of stdout the driver's record keeps (round 5's
    27 KB line was not parsed).  Built from a full result document with every optional entry present (tests/golden/bench_full_sample.json:
    a real N = 1 run's document with eight per-rank records and an RCCL `collective` entry grafted on), then from one whose strings and
    lists are inflated: at most 8000 bytes either way, valid JSON, the contract's keys and the flat roofline / cpu_baseline scalars present."""
    import copy
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    with open(os.path.join(root, "tests", "golden", "bench_full_sample.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full).encode()) > 20000                       # the document itself is far over the budget
    line = bench.compact_line(copy.deepcopy(full), "gpurun_out/bench_full.json")
    raw = json.dumps(line)
    assert len(raw.encode()) <= bench.LINE_BUDGET == 8000 and "\n" not in raw
    back = json.loads(raw)
    assert back == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "steps_requested", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "collective", "summary"):
        assert k in back, k
    assert back["metric"] == bench.METRIC and back["config"]["workload"].startswith("configs[1]")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "one_batch_in_flight_ms",
              "mask_step_frac", "mask_step_avg_launch_ms", "mask_step_literal_frac"):
        assert k in back["roofline"], k
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample", "pixel_decoder_ms", "decoder_ms", "post_process_ms", "mean_shift_images_per_sec"):
        assert k in back["cpu_baseline"], k
    assert back["summary"]["c2"]["dt"] == "f16" and back["summary"]["c2_bf16"]["dt"] == "bf16"
    assert len(back["per_rank"]["images_per_sec"]) == 8 and back["collective"]["world_size"] == 8
    assert all(len(v) <= 120 for v in _strings(back))                   # the driver truncates strings at 120 characters
    # inflated: long strings everywhere, 64 ranks, a summary over its own budget -> still within the budget, contract keys intact
    fat = copy.deepcopy(full)
    fat["per_rank"] = [dict(fat["per_rank"][0], rank=i) for i in range(64)]
    fat["config"]["workload"] = "w" * 5000
    fat["roofline"]["kernel"] = "k" * 5000
    fat["cpu_baseline"]["sample"] = "s" * 5000
    fat["cpu_baseline"].pop("sample_short", None)
    fat["collective"]["transport"] = "t" * 5000
    fat["summary"] = {f"k{i}": {"v": float(i), "note": "n" * 100} for i in range(200)}
    raw = json.dumps(bench.compact_line(fat, "gpurun_out/" + "d" * 500))
    assert len(raw.encode()) <= 8000
    back = json.loads(raw)
    assert back["value"] == full["value"] and back["roofline"]["frac"] == full["roofline"]["frac"] and back["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]


def _strings(obj):
    if isinstance(obj, str):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _strings(v)
    elif isinstance(obj, list):
        for v in obj:
            yield from _strings(v)
