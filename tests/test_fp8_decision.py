"""The fp8 decision, by measurement.

BASELINE.json's north star computes in f16 and bounds the per-pixel MAE of the [0,1]-normalised depth against the CPU
reference at 1e-3. Would an fp8 (OCP e4m3) MFMA path for the encoder GEMMs -- the only place where doubling the MFMA
rate could matter -- stay inside that bound? This test answers with the CPU oracle on the full Depth-Anything-V2-Small
shape (518 x 518, 12 layers, seeded weights): the same forward pass is run with

  * the encoder's linear weights rounded to e4m3 with one scale per output channel ("weights only"), and
  * additionally every linear's input rows rounded to e4m3 with one scale per token (what an fp8 x fp8 MFMA consumes),

and the MAE of the normalised depth against the unquantised oracle is compared with the bound. The numbers this prints
are quoted in DESIGN.md; the assertions pin the decision: weights + activations in e4m3 break the bound by a wide margin,
so the product keeps f16 operands (the f16 path measures 1.9e-4). Nothing here touches the HIP library."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from visioncpp_amd import synth

MAE_BOUND = 1e-3  # BASELINE.json north_star


def _round_e4m3(a: np.ndarray) -> np.ndarray:
    """Nearest OCP e4m3fn value (3 mantissa bits, normals from 2^-6, subnormal step 2^-9, max 448)."""
    a = np.asarray(a, np.float32)
    mag = np.minimum(np.abs(a), 448.0)
    ex = np.floor(np.log2(np.maximum(mag, 2.0 ** -20)))
    ex = np.maximum(ex, -6.0)
    step = np.exp2(ex - 3.0)
    r = np.minimum(np.rint(mag / step) * step, 448.0)
    return (np.sign(a) * r).astype(np.float32)


def test_e4m3_rounding_matches_the_oracle_helper():
    lib = oracle.lib()
    lib.vo_round_e4m3.restype = C.c_float
    lib.vo_round_e4m3.argtypes = [C.c_float]
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(500) * 100, rng.standard_normal(500) * 0.01, [0.0, 448.0, 500.0, -1000.0, 2.0 ** -9, 2.0 ** -10]])
    got = np.array([lib.vo_round_e4m3(float(v)) for v in vals.astype(np.float32)], np.float32)
    assert np.array_equal(got, _round_e4m3(vals))
    # spot values: 3 mantissa bits -> steps of 1/8 of the power of two below
    assert _round_e4m3(np.float32(1.06)) == np.float32(1.0) and _round_e4m3(np.float32(1.07)) == np.float32(1.125)
    assert _round_e4m3(np.float32(300.0)) == np.float32(288.0) and _round_e4m3(np.float32(1e6)) == np.float32(448.0)


def _quantise_encoder_weights(sd):
    out = dict(sd)
    for k, w in sd.items():
        if k.startswith("backbone.encoder.layer.") and k.endswith(".weight") and w.ndim == 2:
            sc = 448.0 / np.maximum(np.abs(w).max(axis=1, keepdims=True), 1e-30)
            out[k] = (_round_e4m3(w * sc) / sc).astype(np.float32)
    return out


@pytest.mark.timeout(600)
def test_fp8_encoder_breaks_the_mae_bound():
    cfg = synth.SMALL
    sd = {k: v.astype(np.float16).astype(np.float32) for k, v in synth.state_dict(cfg, seed=0).items()}  # the f16 file contents
    tensors, conv_idx = synth.gguf_tensors(sd, "whcn")
    params = oracle.make_params(gelu_mode=oracle.GELU_TANH_F32)
    img = synth.images(1, 518, 518, seed=1234)[0]

    lib = oracle.lib()
    lib.vo_set_linear_act_quant.argtypes = [C.c_int]
    ref, _ = oracle.Model(tensors, conv_idx).compute(params, img)

    tq, _ = synth.gguf_tensors(_quantise_encoder_weights(sd), "whcn")
    mq = oracle.Model(tq, conv_idx)
    w_only, _ = mq.compute(params, img)
    lib.vo_set_linear_act_quant(1)
    try:
        w_act, _ = mq.compute(params, img)
    finally:
        lib.vo_set_linear_act_quant(0)

    mae_w = float(np.abs(w_only - ref).mean())
    mae_wa = float(np.abs(w_act - ref).mean())
    print(f"\nfp8 what-if, Depth-Anything-V2-S 518x518: MAE weights-only e4m3 = {mae_w:.2e}, weights + activations e4m3 = {mae_wa:.2e} (bound {MAE_BOUND:.0e})")
    assert mae_wa > 2 * MAE_BOUND, "fp8 x fp8 encoder GEMMs stayed inside the MAE bound: revisit the decision in DESIGN.md"
    assert mae_wa > mae_w > 0


# ---- configs[4]: MobileSAM's TinyViT encoder "fp8 GGUF weights on CDNA4 fp8 MFMA", tolerance = mask IoU -----------------------------

def _iou(a: np.ndarray, b: np.ndarray) -> float:
    a, b = a > 0, b > 0
    u = np.logical_or(a, b).sum()
    return float(np.logical_and(a, b).sum() / u) if u else 1.0


def _quantise_tinyvit_linears(sd):
    """e4m3 with one scale per output channel for the transformer stages' qkv / proj / fc1 / fc2 (the GEMMs an fp8 MFMA path would
    run; the convolutional stage and the depthwise convs stay f16)."""
    out = dict(sd)
    n = 0
    for k, w in sd.items():
        if k.endswith(".weight") and w.ndim == 2 and any(t in k for t in (".attn.qkv.", ".attn.proj.", ".mlp.fc1.", ".mlp.fc2.")):
            sc = 448.0 / np.maximum(np.abs(w).max(axis=1, keepdims=True), 1e-30)
            out[k] = (_round_e4m3(w * sc) / sc).astype(np.float32)
            n += 1
    assert n == 4 * 10, n  # 2 + 6 + 2 blocks of TinyViT-5M
    return out


@pytest.mark.timeout(900)
def test_fp8_tinyvit_mask_iou_on_the_config_that_names_it():
    """BASELINE.json configs[4] asks for fp8 weights on the TinyViT encoder; north_star's tolerance for masks is IoU. CPU-only what-if
    through the oracles (TinyViT-5M at 1024 x 1024 -> prompt encoder + mask decoder, 8 point prompts and 2 boxes): the encoder's
    transformer-stage linears with (a) e4m3 weights, per-output-channel scales, and (b) additionally e4m3 inputs, per-token scales
    (what an fp8 x fp8 MFMA consumes), against the unquantised run. Bar for building the kernel: every prompt's mask IoU >= 0.99.

    Measured (seeded synthetic weights; printed below): weights only -> embedding error 7.4 % of its scale, IoU mean 0.969, min 0.928;
    weights + activations -> 8.0 %, IoU mean 0.967, min 0.927. Both miss the bar, and the encoder is not MFMA-rate bound to begin
    with (its GEMMs run K = 64 .. 320 at 330-585 TFLOP/s on f16, bounded by activation bytes) -- rejected
    for configs[4]; the assertions pin that the numbers still say so. (Caveat recorded in DESIGN.md: random decoders put many logits
    near the threshold, a trained SAM's logits are bimodal; the embedding error itself, 7-8 %, is the robust figure: two orders of magnitude above the f16 path's 6e-4.)"""
    cfg = synth.TINYVIT_5M
    enc_sd = {k: v.astype(np.float16).astype(np.float32) for k, v in synth.tinyvit_state_dict(cfg, seed=3).items()}
    dec_t = synth.sam_decoder_gguf_tensors(synth.sam_decoder_state_dict(5))
    params = oracle.tinyvit_params(cfg.img_size, cfg.layers())
    img = synth.images(1, 1024, 1024, seed=77)[0]
    mean = np.array([123.675, 116.28, 103.53], np.float32) / np.float32(255.0)
    std = np.array([58.395, 57.12, 57.375], np.float32) / np.float32(255.0)
    x = (img.astype(np.float32) / np.float32(255.0) - mean) / std

    def encode(sd, act_quant):
        t, conv_idx = synth.tinyvit_gguf_tensors(sd)
        lib = oracle.lib()
        lib.vo_set_linear_act_quant.argtypes = [C.c_int]
        lib.vo_set_linear_act_quant(int(act_quant))
        try:
            return oracle.tinyvit_encode(oracle.Model(t, conv_idx), params, x).reshape(64, 64, 256)
        finally:
            lib.vo_set_linear_act_quant(0)

    ref = encode(enc_sd, False)
    q = _quantise_tinyvit_linears(enc_sd)
    emb = {"weights e4m3": encode(q, False), "weights + activations e4m3": encode(q, True)}
    dec = oracle.Model(dec_t, [])
    prompts = [(200, 300), (512, 512), (900, 100), (64, 960), (700, 700), (333, 777), (20, 20), (1000, 1000), (100, 100, 600, 500), (400, 300, 900, 1000)]
    want = [oracle.sam_compute(dec, ref, 1024, 1024, p, return_all=True) for p in prompts]
    res = {}
    for name, e in emb.items():
        rel = float(np.abs(e - ref).mean() / np.abs(ref).mean())
        ious = []
        for p, (m_ref, _, _) in zip(prompts, want):
            m_q = oracle.sam_compute(dec, e, 1024, 1024, p)
            ious.append(_iou(m_q, m_ref))
        res[name] = (rel, float(np.mean(ious)), float(np.min(ious)))
        print(f"\nfp8 what-if, MobileSAM TinyViT-5M 1024x1024, {name}: embedding mean |err| / mean |ref| = {rel:.3f}, mask IoU mean {np.mean(ious):.4f} min {np.min(ious):.4f}")
    # the decision: below the IoU >= 0.99 bar in both forms -> fp8 is never the default for configs[4] (round 4 built the kernel as an opt-in to measure it)
    assert res["weights + activations e4m3"][2] < 0.99 and res["weights e4m3"][2] < 0.99, "e4m3 TinyViT met the IoU bar: revisit DESIGN.md section 8"
    assert res["weights + activations e4m3"][0] > res["weights e4m3"][0] > 0.005
