"""The opt-in e4m3 GEMM (csrc/kernels_gemm_fp8.hip; BASELINE.json configs[4] "fp8 GGUF weights on CDNA4 fp8 MFMA"; the reference has no fp8 type):
* CPU: the host quantizer against torch.float8_e4m3fn (the OCP format gfx950 implements), bit for bit, incl. subnormals, ties, saturation;
* GPU: the device quantizer against the host one, and the GEMM on v_mfma_scale_f32_16x16x128_f8f6f4 against an f64 product of the SAME quantized
  operands -- the kernel has to be exact for what it is given; whether e4m3 is accurate enough for a model is tests/test_fp8_decision.py's question."""
import ctypes as C

import numpy as np
import pytest
import torch

from visioncpp_amd import _lib as L


def host_quantize(w, Kp):
    w = np.ascontiguousarray(w, np.float32)
    N, K = w.shape
    q, s = np.zeros((N, Kp), np.uint8), np.zeros(N, np.float32)
    L.vx_check(L.get_lib().vx_quantize_rows_e4m3_host(w.ctypes.data, N, K, Kp, q.ctypes.data, s.ctypes.data))
    return q, s


def dequant(q):  # e4m3 bytes -> f64 through torch's table
    return torch.from_numpy(q.copy()).view(torch.float8_e4m3fn).to(torch.float64).numpy()


def test_host_quantizer_is_ocp_e4m3_round_to_nearest_even():
    rng = np.random.default_rng(0)
    K = 1024
    rows = [rng.standard_normal(K) * 10.0 ** e for e in (-3, -1, 0, 1)]
    rows.append(np.linspace(-448, 448, K))                                   # every binade, the saturation edge
    rows.append(np.concatenate([np.arange(0, 64) * 2.0 ** -10, rng.uniform(-0.02, 0.02, K - 64)]))  # subnormals (units of 2^-9) and their ties
    w = np.stack(rows).astype(np.float32)
    w[:, 0] = 448.0                                                           # absmax 448 -> scale exactly 1: the bytes are e4m3(w) itself
    w = np.clip(w, -448, 448)
    q, s = host_quantize(w, K)
    np.testing.assert_array_equal(s, np.ones(len(rows), np.float32))
    want = torch.from_numpy(w).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    np.testing.assert_array_equal(q, want)
    # scaled rows: value = byte * scale within half an e4m3 step of the input
    w2 = (rng.standard_normal((5, 320)) * np.array([[1e-3], [0.1], [1], [30], [5e3]])).astype(np.float32)
    q2, s2 = host_quantize(w2, 384)
    np.testing.assert_allclose(s2, np.abs(w2).max(1) / 448.0, rtol=1e-6)
    assert (q2[:, 320:] == 0).all()
    back = dequant(q2)[:, :320] * s2[:, None]
    assert np.abs(back - w2).max(1).max() <= (np.abs(w2).max(1) / 448.0 * 16).max()  # one step at the top binade is 32 / 448 of absmax, half of it the bound
    assert (np.abs(back - w2) <= np.maximum(np.abs(w2) * 2.0 ** -4, s2[:, None] * 2.0 ** -10) + 1e-12).all()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,act,bias,n_valid,resid", [(128, 128, 128, 0, False, 128, False), (300, 256, 320, 1, True, 256, False), (4096 + 37, 1280, 320, 1, True, 1280, False),
                                                          (2048, 384, 1280, 0, True, 320, True), (64, 128, 8, 0, False, 64, True)])
def test_fp8_gemm_is_exact_for_its_operands(M, N, K, act, bias, n_valid, resid):
    from gpu_util import dev, empty, release, sync
    from oracle import oracle

    api = L.get_lib()
    assert api.vx_device_count() > 0
    L.vx_check(api.vx_set_device(0))
    rng = np.random.default_rng(M + N + K)
    Kp = (K + 127) // 128 * 128
    x = (rng.standard_normal((M, K)) * rng.uniform(0.2, 3.0, (M, 1))).astype(np.float16)
    x[min(5, M - 1)] = 0                                                      # an all-zero token: scale 1, bytes 0
    w = (rng.standard_normal((N, K)) / np.sqrt(K) * rng.uniform(0.5, 2.0, (N, 1))).astype(np.float32)
    b = (0.1 * rng.standard_normal(N)).astype(np.float32)
    wq, ws = host_quantize(w, Kp)
    xd, qd, sd = dev(x), empty(M * Kp), empty(M * 4)
    L.vx_check(api.vx_quantize_rows_e4m3(xd.ptr, K, qd.ptr, sd.ptr, M, K, Kp, None))
    sync()
    xq, xs = qd.to_numpy(np.uint8, (M, Kp)), sd.to_numpy(np.float32, (M,))
    hq, hs = host_quantize(x.astype(np.float32), Kp)                          # the device quantizer = the host one (f32 division vs multiply by 1 / s: 1 ulp of the scale)
    np.testing.assert_allclose(xs, hs, rtol=2e-7)
    assert (xq != hq).mean() < 2e-3                                           # (a value on a rounding tie may flip with that ulp)
    r = (rng.standard_normal((M, n_valid))).astype(np.float16)
    out = dev(np.full((M, n_valid), 7.0, np.float16))  # columns >= n_valid do not exist in the output: ldo = n_valid
    a = L.GemmFp8Args()
    a.A, a.a_scale, a.W, a.w_scale = qd.ptr, sd.ptr, dev(wq).ptr, dev(ws).ptr
    a.bias = dev(b).ptr if bias else None
    a.res = dev(r).ptr if resid else None
    a.M, a.N, a.Kp, a.n_valid, a.out, a.ldo, a.act = M, N, Kp, n_valid, out.ptr, n_valid, act
    L.vx_check(api.vx_gemm_fp8(C.byref(a), None))
    sync()
    got = out.to_numpy(np.float16, (M, n_valid)).astype(np.float64)
    ref = (dequant(xq) @ dequant(wq).T) * xs[:, None].astype(np.float64) * ws[None, :].astype(np.float64)
    if bias:
        ref = ref + b
    if act:
        ref = oracle.gelu(ref.astype(np.float32), oracle.GELU_TANH_F32).astype(np.float64)
    ref = ref[:, :n_valid]
    if resid:
        ref = np.float64(np.float16(ref)) + r.astype(np.float64)  # the kernel rounds the product to f16, then adds the f16 residual
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())      # f32 accumulation + one f16 rounding of the result
    release()


@pytest.mark.gpu
def test_fp8_gemm_argument_errors():
    api = L.get_lib()
    a = L.GemmFp8Args()
    assert api.vx_gemm_fp8(C.byref(a), None) == 0 and b"null operand" in api.vx_last_error()
    assert api.vx_gemm_fp8_supported(1280, 320) == 1 and api.vx_gemm_fp8_supported(320, 1280) == 0  # N must be a multiple of 128
