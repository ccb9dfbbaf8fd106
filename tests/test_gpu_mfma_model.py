"""The accumulation-error MODEL the fp16 matrix-core bound pass rests on (csrc/ndbhip_common.h (4)):

    | D - (C + sum_{k<16} a_k b_k) |  <=  NDB_MFMA_THETA * ( |C| + sum_k |a_k b_k| ),   NDB_MFMA_THETA = 34 * 2^-24

The ISA text gives no rounding rule for v_mfma_f32_32x32x16_f16, so the model is checked here on the part the
suite runs on, through the library's own probe entry point (ndbhip_mfma_probe): directed and adversarial inputs
against exact float64 arithmetic (a product of two halfs is exact in float64; 17 float64 additions err by less
than 2^-49 of sum |terms|).  A part that breaks the model — or flushes fp16 subnormal inputs beyond what
NDB_S16_SPLIT allows — fails the GPU suite instead of silently losing neighbours.
tools/mfma_probe.hip is the long form of the same probe (profiles/r02_mfma_probe.txt)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U = 2.0 ** -24
THETA = 34 * U


def _run(A16, B16, C, chain):
    import torch
    from neurondb_amd import _lib
    _lib.ensure_init()
    nt = A16.shape[0]
    dA = torch.from_numpy(A16.view(np.int16).copy()).cuda()
    dB = torch.from_numpy(B16.view(np.int16).copy()).cuda()
    dC = torch.from_numpy(C.copy()).cuda()
    dD = torch.zeros_like(dC)
    _lib.check(_lib.lib().ndbhip_mfma_probe(dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), dD.data_ptr(), nt, chain))
    _lib.check(_lib.lib().ndbhip_synchronize())
    return dD.cpu().numpy()


def _exact(A16, B16, C, chain):
    a = A16.astype(np.float64)                       # [nt, 32, 16]
    b = B16.astype(np.float64)                       # [nt, 16, 32]
    dot = np.einsum("tik,tkj->tij", a, b)
    mag = np.einsum("tik,tkj->tij", np.abs(a), np.abs(b))
    return C.astype(np.float64) + chain * dot, np.abs(C.astype(np.float64)) + chain * mag


CASES = ["uniform", "spread", "c_huge", "cancel", "positive", "big_plus_dust", "subnormal_inputs", "tiny_products"]


@pytest.mark.parametrize("kind", CASES)
@pytest.mark.parametrize("chain", [1, 12])
def test_mfma_f16_accumulation_stays_inside_the_model(kind, chain):
    rng = np.random.default_rng(CASES.index(kind) * 10 + chain)
    nt = 512
    A = rng.uniform(-1, 1, (nt, 32, 16))
    B = rng.uniform(-1, 1, (nt, 16, 32))
    C = rng.uniform(-1, 1, (nt, 32, 32))
    if kind == "spread":
        A *= 2.0 ** rng.integers(-12, 13, A.shape)
        B *= 2.0 ** rng.integers(-12, 13, B.shape)
        C *= 2.0 ** rng.integers(-12, 13, C.shape)
    elif kind == "c_huge":
        C *= 2.0 ** 20
    elif kind == "cancel":
        A[:, :, 1:14:2] = -A[:, :, 0:13:2]
        B[:, 1:14:2, :] = B[:, 0:13:2, :]
        C[:] = 0
    elif kind == "positive":
        A, B, C = np.abs(A), np.abs(B), np.abs(C)
    elif kind == "big_plus_dust":
        A *= 2.0 ** -10
        idx = rng.integers(0, 16, (nt, 32))
        np.put_along_axis(A, idx[:, :, None], 700.0, axis=2)
    elif kind == "subnormal_inputs":
        A *= 2.0 ** -15                               # every a is an fp16 subnormal
        B *= 2.0 ** 6
        C *= 2.0 ** -12
    elif kind == "tiny_products":
        A *= 2.0 ** -13
        B *= 2.0 ** -13
        C[:] = 0
    if chain > 1:
        C[:] = 0                                      # the bound pass starts every block from zero
    A16, B16, C32 = A.astype(np.float16), B.astype(np.float16), C.astype(np.float32)
    got = _run(A16, B16, C32, chain).astype(np.float64)
    exact, mag = _exact(A16, B16, C32, chain)
    err = np.abs(got - exact)
    # chain of n instructions: every one adds at most THETA (|C_j| + its products), |C_j| <= the sum so far
    bound = chain * 1.01 * THETA * mag + 2.0 ** -149
    worst = float(np.max(err / np.maximum(mag, 2.0 ** -140)))
    assert np.all(err <= bound), (kind, chain, worst / U)
    # margin actually observed, for the record (5.3 u worst case on gfx950; the model allows 34 u per instruction)
    print(f"{kind} chain {chain}: max err / (|C| + sum|ab|) = {worst / U:.2f} u, model {34 * chain} u")


def test_fp16_subnormal_inputs_are_not_flushed():
    """NDB_S16_SPLIT (csrc/ndbhip_common.h (3)) assumes the lo plane's subnormal halfs count at face value."""
    A = np.zeros((1, 32, 16), np.float16)
    B = np.zeros((1, 16, 32), np.float16)
    C = np.zeros((1, 32, 32), np.float32)
    A[0, 0, 0] = np.float16(2.0 ** -24)               # smallest subnormal
    B[0, 0, 0] = np.float16(1.0)
    A[0, 1, 3] = np.float16(2.0 ** -15)
    B[0, 3, 1] = np.float16(3.0)
    got = _run(A, B, C, 1)
    assert got[0, 0, 0] == np.float32(2.0 ** -24)
    assert got[0, 1, 1] == np.float32(3.0 * 2.0 ** -15)


def _run_f32(A, B, C):
    import torch
    from neurondb_amd import _lib
    _lib.ensure_init()
    dA, dB, dC = (torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda() for x in (A, B, C))
    dD = torch.zeros_like(dC)
    _lib.check(_lib.lib().ndbhip_mfma_probe_f32(dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), dD.data_ptr(), A.shape[0]))
    _lib.check(_lib.lib().ndbhip_synchronize())
    return dD.cpu().numpy()


@pytest.mark.parametrize("kind", ["uniform", "spread", "cancel", "screen_shape"])
def test_mfma_f32_screen_instruction_stays_inside_its_model(kind):
    """Pass 0 of the centred sweep (csrc/ndbhip_screen16c.h) lets v_mfma_f32_32x32x2_f32 evaluate the exclusion test and
    allows each of its two steps an error of 2 u of |C| + |a0 b0| + |a1 b1| (twice what a correctly rounded fma chain
    costs): 4 u in all.  Checked against float64 (products of two floats are exact there)."""
    rng = np.random.default_rng(["uniform", "spread", "cancel", "screen_shape"].index(kind))
    nt = 512
    A = rng.uniform(-1, 1, (nt, 32, 2))
    B = rng.uniform(-1, 1, (nt, 2, 32))
    C = rng.uniform(-1, 1, (nt, 32, 32))
    if kind == "spread":
        A *= 2.0 ** rng.integers(-30, 31, A.shape)
        B *= 2.0 ** rng.integers(-30, 31, B.shape)
        C *= 2.0 ** rng.integers(-30, 31, C.shape)
    elif kind == "cancel":
        C = -(A[:, :, 0:1] * B[:, 0:1, :] + A[:, :, 1:2] * B[:, 1:2, :]) * (1 + rng.uniform(-1e-6, 1e-6, (nt, 32, 32)))
    elif kind == "screen_shape":
        # what the sweep feeds it: C = t1 P, a = (-v, -u), b = (w, s) with the terms nearly cancelling
        A = -np.abs(A) * 2.0 ** rng.integers(0, 40, A.shape)
        B = np.abs(B) * 2.0 ** rng.integers(-20, 21, B.shape)
        C = (np.abs(A[:, :, 0:1] * B[:, 0:1, :]) + np.abs(A[:, :, 1:2] * B[:, 1:2, :])) * rng.uniform(0.99, 1.01, (nt, 32, 32))
    A, B, C = A.astype(np.float32), B.astype(np.float32), C.astype(np.float32)
    got = _run_f32(A, B, C).astype(np.float64)
    a, b, c = A.astype(np.float64), B.astype(np.float64), C.astype(np.float64)
    exact = c + np.einsum("tik,tkj->tij", a, b)
    mag = np.abs(c) + np.einsum("tik,tkj->tij", np.abs(a), np.abs(b))
    err = np.abs(got - exact)
    worst = float(np.max(err / np.maximum(mag, 2.0 ** -120)))
    assert np.all(err <= 4 * U * mag + 2.0 ** -149), (kind, worst / U)
    print(f"f32 {kind}: max err / (|C| + sum|ab|) = {worst / U:.2f} u, allowed 4 u")


def test_mfma_f32_screen_instruction_and_infinities():
    """The flags pass 0 plants: w = +inf (a row the tile does not have: never emitted), w = -inf or u = -inf (always
    emitted); an integer maximum over the results' bit patterns must see the first as negative, the others as not."""
    A = np.zeros((1, 32, 2), np.float32)
    B = np.zeros((1, 2, 32), np.float32)
    C = np.full((1, 32, 32), 3.5, np.float32)
    A[0, :, 0] = -0.75                                 # -v
    A[0, :, 1] = -2.0                                  # -u
    A[0, 5, 1] = np.inf                                # u = -inf: member 5 always emits
    A[0, 6, 1] = -np.inf                               # u = +inf: member 6 never does
    B[0, 0, :] = 1.0                                   # w
    B[0, 1, :] = 1.0                                   # s
    B[0, 0, 3] = np.inf                                # row 3 is not there
    B[0, 0, 4] = -np.inf                               # row 4's norm is not a finite fp32
    got = _run_f32(A, B, C)[0]
    bits = got.view(np.int32)
    assert got[0, 0] == np.float32(3.5 - 0.75 - 2.0)
    assert np.all(got[:, 3][np.arange(32) != 5] == -np.inf) and np.all(bits[:, 3][np.arange(32) != 5] < 0)
    assert np.all(got[:, 4][np.arange(32) != 6] == np.inf)
    assert np.all(got[5, :][np.arange(32) != 3] == np.inf) and np.all(got[6, :][np.arange(32) != 4] == -np.inf)
    assert np.isnan(got[5, 3]) and np.isnan(got[6, 4])      # either verdict is fine there: pass 1 throws both out
