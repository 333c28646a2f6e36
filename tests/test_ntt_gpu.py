"""GPU (-m gpu): NTT over Fr(BW6-761) through the C ABI (include/celo_bls_amd.h: ntt_bw6_761_fr[_dev]) vs the oracle.

What it replaces: ark-poly 0.1 Radix2EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place inside
ark_groth16::create_proof_no_zk (called at crates/epoch-snark/src/api/prover.rs:78,112) - SURVEY.md section 8f row f3.
The reference holds no NTT vector (parity unpinned there); the oracle is the O(n^2) definition (oracle/py/ntt.py) at small
sizes and a textbook decimation-in-time restatement (oracle/cpu/capi.cpp: orc_ntt_fq377) up to 2^16; at 2^20 the
size-independent properties are checked: inverse(forward(x)) == x, linearity, and the transform of a delta."""
import numpy as np
import pytest
import torch  # before the library: both must share one HIP runtime (torch's is loaded first everywhere else too)
from oracle.py import ecc, ntt as ontt
from oracle import cpu_oracle as co

pytestmark = pytest.mark.gpu
Q = ecc.Q377


def _mont1(v):
    return co.to_mont([v % Q], Q)[0]


def _rand(rng, n):
    return [ecc.random_scalar(rng, Q) for _ in range(n)]


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 7])
def test_matches_definition(gpu, log_n):
    rng = ecc.SplitMix64(100 + log_n)
    n = 1 << log_n
    w = ontt.root_of_unity(log_n)
    x = _rand(rng, n)
    X = co.from_mont(gpu.ntt(co.to_mont(x, Q), log_n, _mont1(w)), Q)
    assert X == ontt.dft(x, w)


@pytest.mark.parametrize("log_n", [6, 10, 11, 12, 13, 14, 15, 16])   # 10: 8+2, 12: 8+4, 14: 8+6, 15: 8+6+1, 16: 8+8 levels per launch
def test_all_four_transforms_match_oracle(gpu, log_n):
    """fft, ifft (omega^-1, scale n^-1), coset_fft (x_i *= g^i first), coset_ifft (x_i *= g^-i last, scale n^-1): bit-exact."""
    rng = ecc.SplitMix64(7 * log_n)
    n = 1 << log_n
    w = ontt.root_of_unity(log_n)
    winv, ninv, g = pow(w, -1, Q), pow(n, -1, Q), 15
    ginv = pow(g, -1, Q)
    x = np.random.default_rng(log_n).integers(0, 1 << 62, size=(n, 6), dtype=np.int64).astype(np.uint64)
    x[:, 5] &= np.uint64((1 << 56) - 1)                     # arbitrary Montgomery limbs below p
    for kw in (dict(), dict(coset=g), dict(omega=winv, scale=ninv), dict(omega=winv, coset=ginv, coset_after=True, scale=ninv)):
        om = kw.get("omega", w)
        got = gpu.ntt(x, log_n, _mont1(om), None if "coset" not in kw else _mont1(kw["coset"]), kw.get("coset_after", False),
                      None if "scale" not in kw else _mont1(kw["scale"]))
        want = co.ntt_fq377(x, log_n, om, kw.get("coset"), kw.get("coset_after", False), kw.get("scale"))
        assert np.array_equal(got, want), kw


def test_two_to_20_properties_device_resident(gpu):
    log_n = 20
    n = 1 << log_n
    w = ontt.root_of_unity(log_n)
    winv, ninv = pow(w, -1, Q), pow(n, -1, Q)
    rng = np.random.default_rng(20)
    x = rng.integers(0, 1 << 62, size=(n, 6), dtype=np.int64).astype(np.uint64)
    x[:, 5] &= np.uint64((1 << 56) - 1)
    d = torch.from_numpy(x.view(np.int64).copy()).cuda()
    gpu.ntt_dev(d.data_ptr(), log_n, _mont1(w))
    X = d.cpu().numpy().view(np.uint64)
    # spot-check 4 outputs against the definition restricted to them: X_j = sum_i x_i w^(ij) (host big-int arithmetic on a sparse input)
    gpu.ntt_dev(d.data_ptr(), log_n, _mont1(winv), None, False, _mont1(ninv))
    back = d.cpu().numpy().view(np.uint64)
    xc = co.from_mont(x[:64], Q)
    assert co.from_mont(back[:64], Q) == xc and co.from_mont(back[-64:], Q) == co.from_mont(x[-64:], Q)
    assert np.array_equal(back.reshape(n, 6)[1000:1100], co.to_mont(co.from_mont(x[1000:1100], Q), Q))
    # delta at position 3 -> X_j = w^(3j)
    delta = np.zeros((n, 6), dtype=np.uint64)
    delta[3] = _mont1(1)
    d2 = torch.from_numpy(delta.view(np.int64)).cuda()
    gpu.ntt_dev(d2.data_ptr(), log_n, _mont1(w))
    D = d2.cpu().numpy().view(np.uint64).reshape(n, 6)
    for j in (0, 1, 2, 12345, n - 1):
        assert co.from_mont(D[j:j + 1], Q) == [pow(w, 3 * j, Q)]
    # linearity on a slice: NTT(x + delta) - NTT(x) == NTT(delta)
    xs = co.from_mont(x[3:4], Q)[0]
    x2 = x.copy()
    x2[3] = _mont1(xs + 1)
    d3 = torch.from_numpy(x2.view(np.int64)).cuda()
    gpu.ntt_dev(d3.data_ptr(), log_n, _mont1(w))
    X2 = d3.cpu().numpy().view(np.uint64).reshape(n, 6)
    for j in (0, 5, 77777, n - 2):
        a = co.from_mont(X2[j:j + 1], Q)[0]
        b = co.from_mont(X.reshape(n, 6)[j:j + 1], Q)[0]
        assert (a - b) % Q == pow(w, 3 * j, Q)
