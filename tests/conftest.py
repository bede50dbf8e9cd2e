"""pytest configuration: the `gpu` marker and shared helpers.

`-m "not gpu"` covers the oracle against the golden fixtures, the host logic, and that the
C-ABI library loads and exports every declared symbol.  `-m gpu` tests are the parity tests
proper; they call the HIP path through the C-ABI and fail (not skip) if the library is
missing on a machine that has a GPU.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    rec = {k: z[k] for k in z.files}
    rec["config"] = json.loads(str(rec["config"]))
    return rec


def golden_case(name):
    """(config, params, (mel, short, emo), golden record) regenerated from the seeds."""
    from koemorph_amd import synth
    g = load_golden(name)
    c = g["config"]
    params = synth.make_core_params(c["seed"], c["d"], c["T"], 256, c["pstyle"])
    assert abs(synth.params_checksum(params) - float(g["params_checksum"])) <= 1e-6 * abs(float(g["params_checksum"])), \
        "synthetic parameter generator drifted from the one the fixtures were made with"
    inputs = synth.make_core_inputs(c["seed"], c["B"], c["t_in"], style=c["istyle"])
    return c, params, inputs, g


def full_loss_inputs(seed, B):
    """target, prev_pred, prev_target, landmark matrix of the full-KoeMorphLoss fixtures (oracle/gen_golden.py)."""
    import numpy as np
    from koemorph_amd import synth
    return (synth.uniform(seed * 3 + 1, (B, 52), 0.0, 1.0), synth.uniform(seed * 3 + 2, (B, 52), 0.0, 1.0),
            synth.uniform(seed * 3 + 3, (B, 52), 0.0, 1.0), (0.01 * synth.normal(seed * 3 + 4, (136, 52))).astype(np.float32))


def golden_masks(g):
    """The three dropout keep masks of a training-mode fixture ({"mel","emo","dec"} -> bool arrays), unpacked."""
    import numpy as np
    out = {}
    for k in ("mel", "emo", "dec"):
        shape = tuple(int(v) for v in g["maskshape/" + k])
        out[k] = np.unpackbits(g["mask/" + k])[:int(np.prod(shape))].reshape(shape).astype(bool)
    return out


def assert_grads_match(grads, g, tol):
    """grads vs a golden record holding grad/<key> (small tensors) or gradsample/<key> + gradnorm/<key>."""
    import numpy as np
    for k, v in grads.items():
        if "grad/" + k in g:
            ref = g["grad/" + k]
            np.testing.assert_allclose(v, ref, atol=1e-7 + tol * np.abs(ref).max(), rtol=tol, err_msg=k)
        else:
            ref = g["gradsample/" + k]
            np.testing.assert_allclose(v.ravel()[::97], ref, atol=1e-7 + tol * np.abs(ref).max(), rtol=tol, err_msg=k)
            n = np.sqrt(np.sum(v.astype(np.float64) ** 2))
            assert abs(n - float(g["gradnorm/" + k])) <= tol * float(g["gradnorm/" + k]) + 1e-9, k


CORE_CASES_D256 = [
    "core_d256_T256_H8_init", "core_d256_T256_H8_trained", "core_d256_T256_H8_randn",
    "core_d256_pad_T100", "core_d256_trunc_T300", "core_d256_rt_T255", "core_d256_pad_T1", "core_d256_trunc_T700",
]
CORE_CASES_OTHER = ["core_d512_T512_H8", "core_d512_T512_H16", "core_d64_T32_H4_small"]


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
