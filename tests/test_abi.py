"""The C-ABI library loads on a CPU-only machine and exports every symbol include/koemorph.h
declares (no compute calls here)."""
import ctypes
import os
import re

from koemorph_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "koemorph.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(km_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in koemorph.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in koemorph_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.km_abi_version() == _lib.KM_ABI_VERSION


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.KMMelConfig) == 16 * 4
    assert ctypes.sizeof(_lib.KMConfig) == 9 * 4 + 16 * 4


def test_create_validates_like_the_reference_constructors():
    import pytest
    from koemorph_amd.engine import Engine, MelConfig
    from koemorph_amd._lib import KoeMorphError
    with pytest.raises(KoeMorphError, match="divisible by num_heads"):       # nn.MultiheadAttention assertion
        Engine(d_model=250, num_heads=8)
    with pytest.raises(KoeMorphError, match="hop_length"):                    # stft.py:78-81 ValueError
        Engine(mel=MelConfig(hop_length=0))
    e = Engine()
    assert e.mel_num_frames(136448) == 257 and e.mel_num_frames(136000) == 256
    assert e.sequence_num_outputs(136448 + 533 * 9, 1) == 10
    assert e.sequence_num_outputs(1000, 1) == 1
    e.close()
