"""CPU oracle for the KoeMorph hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of the reference's hot path
(audio window -> log-mel -> dual-stream cross-attention -> decoder -> smoothing) so
that the HIP kernels in ``koemorph_amd/csrc`` can be checked against it.

Rules ("Oracle"):
  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import anything from here -- always as the checker or as the
    reported CPU baseline, never as the thing measured or shipped;
  * nothing in ``koemorph_amd/`` imports it; the product path has no CPU fallback;
  * nothing here reads ``/root/reference`` at run time except ``gen_golden.py``, which
    runs only in the build container to (re)generate ``tests/golden/*.npz``.

Pinning status:
  * attention core (oracle.core)        PINNED  -- checked against the reference's own
    ``DualStreamCrossAttention`` imported from /root/reference (gen_golden.py), golden
    outputs committed under tests/golden/core_*.npz;
  * temporal smoothing (oracle.smoothing) restated from source text; trivially small;
    PARITY UNPINNED by a runnable reference (the wrapper model imports librosa);
  * mel front ends (oracle.mel)         PARITY UNPINNED -- librosa / torchaudio are not
    installed in this image and the reference ships no mel fixtures.  The restatement
    follows the published librosa 0.10 / torchaudio 2.x algorithms, and its filterbank
    and dB conversion are cross-checked against ``transformers.audio_utils`` (pure numpy);
  * ring buffers (oracle.buffers)       restated from source text, integer/index logic.
"""
