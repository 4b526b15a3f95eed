"""CPU oracle for the STFT -> (ESACF | Harmonic-Energy) -> chromagram hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy (float64) restatement of the
reference algorithm (sevagh/chord-detection, Python) and exists to *check* the
HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product package (``chord-detection_amd``)
never imports it and has no CPU fallback.

Parity pinning (see DESIGN.md "Oracle"):
  * every function here cites the reference file:line it follows;
  * ``tests/golden/make_golden.py`` (run in the authoring container, where
    /root/reference is mounted) imports the *reference's own code* and stores
    its outputs as fixtures under ``tests/golden/``; ``tests/test_oracle_*.py``
    check this oracle against those fixtures (provenance label ``ref-code``);
  * two third-party call sites of the reference are NOT under /root/reference
    and are not installed anywhere we can reach: ``librosa.effects.time_stretch``
    (esacf.py:121) and ``peakutils.indexes/interpolate`` (esacf.py:56-62).
    They are restated from their published algorithm in ``oracle/thirdparty.py``;
    no reference test constrains them => those two stages are
    **parity unpinned** (fixtures labelled ``ref-code+stub``).  The peak fit is
    additionally cross-checked against the real ``scipy.optimize.curve_fit``
    (MINPACK), which is what peakutils calls.
"""
