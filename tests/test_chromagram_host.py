"""CPU: host-side Chromagram / detect_key of the product package against the
reference-generated fixtures (these 12-number operations stay on the host)."""
import json
import os
import warnings

import numpy as np
import pytest

import chord_detection_amd as cd


def test_pack_and_key_against_reference_fixtures(golden_dir):
    with open(os.path.join(golden_dir, "constants.json")) as fh:
        cases = json.load(fh)["pack_key_cases"]
    for case in cases:
        c = cd.Chromagram(case["chroma"])
        assert repr(c) == case["pack"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert c.key() == case["key"]
            assert cd.detect_key(np.asarray(case["chroma"], dtype=float)) == case["key"]


def test_reference_key_detection_known_answers():
    cases = {
        "Cmaj": [100.0, 0, 0, 0, 100.0, 0, 0, 100.0, 0, 0, 0, 0],
        "Cmin": [50.0, 0, 50.0, 50.0, 0, 0, 0, 10.0, 0, 0, 0, 0],
        "G#maj": [0, 10.0, 0, 10.0, 0, 0, 0, 0, 10.0, 0, 10.0, 0],
    }
    for key, v in cases.items():
        assert cd.detect_key(np.asarray(v)) == key
    with pytest.raises(ValueError):
        cd.detect_key(np.zeros(7))


def test_chromagram_semantics():
    a, b = cd.Chromagram(), cd.Chromagram()
    a[0] = 1.0
    a["E"] = 2.0
    b["C"] = 3.0
    assert a["C"] == 1.0 and a[4] == 2.0 and len(a) == 12
    c = a + b
    assert c is a and a["C"] == 4.0          # in-place add returning self (chromagram.py:42-45)
    assert a["C♯"] == a["C#"]                # unicode sharp accepted on read (chromagram.py:21)
    a["C♯"] = 9.0                            # ... but writes under it go nowhere (chromagram.py:29)
    assert a["C#"] == 0.0 and len(a) == 13
    with pytest.raises(ValueError):
        a[1.5]
    assert repr(cd.Chromagram([5, 0, 0, 0, 2, 0, 0, 1, 0, 0, 0, 0])) == "500020010000"
