"""Model-based random test of the field wrappers' metadata (R: fields.py:468-568 `_NewMetadataField.metadata`, :645-662
`new_field_with_metadata`, :131-144 `clone`): chains of re-labelled fields over a list-of-dicts field, with and without a MARS
namespace, queried every way the filters query them.

The model is the reference's rule stated once: an override is consulted before the wrapped field, outermost wrapper first
(R: fields.py:532-546); `metadata()` lists the keys of the innermost field only (:508-509 — hence the reference's own xfails at
tests/test_fields.py:45-46); `metadata(namespace="mars")` is the innermost field's namespace with overrides applied only to keys
already in it (:523-530); a missing key is a KeyError unless a default is given.
"""

from __future__ import annotations

import numpy as np
import pytest

from anemoi_transform_amd.fields import MARS_KEYS, ArrayField, DerivedField, new_field_with_metadata

KEYS = ["param", "levelist", "step", "date", "time", "number", "units", "valid_datetime", "shortName", "level", "custom"]
VALUES = ["t", "q", 850, 500, 0, 6, 20200101, 1200, "K", "m", "2020-01-01T00:00:00Z", None, 3.5]


@pytest.mark.parametrize("seed", range(200))
def test_metadata_of_a_chain_of_wrappers(seed):
    rng = np.random.default_rng(30_000 + seed)

    def some(n):
        picked = rng.choice(len(KEYS), size=n, replace=False)
        return {KEYS[i]: VALUES[int(rng.integers(0, len(VALUES)))] for i in picked}

    base = some(int(rng.integers(1, 8)))
    mars = bool(rng.random() < 0.5)
    field = ArrayField(np.zeros(4), base, np.zeros(4), np.zeros(4), mars=mars)
    layers = []
    for _ in range(int(rng.integers(0, 5))):
        over = some(int(rng.integers(0, 4)))
        layers.append(over)
        how = rng.integers(0, 3)
        field = (new_field_with_metadata(field, **over) if how == 0 else field.clone(**over) if how == 1 else DerivedField(field, metadata=over))

    def model(key):
        for over in reversed(layers):
            if key in over:
                return True, over[key]
        return (True, base[key]) if key in base else (False, None)

    for key in KEYS + ["absent"]:
        found, want = model(key)
        if found:
            assert field.metadata(key) == want and field.metadata()[key] == want and field.metadata().get(key, "dflt") == want
            assert key in field.metadata()
        else:
            with pytest.raises(KeyError):
                field.metadata(key)
            with pytest.raises(KeyError):
                field.metadata()[key]
            assert field.metadata(key, default="dflt") == "dflt" and field.metadata().get(key, "dflt") == "dflt" and field.metadata().get(key) is None
            assert key not in field.metadata()
    pair = [KEYS[i] for i in rng.choice(len(KEYS), size=2, replace=False)]
    if all(model(k)[0] for k in pair):
        assert field.metadata(*pair) == tuple(model(k)[1] for k in pair)
    assert list(field.metadata().keys()) == list(base.keys())  # the innermost field's keys, whatever the wrappers add
    want_ns = {k: model(k)[1] for k in base if k in MARS_KEYS} if mars else {}
    assert field.metadata(namespace="mars") == want_ns
    assert field.metadata(namespace="geography") == {}  # (only the MARS namespace is modelled: R: tests/conftest.py:27-38)
