"""Measured-and-margined bounds of the GPU parity tests.

Every floating-point distance a `-m gpu` test asserts on goes through ``check(key, measured, ceiling)``: the bound is
``tests/golden/bounds.json[key]`` = the value MEASURED on MI355X with the committed kernels x 1.3 (written by
``tools/derive_test_bounds.py`` from a recording run), so a fixture that sits 8 % under a hand-picked bound cannot flake and a 2x
regression cannot pass.  ``ceiling`` is the documented tolerance of that quantity (north_star 1e-3 for the fp16-operand build, the
bf16 price, ...): it always applies too, and is the only bound for a key that has never been recorded.

    TTL_RECORD_BOUNDS=/path/measured.json python -m pytest tests -m gpu      # record (asserts the ceilings only)
    python tools/derive_test_bounds.py /path/measured.json                   # -> tests/golden/bounds.json
"""
import json
import os

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bounds.json")
MARGIN = 1.3
_record = os.environ.get("TTL_RECORD_BOUNDS")
_bounds = None
_seen = {}


def _load():
    global _bounds
    if _bounds is None:
        try:
            with open(PATH) as f:
                _bounds = json.load(f)["bounds"]
        except (OSError, ValueError, KeyError):
            _bounds = {}
    return _bounds


def check(key, measured, ceiling):
    measured = float(measured)
    assert measured == measured, (key, "nan")
    assert measured < ceiling, (key, measured, "documented ceiling", ceiling)
    if _record:
        if os.path.exists(_record) and not _seen:
            try:
                with open(_record) as f:
                    _seen.update(json.load(f))
            except ValueError:
                pass
        _seen[key] = max(_seen.get(key, 0.0), measured)
        with open(_record, "w") as f:
            json.dump(_seen, f, indent=0, sort_keys=True)
        return
    b = _load().get(key)
    if b is not None:
        assert measured < b, (key, measured, f"measured-on-MI355X x {MARGIN}", b)
