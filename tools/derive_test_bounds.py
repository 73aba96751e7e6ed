#!/usr/bin/env python3
"""measured.json (written by a `TTL_RECORD_BOUNDS=... pytest -m gpu` run on MI355X) -> tests/golden/bounds.json
(merged into the measurements already there; `--replace` to start over).

bound = measured x 1.3, rounded up to 3 significant digits; quantities that measure (almost) zero get a floor of 1e-6 so that a
last-bit change of a summation order cannot fail them.  tests/bounds.py applies them on top of each assertion's documented ceiling."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARGIN, FLOOR = 1.3, 1e-6


def up3(v):
    if v <= 0:
        return FLOOR
    e = math.floor(math.log10(v)) - 2
    return round(math.ceil(v / 10 ** e) * 10 ** e, 12)


def main():
    src = sys.argv[1]
    meas = json.load(open(src))
    dst0 = os.path.join(ROOT, "tests", "golden", "bounds.json")
    if "--replace" not in sys.argv[2:] and os.path.exists(dst0):
        # default: MERGE — keys measured in this recording run replace their old value, keys it did not touch (a partial run, e.g.
        # `pytest tests/test_gpu_dropin.py`) keep theirs; --replace starts from the recording alone
        old = json.load(open(dst0)).get("measured", {})
        meas = {**old, **meas}
    out = {"note": "tools/derive_test_bounds.py: bound = measured-on-MI355X x %.1f (3 significant digits, floor %g); "
                   "regenerate after any kernel change that moves a summation order" % (MARGIN, FLOOR),
           "margin": MARGIN, "measured": {k: meas[k] for k in sorted(meas)},
           # strict/ keys (fp32 build, tests/test_gpu_strict.py) measure fp32 summation-order noise, 1e-6 ... 1e-5 against documented
           # ceilings of 1e-5 ... 1e-4: x 3 and a floor of 5e-6, so that a harmless re-association does not fail them
           # (the same for the strict-build cases of other test families, ".../strict/...": tests/test_gpu_dropin.py's surface loop).
           # Fractions of elements ("frac_...") that measure ~0 get a floor of 1e-4: one element of 10^4 ... 10^5 is 1e-5 ... 1e-4
           "bounds": {k: max(max(up3(meas[k] * 3.0), 5e-6) if (k.startswith("strict/") or "/strict/" in k) else max(up3(meas[k] * MARGIN), FLOOR),
                             1e-4 if "/frac_" in k else 0.0) for k in sorted(meas)}}
    dst = os.path.join(ROOT, "tests", "golden", "bounds.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{len(meas)} bounds -> {dst}")


if __name__ == "__main__":
    main()
