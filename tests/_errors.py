"""Per-element error accounting for the gradient parity tests.

``assert_allclose(g / s, g_o / s, rtol=1e-5, atol=1e-5)`` with s = max|g_o| bounds every element's error by 1e-5 of the
tensor's LARGEST element: an element a thousand times smaller may be off by 1 %.  A sum's honest per-element yardstick is
the sum of the MAGNITUDES of its terms, ``mag`` — the quantity floating-point summation error is proportional to,
whatever the cancellation: |computed - exact| <= n * eps * mag for any order of n terms.  The tests get ``mag`` from the
oracle itself, run on the absolute values of the operands (every term of a gradient element is a product of operand
entries and a positive 1/count).  Two numbers are reported and asserted:
  * max over elements of |got - ref| / mag            (bound: 1e-5, the north-star tolerance applied per element)
  * max ULP distance over the elements that are not dominated by cancellation (|ref| >= mag / 4)
"""
import numpy as np


def ulp_distance(a, b):
    """distance in units in the last place between two float32 arrays (same sign assumed where it matters)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2147483648) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2147483648) - ib, ib)
    return np.abs(ia - ib)


def error_report(got, ref, mag):
    got, ref, mag = (np.asarray(x, np.float64) for x in (got, ref, mag))
    err = np.abs(got - ref)
    live = mag > 0
    rel = np.zeros_like(err)
    rel[live] = err[live] / mag[live]
    well = live & (np.abs(ref) >= 0.25 * mag)
    ulps = ulp_distance(got.astype(np.float32), ref.astype(np.float32))
    return {"max_err_over_mag": float(rel.max()) if rel.size else 0.0,
            "max_abs_err": float(err.max()) if err.size else 0.0,
            "max_ulp_well_conditioned": int(ulps[well].max()) if well.any() else 0,
            "elements": int(err.size), "well_conditioned": int(well.sum()),
            "dead_elements_nonzero": int((err[~live] != 0).sum())}


def assert_per_element(got, ref, mag, what, bound=1e-5, max_ulp=256):
    r = error_report(got, ref, mag)
    print("%s: max |err|/mag %.2e, max |err| %.2e, max ULP (|ref| >= mag/4: %d of %d elements) %d"
          % (what, r["max_err_over_mag"], r["max_abs_err"], r["well_conditioned"], r["elements"], r["max_ulp_well_conditioned"]))
    assert r["dead_elements_nonzero"] == 0, "%s: elements with no contributing term must be exactly 0" % what
    assert r["max_err_over_mag"] <= bound, "%s: per-element error %.3e of the magnitude sum exceeds %.1e" % (what, r["max_err_over_mag"], bound)
    assert r["max_ulp_well_conditioned"] <= max_ulp, "%s: %d ULP on a well-conditioned element" % (what, r["max_ulp_well_conditioned"])
    return r
