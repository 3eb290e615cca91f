"""Gain-map application (avifRGBImageApplyGainMap, reference src/gainmap.c:73-315).

CPU: the oracle's restatement (oracle/gainmap_oracle.c: transfer functions, primaries matrices, float pixel accessors, the
tone-mapping loop) against avifRGBImageApplyGainMap of the reference compiled from its own sources -- same libm, same
flags: every output byte, the result code and the CLLI values are IDENTICAL."""
import ctypes as C

import numpy as np
import pytest

import gainmap_cases as G
import oracle_lib
from libavif_amd import abi

libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]


def run(fn, c, extra):
    base = G.make_base(c)
    gm, keep = G.make_gain_map(c)
    out = G.make_output(c)
    clli = abi.avifContentLightLevelInformationBox(0xFFFF, 0xFFFF)
    res = fn(base.struct, c.base_primaries, c.base_tc, C.byref(gm), c.headroom, c.out_primaries, c.out_tc, out.struct, C.byref(clli), extra)
    pixels = G.output_bytes(out) if (res == 0 and out.struct.pixels) else None
    if out.struct.pixels:
        libc.free(C.cast(out.struct.pixels, C.c_void_p))
    return res, pixels, (clli.maxCLL, clli.maxPALL)


@pytest.mark.skipif(oracle_lib.ref() is None, reason="oracle/_ref/libavif_ref.so not built (needs /root/reference)")
def test_oracle_equals_reference():
    ref, o = oracle_lib.ref(), oracle_lib.oracle()
    diag = abi.avifDiagnostics()
    bad = []
    cases = G.cases(500, seed=1)
    for c in cases:
        ra, pa, ca = run(ref.avifRGBImageApplyGainMap, c, C.byref(diag))
        rb, pb, cb = run(o.oracleRGBImageApplyGainMap, c, 0)  # the from-source reference build has no libyuv
        if ra != rb or (ra == 0 and (not np.array_equal(pa, pb) or ca != cb)):
            bad.append(f"{c.ident()}: results {ra}/{rb} clli {ca}/{cb}" + ("" if ra or rb or pa is None or np.array_equal(pa, pb) else
                                                                       f" {int((pa != pb).sum())} bytes differ"))
    assert not bad, f"{len(bad)} of {len(cases)} cases differ:\n" + "\n".join(bad[:20])


@pytest.mark.skipif(oracle_lib.ref() is None, reason="oracle/_ref/libavif_ref.so not built (needs /root/reference)")
def test_primaries_matrices_equal_reference():
    ref, o = oracle_lib.ref(), oracle_lib.oracle()
    for a in G.PRIMARIES + [10]:
        for b in G.PRIMARIES + [10]:
            ma, mb = (C.c_double * 9)(), (C.c_double * 9)()
            ra, rb = ref.avifColorPrimariesComputeRGBToRGBMatrix(a, b, C.byref(ma)), o.oracleColorPrimariesComputeRGBToRGBMatrix(a, b, C.byref(mb))
            assert bool(ra) == bool(rb), (a, b)
            if ra:
                assert list(ma) == list(mb), (a, b)


def test_argument_errors():
    o = oracle_lib.oracle()
    c = G.GainMapCase(8, 8)
    assert run(o.oracleRGBImageApplyGainMap, G.GainMapCase(8, 8, headroom=-1.0), 0)[0] == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert run(o.oracleRGBImageApplyGainMap, G.GainMapCase(8, 8, gm_gamma=((0, 1), (1, 1), (1, 1))), 0)[0] == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert run(o.oracleRGBImageApplyGainMap, G.GainMapCase(8, 8, gm_min=((2, 1), (0, 1), (0, 1)), gm_max=((1, 1), (1, 1), (1, 1))), 0)[0] == abi.AVIF_RESULT_INVALID_ARGUMENT
    assert run(o.oracleRGBImageApplyGainMap, c, 0)[0] == 0
