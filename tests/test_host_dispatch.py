"""Host logic without a GPU: the product's plan layer (libavif_amd/csrc/plan.cpp, seen through the host-only
avifhipExplainYUVToRGB / avifhipExplainRGBToYUV) and the oracle (oracle/libyuv_oracle.c + reformat_oracle.c) are two
independent restatements of libavif's dispatch -- which libyuv function or which built-in loop serves a configuration,
who writes the alpha byte, where a pending alpha multiply happens (src/reformat.c:1445-1593, :242-272,
src/reformat_libyuv.c:270-381, :544-1108).  They must agree on every configuration of the sweeps."""
import ctypes as C
from dataclasses import replace

import harness as H
import oracle_lib
from libavif_amd import abi, native

SIZES = [(37, 21), (2, 2), (127, 10), (256, 16)]


def _mul_mode(c):
    """src/reformat.c:1662-1677"""
    if not c.alpha:
        return 0
    if not abi.rgb_format_has_alpha(c.rgb_format) or c.ignore_alpha:
        return 0 if c.image_premultiplied else 1
    if not c.image_premultiplied and c.rgb_premultiplied:
        return 1
    if c.image_premultiplied and not c.rgb_premultiplied:
        return 2
    return 0


def test_yuv_to_rgb_dispatch_agrees_with_the_oracle():
    lib, o = native.load(), oracle_lib.oracle()
    cases = H.libyuv_y2r_cases(SIZES, n_random=1500) + [replace(c, avoid_libyuv=False) for c in H.y2r_sweep(SIZES, n_random=800, seed=3)]
    cases += H.y2r_sweep(SIZES[:2], n_random=300, seed=4)  # avoidLibYUV = 1
    lib.avifhipSetArithmetic(0)
    fixed = 0
    for c in cases:
        img, rgb = H.make_y2r_inputs(c), H.make_y2r_output(c)
        res, plan = native.explain_y2r(img.struct, rgb.struct)
        assert res == 0, c.ident()
        mul = _mul_mode(c)
        has_alpha = abi.rgb_format_has_alpha(c.rgb_format)
        reformat_alpha = has_alpha and (not c.ignore_alpha or mul != 0)
        asked = (not c.avoid_libyuv) and (mul == 0 or has_alpha)  # src/reformat.c:1453
        done = C.c_int(0)
        hook = o.oracleLibyuvHookYUVToRGB(img.struct, rgb.struct, int(reformat_alpha), C.byref(done)) if asked else abi.AVIF_RESULT_NOT_IMPLEMENTED
        assert hook in (0, abi.AVIF_RESULT_NOT_IMPLEMENTED), c.ident()
        assert (plan["arith"] == "libyuv") == (hook == 0), (c.ident(), plan)
        if hook == 0:
            fixed += 1
            assert plan["inloopmul"] == "0" and int(plan["postmul"]) == mul, (c.ident(), plan)
            if reformat_alpha:  # alphaReformattedWithLibYUV is meaningful only then (include/avif/internal.h:361-363)
                assert (plan["alpha"] in ("fill", "plane-shift")) == bool(done.value), (c.ident(), plan, done.value)
            elif has_alpha:
                assert plan["alpha"] == "fill", (c.ident(), plan)  # libyuv writes its 255 even under ignoreAlpha
        else:
            assert int(plan["inloopmul"]) + int(plan["postmul"]) == mul, (c.ident(), plan)
        # ARGBAttenuate / ARGBUnattenuate: asked whatever avoidLibYUV says (src/alpha.c:163,350)
        expect_fx = int(plan["postmul"]) != 0 and c.rgb_depth == 8 and c.rgb_format in (abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_BGRA)
        assert plan["postmulfx"] == str(int(expect_fx)), (c.ident(), plan)
    assert fixed > 1200
    # a libavif built without libyuv: never the integer path
    lib.avifhipSetArithmetic(1)
    try:
        for c in cases[:300]:
            img, rgb = H.make_y2r_inputs(c), H.make_y2r_output(c)
            res, plan = native.explain_y2r(img.struct, rgb.struct)
            assert res == 0 and plan["arith"] == "fp32" and plan["postmulfx"] == "0", (c.ident(), plan)
    finally:
        lib.avifhipSetArithmetic(0)


def test_rgb_to_yuv_dispatch_agrees_with_the_oracle():
    lib, o = native.load(), oracle_lib.oracle()
    lib.avifhipSetArithmetic(0)
    cases = H.libyuv_r2y_cases(SIZES, n_random=800) + [replace(c, avoid_libyuv=False) for c in H.r2y_sweep(SIZES, n_random=400, seed=5)]
    cases += H.r2y_sweep(SIZES[:2], n_random=200, seed=6)
    fixed = 0
    for c in cases:
        rgb, img = H.make_r2y_inputs(c), H.make_r2y_output(c)
        res, plan = native.explain_r2y(img.struct, rgb.struct)
        assert res == 0, c.ident()
        has_alpha = abi.rgb_format_has_alpha(c.rgb_format) and not c.ignore_alpha
        mul = 0
        if has_alpha:  # src/reformat.c:242-249
            mul = 1 if (not c.rgb_premultiplied and c.image_premultiplied) else 2 if (c.rgb_premultiplied and not c.image_premultiplied) else 0
        assert int(plan["mul"]) == mul, (c.ident(), plan)
        gray = c.rgb_format in (abi.AVIF_RGB_FORMAT_GRAY, abi.AVIF_RGB_FORMAT_GRAYA, abi.AVIF_RGB_FORMAT_AGRAY)
        asked = not gray and not c.avoid_libyuv and mul == 0  # src/reformat.c:264-265
        hook = o.oracleLibyuvHookRGBToYUV(img.struct, rgb.struct) if asked else abi.AVIF_RESULT_NOT_IMPLEMENTED
        assert (plan["arith"] == "libyuv") == (hook == 0), (c.ident(), plan)
        fixed += hook == 0
    assert fixed > 500


def test_error_codes_come_from_the_plan_layer():
    for c in (H.Y2RCase(8, 8, matrix=3), H.Y2RCase(8, 8, rgb_depth=9), H.Y2RCase(8, 8, matrix=0, yuv_format=abi.AVIF_PIXEL_FORMAT_YUV420, yuv_range=1)):
        img, rgb = H.make_y2r_inputs(replace(c, matrix=1, rgb_depth=8, yuv_format=1)), H.make_y2r_output(replace(c, rgb_depth=8))
        img.struct.matrixCoefficients, img.struct.yuvFormat, rgb.struct.depth = c.matrix, c.yuv_format, c.rgb_depth
        res, _ = native.explain_y2r(img.struct, rgb.struct)
        assert res == abi.AVIF_RESULT_REFORMAT_FAILED, c.ident()
