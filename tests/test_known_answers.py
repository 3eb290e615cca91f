"""Known answers the reference's own tests pin for the reformat path, restated as Python over the
C ABI and asked of every implementation of the path we hold:

  * the two oracles (fp32 restatement = a libavif built without libyuv; integer-path restatement = the default build),
  * the reference compiled from its sources (oracle/_ref, when present: build container only),
  * the HIP library in both arithmetics (-m gpu).

Facts and thresholds come from tests/gtest/avifrgbtoyuvtest.cc and tests/gtest/avifalphapremtest.cc of the reference
(file:line cited per test).  CPU backends use the reference's own steps through the RGB cube; on the GPU (one synchronous host call per 4x4
image) the two densest walks are coarsened, which the suite table marks."""
from __future__ import annotations

import math

import numpy as np
import pytest

import harness as H
from libavif_amd import abi, synth

# tests/gtest/avifrgbtoyuvtest.cc:95-110
RED_NOISE = np.array([7, 14, 11, 5, 4, 6, 8, 15, 2, 9, 13, 3, 12, 1, 10, 0], dtype=np.int64)
GREEN_NOISE = np.array([3, 2, 12, 15, 14, 10, 7, 13, 5, 1, 9, 0, 8, 4, 11, 6], dtype=np.int64)
BLUE_NOISE = np.array([0, 8, 14, 9, 13, 12, 2, 7, 3, 1, 11, 10, 6, 15, 5, 4], dtype=np.int64)

F444, F422, F420, F400 = 1, 2, 3, 4
LIMITED, FULL = 0, 1
BT709, BT601, IDENTITY, YCGCO_RE = 1, 6, 0, 16


# ---------------------------------------------------------------------------------------------------
# backends


def _cpu_backends():
    out = [pytest.param("oracle", id="oracle-fp32"), pytest.param("oracle-libyuv", id="oracle-integer")]
    out.append(pytest.param("ref", id="reference-build"))
    out.append(pytest.param("hip-float", id="hip-fp32", marks=pytest.mark.gpu))
    out.append(pytest.param("hip-auto", id="hip-default", marks=pytest.mark.gpu))
    return out


@pytest.fixture(params=_cpu_backends())
def backend(request):
    name = request.param
    if name == "oracle":
        yield H.oracle_backend()
    elif name == "oracle-libyuv":
        yield H.oracle_libyuv_backend()
    elif name == "ref":
        import oracle_lib

        lib = oracle_lib.ref()
        if lib is None:
            pytest.skip("oracle/_ref/libavif_ref.so not built (needs /root/reference)")
        yield H.libavif_backend(lib, "reference")
    else:
        hip = request.getfixturevalue("hip")
        hip.avifhipSetArithmetic(1 if name == "hip-float" else 0)
        try:
            yield H.hip_host_backend()
        finally:
            hip.avifhipSetArithmetic(1)


def _is_gpu(be) -> bool:
    return be.name.startswith("hip")


# ---------------------------------------------------------------------------------------------------
# helpers mirroring the reference's test utilities


def _modify(channel: np.ndarray, noise: np.ndarray) -> None:
    """ModifyImageChannel (avifrgbtoyuvtest.cc:25-38): adds noise[i % 16] in row order, wrapping like the pixel type."""
    h, w = channel.shape
    idx = np.arange(h * w).reshape(h, w) % 16
    channel[...] = (channel.astype(np.int64) + noise[idx]).astype(channel.dtype)


def _offsets(fmt: int):
    return {abi.AVIF_RGB_FORMAT_RGB: (0, 1, 2, None), abi.AVIF_RGB_FORMAT_RGBA: (0, 1, 2, 3), abi.AVIF_RGB_FORMAT_ARGB: (1, 2, 3, 0),
            abi.AVIF_RGB_FORMAT_BGR: (2, 1, 0, None), abi.AVIF_RGB_FORMAT_BGRA: (2, 1, 0, 3), abi.AVIF_RGB_FORMAT_ABGR: (3, 2, 1, 0)}[fmt]


class _RoundTrip:
    """RGB -> YUV -> RGB of one image through a backend, buffers allocated once (avifrgbtoyuvtest.cc:117-148)."""

    def __init__(self, be, w, h, rgb_depth, yuv_depth, rgb_format, yuv_format, yuv_range, matrix):
        self.be = be
        self.yuv = abi.make_yuv(w, h, yuv_depth, yuv_format, yuv_range, matrix)
        # avifRGBImageSetDefaults: avoidLibYUV = 0, automatic up/downsampling (src/avif.c:700-718)
        self.src = abi.make_rgb(w, h, rgb_depth, rgb_format, avoid_libyuv=False)
        self.dst = abi.make_rgb(w, h, rgb_depth, rgb_format, avoid_libyuv=False)
        self.r, self.g, self.b, self.a = _offsets(rgb_format)
        self.ch = self.src.channels()
        if self.a is not None:
            self.ch[:, :, self.a] = (1 << rgb_depth) - 1

    def run(self):
        assert self.be.rgb_to_yuv(self.yuv.struct, self.src.struct) == abi.AVIF_RESULT_OK
        assert self.be.yuv_to_rgb(self.yuv.struct, self.dst.struct) == abi.AVIF_RESULT_OK
        d = self.dst.channels().astype(np.int64) - self.ch.astype(np.int64)
        return int(np.abs(d).sum()), int((d * d).sum()), int(np.abs(d).max())


def _psnr(sq_sum: float, n: float, peak: float) -> float:
    """GetPsnr, avifrgbtoyuvtest.cc:82-90."""
    if sq_sum == 0:
        return 99.0
    dist = sq_sum / (n * peak * peak)
    return min(-10 * math.log10(dist), 98.9) if dist > 0 else 98.9


def _walk(max_value: int, step: int):
    v = 0
    while v < max_value + step:
        v = min(v, max_value)
        yield v
        v += step


def convert_whole_range(be, rgb_depth, yuv_depth, rgb_format, yuv_format, yuv_range, matrix, add_noise, step):
    """ConvertWholeRange, avifrgbtoyuvtest.cc:117-222: returns (average |diff|, PSNR)."""
    rgb_max = (1 << rgb_depth) - 1
    rt = _RoundTrip(be, 4, 4, rgb_depth, yuv_depth, rgb_format, yuv_format, yuv_range, matrix)
    mono = yuv_format == F400
    max_value = rgb_max - (15 if add_noise else 0)
    abs_sum = sq_sum = n = 0

    def fill(off, value, noise):
        rt.ch[:, :, off] = value
        if add_noise:
            _modify(rt.ch[:, :, off], noise)

    for r in _walk(max_value, step):
        fill(rt.r, r, RED_NOISE)
        if mono:
            fill(rt.g, r, GREEN_NOISE)
            fill(rt.b, r, BLUE_NOISE)
            a, s, _ = rt.run()
            abs_sum, sq_sum, n = abs_sum + a, sq_sum + s, n + 48
            continue
        for g in _walk(max_value, step):
            fill(rt.g, g, GREEN_NOISE)
            for b in _walk(max_value, step):
                fill(rt.b, b, BLUE_NOISE)
                a, s, _ = rt.run()
                abs_sum, sq_sum, n = abs_sum + a, sq_sum + s, n + 48
    return abs_sum / n, _psnr(sq_sum, n, rgb_max)


def convert_whole_buffer(be, rgb_depth, yuv_depth, rgb_format, yuv_format, yuv_range, matrix, add_noise):
    """ConvertWholeBuffer, avifrgbtoyuvtest.cc:226-286: PSNR over nine buffer shapes."""
    rgb_max = (1 << rgb_depth) - 1
    mono = yuv_format == F400
    sq_sum = n = 0
    for w in (1, 2, 127):
        for h in (1, 2, 251):
            rt = _RoundTrip(be, w, h, rgb_depth, yuv_depth, rgb_format, yuv_format, yuv_range, matrix)
            for off, noise in ((rt.r, RED_NOISE), (rt.g, RED_NOISE if mono else GREEN_NOISE), (rt.b, RED_NOISE if mono else BLUE_NOISE)):
                rt.ch[:, :, off] = 0
                if add_noise:
                    _modify(rt.ch[:, :, off], noise)
            _, s, _ = rt.run()
            sq_sum, n = sq_sum + s, n + w * h * 3
    return _psnr(sq_sum, n, rgb_max)


# ---------------------------------------------------------------------------------------------------
# the reference's instantiations (avifrgbtoyuvtest.cc:583-880, SharpYUV ones excepted: not on this path)
# (name, rgb_depth, yuv_depth, rgb_format, yuv_format, range, matrix, noise, step, coarse step for slow backends,
#  max average |diff|, min PSNR)

RGBA, BGR = abi.AVIF_RGB_FORMAT_RGBA, abi.AVIF_RGB_FORMAT_BGR
SUITES = []


def _suite(name, rd, yds, fmts, yfs, rng, mc, noises, step, max_avg, min_psnr, coarse=None):
    for yd in yds:
        for fmt in fmts:
            for yf in yfs:
                for noise in noises:
                    ident = f"{name}-rgb{rd}-yuv{yd}-{abi.RGB_FORMAT_NAMES[fmt]}-{ {1: '444', 2: '422', 3: '420', 4: '400'}[yf] }-{'noisy' if noise else 'plain'}"
                    SUITES.append(pytest.param(rd, yd, fmt, yf, rng, mc, noise, step, coarse or step, max_avg, min_psnr, id=ident))


_suite("DefaultFormat", 8, [8], [RGBA], [F420], FULL, BT601, [True], 3, 2.88, 36.0, coarse=7)  # :583-596
_suite("Identity8b", 8, [8, 12, 16], [RGBA], [F444], FULL, IDENTITY, [True], 31, 0.0, 99.0)  # :598-609
_suite("Identity10b", 10, [10, 12, 16], [RGBA], [F444], FULL, IDENTITY, [True], 101, 0.0, 99.0)  # :610-621
_suite("Identity12b", 12, [12, 16], [RGBA], [F444], FULL, IDENTITY, [True], 401, 0.0, 99.0)  # :622-633
_suite("Identity16b", 16, [16], [RGBA], [F444], FULL, IDENTITY, [True], 6421, 0.0, 99.0)  # :634-645
_suite("PlainAnySubsampling8b", 8, [8], [RGBA], [F444, F420], FULL, BT601, [False], 17, 0.84, 45.0)  # :648-662
_suite("MonochromeLossless8b", 8, [8], [RGBA], [F400], FULL, BT601, [False], 1, 0.0, 99.0)  # :665-676
_suite("MonochromeLossless10b", 10, [10], [RGBA], [F400], FULL, BT601, [False], 1, 0.0, 99.0)  # :677-688
_suite("MonochromeLossless12b", 12, [12], [RGBA], [F400], FULL, BT601, [False], 1, 0.0, 99.0, coarse=7)  # :689-700
_suite("MonochromeLossless16b", 16, [16], [RGBA], [F400], FULL, BT601, [False], 401, 0.0, 99.0)  # :701-712
_suite("YCgCo_Re8b", 8, [10], [RGBA], [F444], FULL, YCGCO_RE, [True], 101, 0.0, 99.0)  # :715-726
_suite("All8bTo8b", 8, [8], [RGBA, BGR], [F444, F422, F420], LIMITED, BT601, [False, True], 61, 2.96, 36.0)  # :817-830
_suite("All10b", 10, [10], [RGBA], [F444, F420], FULL, BT601, [False, True], 211, 2.83, 47.0)  # :831-842
_suite("All12b", 12, [12], [RGBA], [F444, F420], LIMITED, BT601, [False, True], 809, 2.82, 52.0)  # :843-854
_suite("All16b", 16, [16], [RGBA], [F444, F420], FULL, BT601, [False, True], 16001, 2.82, 80.0)  # :855-866


@pytest.mark.parametrize("rd,yd,fmt,yf,rng,mc,noise,step,coarse,max_avg,min_psnr", SUITES)
def test_convert_whole_range(backend, rd, yd, fmt, yf, rng, mc, noise, step, coarse, max_avg, min_psnr):
    """TEST_P(RGBToYUVTest, ConvertWholeRange), avifrgbtoyuvtest.cc:533-553.  CPU implementations walk the cube
    with the reference's own step; the GPU's synchronous host entry points (one upload + launch + download per 4x4 image)
    use the coarse step where the reference's would need > 10^5 round trips."""
    use = coarse if _is_gpu(backend) else step
    avg, psnr = convert_whole_range(backend, rd, yd, fmt, yf, rng, mc, noise, use)
    assert avg <= max_avg, (avg, psnr)
    assert psnr >= min_psnr, (avg, psnr)


@pytest.mark.parametrize("rd,yd,fmt,yf,rng,mc,noise,step,coarse,max_avg,min_psnr", SUITES)
def test_convert_whole_buffer(backend, rd, yd, fmt, yf, rng, mc, noise, step, coarse, max_avg, min_psnr):
    """TEST_P(RGBToYUVTest, ConvertWholeBuffer), avifrgbtoyuvtest.cc:555-573."""
    assert convert_whole_buffer(backend, rd, yd, fmt, yf, rng, mc, noise) >= min_psnr


# ---------------------------------------------------------------------------------------------------
# gray known answers


@pytest.mark.parametrize("depth,half", [(8, 128), (10, 512), (12, 2048), (16, 32768)])
def test_gray_to_yuv420(backend, depth, half):
    """8BitGrayToYUV420 / HighBitDepthGrayToYUV420, avifrgbtoyuvtest.cc:415-466: gray samples land in Y unchanged and both
    chroma planes hold the half value."""
    img = abi.make_yuv(2, 2, depth, F420, FULL, BT601)
    rgb = abi.make_rgb(2, 2, depth, abi.AVIF_RGB_FORMAT_GRAY, avoid_libyuv=True)
    rgb.channels()[:, :, 0] = np.array([[4, 3], [2, 1]])
    assert backend.rgb_to_yuv(img.struct, rgb.struct) == abi.AVIF_RESULT_OK
    assert img.plane_samples(0).tolist() == [[4, 3], [2, 1]]
    assert img.plane_samples(1).tolist() == [[half]]
    assert img.plane_samples(2).tolist() == [[half]]


@pytest.mark.parametrize("yuv_range", [LIMITED, FULL])
def test_gray_round_trip_with_lift(backend, yuv_range):
    """8BitGrayRoundTripWithLift, avifrgbtoyuvtest.cc:468-501: 8-bit gray -> 12-bit YUV400 -> 8-bit gray is lossless."""
    img = abi.make_yuv(2, 2, 12, F400, yuv_range, BT601)
    src = abi.make_rgb(2, 2, 8, abi.AVIF_RGB_FORMAT_GRAY, avoid_libyuv=True)
    src.channels()[:, :, 0] = np.array([[5, 3], [2, 1]])
    dst = abi.make_rgb(2, 2, 8, abi.AVIF_RGB_FORMAT_GRAY, avoid_libyuv=True)
    assert backend.rgb_to_yuv(img.struct, src.struct) == abi.AVIF_RESULT_OK
    assert backend.yuv_to_rgb(img.struct, dst.struct) == abi.AVIF_RESULT_OK
    assert dst.channels()[:, :, 0].tolist() == [[5, 3], [2, 1]]


# ---------------------------------------------------------------------------------------------------
# alpha known answers


def test_opaque_alpha_is_no_op(backend):
    """AlphaMultiplyTest.OpaqueIsNoOp, avifalphapremtest.cc:15-62: premultiplied YUVA with an opaque plane converts to the
    same RGB bytes as the same image without the alpha plane."""
    w = h = 1024
    with_alpha = abi.make_yuv(w, h, 8, F444, FULL, BT601, with_alpha=True, alpha_premultiplied=True)
    synth.gradient_planes(with_alpha)
    with_alpha.alpha[...] = 255
    without = abi.make_yuv(w, h, 8, F444, FULL, BT601, with_alpha=False, alpha_premultiplied=True)
    for p in range(3):
        without.planes[p][...] = with_alpha.planes[p]
    out_a = abi.make_rgb(w, h, 8, abi.AVIF_RGB_FORMAT_RGB, avoid_libyuv=False)
    out_b = abi.make_rgb(w, h, 8, abi.AVIF_RGB_FORMAT_RGB, avoid_libyuv=False)
    assert backend.yuv_to_rgb(with_alpha.struct, out_a.struct) == abi.AVIF_RESULT_OK
    assert backend.yuv_to_rgb(without.struct, out_b.struct) == abi.AVIF_RESULT_OK
    assert np.array_equal(out_a.pixels, out_b.pixels)


def test_gray_alpha_premultiply(backend):
    """AlphaMultiplyTest.GrayAImagePremultiplyAlpha, avifalphapremtest.cc:64-75, asks only for AVIF_RESULT_OK on a 6x1
    10-bit GRAYA image; this also checks what the call must have done (src/alpha.c:147-342): the gray channel is scaled
    by alpha/max like a colour channel, the alpha channel is untouched, and premultiply followed by unpremultiply restores
    every sample whose alpha is not zero to within two code values."""
    rgb = abi.make_rgb(6, 1, 10, abi.AVIF_RGB_FORMAT_GRAYA, avoid_libyuv=True)
    ch = rgb.channels()
    ch[0, :, 0] = [1023, 1023, 512, 512, 100, 0]
    ch[0, :, 1] = [1023, 0, 1023, 512, 256, 77]
    before = ch.copy()
    assert backend.premultiply(rgb.struct) == abi.AVIF_RESULT_OK
    assert ch[0, :, 1].tolist() == before[0, :, 1].tolist()
    expect = np.floor(before[0, :, 0].astype(np.float64) * before[0, :, 1] / 1023.0 + 0.5).astype(np.int64)
    assert np.abs(ch[0, :, 0].astype(np.int64) - expect).max() <= 1, (ch[0, :, 0], expect)
    assert ch[0, 0, 0] == 1023 and ch[0, 1, 0] == 0
    rgb.struct.alphaPremultiplied = 1
    assert backend.unpremultiply(rgb.struct) == abi.AVIF_RESULT_OK
    nz = before[0, :, 1] != 0
    assert np.abs(ch[0, nz, 0].astype(np.int64) - before[0, nz, 0].astype(np.int64)).max() <= 2
