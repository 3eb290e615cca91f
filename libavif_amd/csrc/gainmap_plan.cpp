// gainmap_plan.cpp -- see gainmap_plan.h.  Host-only (no HIP): transfer functions, primaries matrices, tables.
#include "gainmap_plan.h"

#include <float.h>
#include <math.h>
#include <string.h>

#include <map>
#include <mutex>

namespace avifhip {

namespace {

inline float clampf(float x, float lo, float hi)
{
    return (x < lo) ? lo : ((hi < x) ? hi : x);
}

const float kPqMaxNits = 10000.0f, kHlgPeakNits = 1000.0f, kSdrWhiteNits = 203.0f;

} // namespace

float gainMapToLinear(int tc, float g)
{
    switch (tc) {
        case 4: return powf(clampf(g, 0.0f, 1.0f), 2.2f);  // BT.470M, src/colr.c:240-243
        case 5: return powf(clampf(g, 0.0f, 1.0f), 2.8f);  // BT.470BG, :250-253
        case 7:                                             // SMPTE 240M, :260-271
            if (g < 0.0f)
                return 0.0f;
            if (g < 4.0f * 0.022821585529445f)
                return g / 4.0f;
            if (g < 1.0f)
                return powf((g + 0.111572195921731f) / 1.111572195921731f, 1.0f / 0.45f);
            return 1.0f;
        case 8: return clampf(g, 0.0f, 1.0f);               // linear, :286-289
        case 9:                                             // log 100:1, :291-296
            return (g <= 0.0f) ? (0.01f / 2.f) : powf(10.0f, 2.f * (fminf(g, 1.f) - 1.0f));
        case 10:                                            // log 100*sqrt(10):1, :303-308
            return (g <= 0.0f) ? (0.00316227766f / 2.f) : powf(10.0f, 2.5f * (fminf(g, 1.f) - 1.0f));
        case 11:                                            // IEC 61966-2-4, :315-324
            if (g < -4.5f * 0.018053968510807f)
                return -powf((g - 0.09929682680944f) / -1.09929682680944f, 1.0f / 0.45f);
            if (g < 4.5f * 0.018053968510807f)
                return g / 4.5f;
            return powf((g + 0.09929682680944f) / 1.09929682680944f, 1.0f / 0.45f);
        case 12:                                            // BT.1361, :337-350
            if (g < -0.25f)
                return -0.25f;
            if (g < 0.0f)
                return powf((g - 0.02482420670236f) / -0.27482420670236f, 1.0f / 0.45f) / -4.0f;
            if (g < 4.5f * 0.018053968510807f)
                return g / 4.5f;
            if (g < 1.0f)
                return powf((g + 0.09929682680944f) / 1.09929682680944f, 1.0f / 0.45f);
            return 1.0f;
        case 13:                                            // sRGB, :367-378
            if (g < 0.0f)
                return 0.0f;
            if (g < 12.92f * 0.0030412825601275209f)
                return g / 12.92f;
            if (g < 1.0f)
                return powf((g + 0.0550107189475866f) / 1.0550107189475866f, 2.4f);
            return 1.0f;
        case 16: {                                          // PQ, :397-409: extended SDR, 1.0 = 203 nits
            if (!(g > 0.0f))
                return 0.0f;
            const float p = powf(g, 1.0f / 78.84375f);
            const float num = (p - 0.8359375f > 0.0f) ? p - 0.8359375f : 0.0f;
            const float denRaw = 18.8515625f - 18.6875f * p;
            const float den = (denRaw > FLT_MIN) ? denRaw : FLT_MIN;
            return powf(num / den, 1.0f / 0.1593017578125f) * kPqMaxNits / kSdrWhiteNits;
        }
        case 17: return powf((g > 0.0f) ? g : 0.0f, 2.6f) / 0.91655527974030934f; // SMPTE 428, :425-428
        case 18: {                                          // HLG with OOTF, :439-455
            if (g < 0.0f)
                return 0.0f;
            const float l = (g <= 0.5f) ? powf((g * g) * (1.0f / 3.0f), 1.2f)
                                        : powf((expf((g - 0.55991073f) / 0.17883277f) + 0.28466892f) / 12.0f, 1.2f);
            return l * kHlgPeakNits / kSdrWhiteNits;
        }
        default: // BT.709, BT.601, BT.2020 10/12-bit, and libavif's default for everything else, :214-225, :494-503
            if (g < 0.0f)
                return 0.0f;
            if (g < 4.5f * 0.018053968510807f)
                return g / 4.5f;
            if (g < 1.0f)
                return powf((g + 0.09929682680944f) / 1.09929682680944f, 1.0f / 0.45f);
            return 1.0f;
    }
}

float gainMapToGamma(int tc, float l)
{
    switch (tc) {
        case 4: return powf(clampf(l, 0.0f, 1.0f), 1.0f / 2.2f); // :245-248
        case 5: return powf(clampf(l, 0.0f, 1.0f), 1.0f / 2.8f); // :255-258
        case 7:                                                   // :273-284
            if (l < 0.0f)
                return 0.0f;
            if (l < 0.022821585529445f)
                return l * 4.0f;
            if (l < 1.0f)
                return 1.111572195921731f * powf(l, 0.45f) - 0.111572195921731f;
            return 1.0f;
        case 8: return clampf(l, 0.0f, 1.0f);
        case 9: return l <= 0.01f ? 0.0f : 1.0f + log10f(fminf(l, 1.0f)) / 2.0f;          // :298-301
        case 10: return l <= 0.00316227766f ? 0.0f : 1.0f + log10f(fminf(l, 1.0f)) / 2.5f; // :310-313
        case 11:                                                                            // :326-335
            if (l < -0.018053968510807f)
                return -1.09929682680944f * powf(-l, 0.45f) + 0.09929682680944f;
            if (l < 0.018053968510807f)
                return l * 4.5f;
            return 1.09929682680944f * powf(l, 0.45f) - 0.09929682680944f;
        case 12: // :352-365
            if (l < -0.25f)
                return -0.25f;
            if (l < 0.0f)
                return -0.27482420670236f * powf(-4.0f * l, 0.45f) + 0.02482420670236f;
            if (l < 0.018053968510807f)
                return l * 4.5f;
            if (l < 1.0f)
                return 1.09929682680944f * powf(l, 0.45f) - 0.09929682680944f;
            return 1.0f;
        case 13: // :380-391
            if (l < 0.0f)
                return 0.0f;
            if (l < 0.0030412825601275209f)
                return l * 12.92f;
            if (l < 1.0f)
                return 1.0550107189475866f * powf(l, 1.0f / 2.4f) - 0.0550107189475866f;
            return 1.0f;
        case 16: { // :411-423
            if (!(l > 0.0f))
                return 0.0f;
            const float s = clampf(l * kSdrWhiteNits / kPqMaxNits, 0.0f, 1.0f);
            const float p = powf(s, 0.1593017578125f);
            const float num = 0.1640625f * p - 0.1640625f;
            const float den = 1.0f + 18.6875f * p;
            return powf(1.0f + num / den, 78.84375f);
        }
        case 17: return powf(0.91655527974030934f * ((l > 0.0f) ? l : 0.0f), 1.0f / 2.6f); // :430-433
        case 18: {                                                                         // :457-470
            float s = clampf(l * kSdrWhiteNits / kHlgPeakNits, 0.0f, 1.0f);
            s = powf(s, 1.0f / 1.2f);
            if (s < 0.0f)
                return 0.0f;
            if (s <= (1.0f / 12.0f))
                return sqrtf(3.0f * s);
            return 0.17883277f * logf(12.0f * s - 0.28466892f) + 0.55991073f;
        }
        default: // :227-238
            if (l < 0.0f)
                return 0.0f;
            if (l < 0.018053968510807f)
                return l * 4.5f;
            if (l < 1.0f)
                return 1.09929682680944f * powf(l, 0.45f) - 0.09929682680944f;
            return 1.0f;
    }
}

// ---- primaries, src/colr.c:16-43; matrices, src/colrconvert.c ----

namespace {

struct Mat3
{
    double m[3][3];
};

bool invert(const Mat3 & M, Mat3 & I) // :26-47
{
    const double(*a)[3] = M.m;
    double det = a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                 a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (fabs(det) < 1e-12)
        return false;
    det = 1.0 / det;
    I.m[0][0] = (a[1][1] * a[2][2] - a[2][1] * a[1][2]) * det;
    I.m[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * det;
    I.m[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * det;
    I.m[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) * det;
    I.m[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * det;
    I.m[1][2] = (a[1][0] * a[0][2] - a[0][0] * a[1][2]) * det;
    I.m[2][0] = (a[1][0] * a[2][1] - a[2][0] * a[1][1]) * det;
    I.m[2][1] = (a[2][0] * a[0][1] - a[0][0] * a[2][1]) * det;
    I.m[2][2] = (a[0][0] * a[1][1] - a[1][0] * a[0][1]) * det;
    return true;
}
Mat3 mul(const Mat3 & A, const Mat3 & B) // :50-61
{
    Mat3 C;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            C.m[r][c] = A.m[r][0] * B.m[0][c] + A.m[r][1] * B.m[1][c] + A.m[r][2] * B.m[2][c];
    return C;
}
Mat3 diag(const double d[3]) // :64-75
{
    Mat3 M;
    memset(&M, 0, sizeof(M));
    M.m[0][0] = d[0], M.m[1][1] = d[1], M.m[2][2] = d[2];
    return M;
}
void apply(const Mat3 & M, const double x[3], double y[3]) // :78-83
{
    for (int r = 0; r < 3; ++r)
        y[r] = M.m[r][0] * x[0] + M.m[r][1] * x[1] + M.m[r][2] * x[2];
}

const float * primariesOf(int cp) // rX rY gX gY bX bY wX wY
{
    static const struct
    {
        int cp;
        float v[8];
    } table[] = { { 1, { 0.64f, 0.33f, 0.3f, 0.6f, 0.15f, 0.06f, 0.3127f, 0.329f } },
                  { 4, { 0.67f, 0.33f, 0.21f, 0.71f, 0.14f, 0.08f, 0.310f, 0.316f } },
                  { 5, { 0.64f, 0.33f, 0.29f, 0.60f, 0.15f, 0.06f, 0.3127f, 0.3290f } },
                  { 6, { 0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f } },
                  { 7, { 0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f } },
                  { 8, { 0.681f, 0.319f, 0.243f, 0.692f, 0.145f, 0.049f, 0.310f, 0.316f } },
                  { 9, { 0.708f, 0.292f, 0.170f, 0.797f, 0.131f, 0.046f, 0.3127f, 0.3290f } },
                  { 10, { 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.3333f, 0.3333f } },
                  { 11, { 0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.314f, 0.351f } },
                  { 12, { 0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.3127f, 0.3290f } },
                  { 22, { 0.630f, 0.340f, 0.295f, 0.605f, 0.155f, 0.077f, 0.3127f, 0.3290f } } };
    for (const auto & e : table)
        if (e.cp == cp)
            return e.v;
    return table[0].v; // "a reasonable default", src/colr.c:40-42
}

bool rgbToXyzD50(int cp, Mat3 & out) // :97-153
{
    static const Mat3 bradford = { { { 0.8951, 0.2664, -0.1614 }, { -0.7502, 1.7135, 0.0367 }, { 0.0389, -0.0685, 1.0296 } } };
    static const double lmsD50[3] = { 0.996284, 1.02043, 0.818644 };
    const float * p = primariesOf(cp);
    if (fabsf(p[7]) < 1e-12)
        return false;
    const double factor = 1.0 / p[7];
    const double white[3] = { p[6] * factor, 1, (1 - p[6] - p[7]) * factor };
    const Mat3 prim = { { { p[0], p[2], p[4] }, { p[1], p[3], p[5] }, { 1.0 - p[0] - p[1], 1.0 - p[2] - p[3], 1.0 - p[4] - p[5] } } };
    Mat3 primInv;
    if (!invert(prim, primInv))
        return false;
    double coefficients[3];
    apply(primInv, white, coefficients);
    const Mat3 rgbXyz = mul(prim, diag(coefficients));
    double lms[3];
    apply(bradford, white, lms);
    for (int i = 0; i < 3; ++i) {
        if (fabs(lms[i]) < 1e-12)
            return false;
        lms[i] = lmsD50[i] / lms[i];
    }
    Mat3 bradfordInv;
    if (!invert(bradford, bradfordInv))
        return false;
    const Mat3 adaptation = mul(bradfordInv, mul(diag(lms), bradford));
    out = mul(adaptation, rgbXyz);
    return true;
}

} // namespace

bool gainMapPrimariesMatrix(int src, int dst, double coeffs[9])
{
    Mat3 srcToXyz, dstToXyz, xyzToDst;
    if (!rgbToXyzD50(src, srcToXyz) || !rgbToXyzD50(dst, dstToXyz) || !invert(dstToXyz, xyzToDst))
        return false;
    const Mat3 M = mul(xyzToDst, srcToXyz);
    memcpy(coeffs, M.m, sizeof(M.m));
    return true;
}

// ---- tables ----

namespace {

const float kF16Multiplier = 1.9259299444e-34f; // src/reformat.c:1411

inline float f16ToFloat(uint32_t code)
{
    const uint32_t u = code << 13;
    float f;
    memcpy(&f, &u, 4);
    return f / kF16Multiplier;
}
inline uint32_t floatToF16(float v)
{
    const float f = v * kF16Multiplier;
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u >> 13) & 0xffffu;
}

// the ordered fp32 values: key 0 = -inf ... increasing with the value; NaNs excluded
inline float floatOfKey(uint32_t key)
{
    const uint32_t bits = (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
inline uint32_t keyOfFloat(float f)
{
    uint32_t bits;
    memcpy(&bits, &f, 4);
    return (bits & 0x80000000u) ? ~bits : (bits ^ 0x80000000u);
}

inline uint32_t bitsOf(float f)
{
    uint32_t b;
    memcpy(&b, &f, 4);
    return b;
}

// GainMapSteps::locator (gainmap_plan.h) from the finished step tables; leaves it empty when the curve does not qualify
void buildLocator(GainMapSteps & S, bool isFloat)
{
    if (isFloat || S.maxCode > 4095 || S.maxCode < 1)
        return;
    const uint32_t n = S.pieceEntries;
    const float * Tn = S.steps.data();
    const float * T = S.steps.data() + n;
    if (Tn[1] != INFINITY || !(T[1] > 0.0f) || T[1] == INFINITY || bitsOf(T[1]) < 2)
        return; // a negative x reaches a code above 0, or +0 does
    uint32_t K = 1;
    while (K < S.maxCode && T[K + 1] != INFINITY)
        ++K;
    for (uint32_t k = 1; k < K; ++k)
        if (!(T[k] < T[k + 1]))
            return;
    // codes of more than 8 bits are packed by their low 16 bits: the threshold field has to stay clear of those
    for (uint32_t shift = (S.maxCode > 255) ? 16 : 19; shift >= 6; --shift) {
        const uint32_t mask = (1u << shift) - 1, first = (bitsOf(T[1]) - 1) & ~mask;
        bool distinct = true;
        for (uint32_t k = 1; k < K && distinct; ++k)
            distinct = ((bitsOf(T[k]) - first) >> shift) != ((bitsOf(T[k + 1]) - first) >> shift);
        if (!distinct)
            continue;
        const uint32_t buckets = ((bitsOf(T[K]) - first) >> shift) + 2;
        if (buckets > kGainMapLocatorMaxBuckets)
            return; // finer buckets only grow the table
        const uint32_t none = mask << (32 - shift);
        std::vector<uint32_t> loc(buckets);
        uint32_t k = 0; // the code at the start of the bucket
        for (uint32_t b = 0; b < buckets; ++b) {
            const uint32_t start = first + (b << shift);
            while (k < K && bitsOf(T[k + 1]) <= start)
                ++k;
            uint32_t e = none | k;
            if (k < K && ((bitsOf(T[k + 1]) - first) >> shift) == b)
                e = ((bitsOf(T[k + 1]) - start - 1) << (32 - shift)) | k; // its offset is >= 1: a step AT the start is in k already
            loc[b] = e;
        }
        S.locator = std::move(loc), S.locFirstBits = first, S.locShift = shift, S.locBuckets = buckets;
        // the locator must say what the steps say at, and just below, every step and at the ends of the piece
        bool ok = gainMapLocate(S, 0.0f) == 0 && gainMapLocate(S, -0.0f) == 0 && gainMapLocate(S, INFINITY) == K && gainMapLocate(S, -1.0f) == 0;
        for (uint32_t c = 1; c <= K && ok; ++c)
            ok = gainMapLocate(S, T[c]) == c && gainMapLocate(S, nextafterf(T[c], -INFINITY)) == c - 1;
        if (!ok)
            S.locator.clear();
        return;
    }
}

} // namespace

// (a function of its three arguments only: built once per process and kept -- two powf per entry, ~80 us of every computation call for an
//  8-bit / 10-bit pair until round 6)
const std::vector<float> & gainMapLinearLut(int tc, uint32_t depth, bool isFloat)
{
    static std::mutex mutex;
    static std::map<uint32_t, std::vector<float>> cache;
    const uint32_t key = ((uint32_t)tc << 16) | (depth << 1) | (isFloat ? 1u : 0u);
    std::lock_guard<std::mutex> lock(mutex);
    auto it = cache.find(key);
    if (it != cache.end())
        return it->second;
    const uint32_t n = isFloat ? 65536u : (1u << depth);
    std::vector<float> lut(n);
    const float maxF = (float)((1u << depth) - 1);
    for (uint32_t v = 0; v < n; ++v)
        lut[v] = gainMapToLinear(tc, isFloat ? f16ToFloat(v) : (float)v / maxF); // avifGetRGBAPixel, src/reformat.c:1857-1888
    return cache.emplace(key, std::move(lut)).first->second; // (map nodes stay where they are: the reference outlives the lock)
}

std::vector<float> gainMapGainLut(uint32_t depth, float gammaInv, float minLog2, float maxLog2, float weight)
{
    const uint32_t n = 1u << depth;
    std::vector<float> lut(n);
    const float maxF = (float)(n - 1);
    for (uint32_t v = 0; v < n; ++v) {
        const float w = powf((float)v / maxF, gammaInv);
        const float gainMapLog2 = (1.0f - w) * minLog2 + w * maxLog2; // lerp, src/gainmap.c:66-69
        lut[v] = exp2f(gainMapLog2 * weight);
    }
    return lut;
}

const GainMapSteps & gainMapOutputSteps(int tc, uint32_t depth, bool isFloat)
{
    static std::mutex mutex;
    static std::map<uint32_t, GainMapSteps> cache;
    // transfer characteristics that share libavif's default curve share an entry
    const int curve = (tc == 4 || tc == 5 || (tc >= 7 && tc <= 13) || (tc >= 16 && tc <= 18)) ? tc : 1;
    const uint32_t key = ((uint32_t)curve << 16) | (depth << 1) | (isFloat ? 1u : 0u);
    std::lock_guard<std::mutex> lock(mutex);
    auto it = cache.find(key);
    if (it != cache.end())
        return it->second;
    GainMapSteps S;
    S.maxCode = isFloat ? 0x3c00u : ((1u << depth) - 1);
    const float maxF = (float)((1u << depth) - 1);
    auto codeOf = [&](float x) -> uint32_t {
        const float v = fminf(1.0f, fmaxf(0.0f, gainMapToGamma(curve, x))); // avifNanSafeClamp, src/gainmap.c:13-16
        return isFloat ? floatToF16(v) : (uint32_t)(0.5f + v * maxF);        // avifSetRGBAPixel, src/reformat.c:1906-1938
    };
    // Two pieces, x < 0 and x >= 0, each monotone for every curve (BT.1361's negative branch ends ABOVE its value at +0,
    // src/colr.c:352-365, so one table over all of fp32 would not be a step function): piece p occupies
    // steps[p * pieceEntries ...], unreachable codes keep +inf.
    uint32_t n = 1;
    while (n < S.maxCode + 1)
        n <<= 1; // a power of two per piece (branch-free halving search); entries past maxCode are NaN: never <= x
    S.pieceEntries = n;
    S.steps.assign((size_t)2 * n, NAN);
    const uint32_t pieceLo[2] = { keyOfFloat(-INFINITY), keyOfFloat(0.0f) }, pieceHi[2] = { keyOfFloat(-0.0f) - 1, keyOfFloat(INFINITY) };
    for (int p = 0; p < 2; ++p) {
        float * T = S.steps.data() + (size_t)p * n;
        for (uint32_t k = 1; k <= S.maxCode; ++k)
            T[k] = INFINITY;
        T[0] = -INFINITY;
        uint32_t lowKey = pieceLo[p]; // steps are non-decreasing in k: each search starts where the previous one ended
        for (uint32_t k = 1; k <= S.maxCode; ++k) {
            if (codeOf(floatOfKey(pieceHi[p])) < k)
                break;
            uint32_t lo = lowKey, hi = pieceHi[p]; // invariant: codeOf(hi) >= k; the answer lies in [lo, hi]
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (codeOf(floatOfKey(mid)) >= k)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            T[k] = floatOfKey(lo);
            lowKey = lo;
        }
    }
    // the guide over the x >= 0 piece
    S.guide.resize(kGainMapGuideBuckets + 1);
    {
        const float * T = S.steps.data() + n;
        uint32_t k = 0;
        for (uint32_t b = 0; b <= kGainMapGuideBuckets; ++b) {
            const uint32_t bits = kGainMapGuideFirstBits + (b << kGainMapGuideShift);
            float x;
            memcpy(&x, &bits, 4);
            while (k < S.maxCode && T[k + 1] <= x)
                ++k; // codes are non-decreasing along the buckets
            S.guide[b] = (uint16_t)k;
        }
    }
    buildLocator(S, isFloat);
    return cache.emplace(key, std::move(S)).first->second;
}

uint32_t gainMapLocate(const GainMapSteps & S, float x)
{
    uint32_t bits;
    memcpy(&bits, &x, 4);
    const int32_t b = (int32_t)bits, lo = (int32_t)S.locFirstBits, hi = lo + (int32_t)((S.locBuckets << S.locShift) - 1);
    const uint32_t tc = (uint32_t)(((b < lo) ? lo : ((b > hi) ? hi : b)) - lo); // as signed integers: negative x sorts below every x >= 0
    const uint32_t e = S.locator[tc >> S.locShift];
    return (e & 0xfffu) + (((tc << (32 - S.locShift)) > e) ? 1u : 0u);
}

} // namespace avifhip

// =====================================================================================================================
// gain-map computation, host side
// =====================================================================================================================
namespace avifhip {

namespace {

void convertFloat3(float v[3], const double M[9]) // avifLinearRGBConvertColorSpace
{
    const double x = v[0], y = v[1], z = v[2];
    const double r0 = M[0] * x + M[1] * y + M[2] * z, r1 = M[3] * x + M[4] * y + M[5] * z, r2 = M[6] * x + M[7] * y + M[8] * z;
    v[0] = (float)r0, v[1] = (float)r1, v[2] = (float)r2;
}

bool doubleToFractionImpl(double v, uint32_t maxNumerator, uint32_t * numerator, uint32_t * denominator) // src/utils.c:238-281
{
    if (std::isnan(v) || v < 0 || v > maxNumerator)
        return false;
    const uint32_t maxD = (v <= 1) ? UINT32_MAX : (uint32_t)floor(maxNumerator / v);
    *denominator = 1;
    uint32_t previousD = 0;
    double currentV = v - floor(v);
    for (int iter = 0; iter < 39; ++iter) {
        const double numeratorDouble = (double)(*denominator) * v;
        *numerator = (uint32_t)round(numeratorDouble);
        if (fabs(numeratorDouble - (*numerator)) == 0.0)
            return true;
        currentV = 1.0 / currentV;
        const double newD = previousD + floor(currentV) * (*denominator);
        if (newD > (double)maxD)
            return true;
        previousD = *denominator;
        *denominator = (uint32_t)newD;
        currentV -= floor(currentV);
    }
    *numerator = (uint32_t)round((double)(*denominator) * v);
    return true;
}

inline float roundHalfUp(float v) // avifRoundf
{
    return floorf(v + 0.5f);
}
inline float valueOfRatio(float sign, float r)
{
    return sign * log2f(r); // (the reference multiplies by -1.f afterwards: the same fp32 value)
}
inline int bucketOfValue(float v, float lo, float hi, int n) // avifValueToBucketIdx, :363-367
{
    v = clampf(v, lo, hi);
    const int idx = (int)roundHalfUp((v - lo) / (hi - lo) * n);
    return idx < n - 1 ? idx : n - 1;
}

// T[i] = smallest r in [minRatio, maxRatio] (fp32 order) with m(r) >= i, for i = 1 .. count - 1; m non-decreasing.  `guess(i)`
// is an estimate of T[i] (any float): the search brackets the step by galloping away from it before bisecting, which costs 3-4
// evaluations of m instead of 32 when the estimate is good; correctness never depends on it.  The gallop starts two keys from the
// estimate and quadruples (round 6: it started 64 keys away -- 8-9 evaluations per step, and the three channels' bucket and code tables
// were ~250 us of the device-resident computation's 490 us on the host between its passes; same tables, tests/tools/hostlogic.cpp)
#ifndef AVIFHIP_GALLOP_START
#define AVIFHIP_GALLOP_START 2
#endif
constexpr uint32_t kGallopStart = AVIFHIP_GALLOP_START;
template <typename Fn, typename Guess>
std::vector<float> monotoneSteps(float minRatio, float maxRatio, uint32_t count, uint32_t entries, Fn m, Guess guess)
{
    std::vector<float> T(entries, NAN);
    for (uint32_t i = 1; i < count; ++i)
        T[i] = INFINITY;
    T[0] = -INFINITY;
    const uint32_t keyHi = keyOfFloat(maxRatio);
    uint32_t lowKey = keyOfFloat(minRatio);
    const uint32_t top = m(maxRatio);
    for (uint32_t i = 1; i < count && i <= top; ++i) {
        // invariant: every key < lo has m < i; key hi has m >= i
        uint32_t lo = lowKey, hi = keyHi;
        const float g = guess(i);
        if (g == g && g > 0.0f) {
            uint32_t k = keyOfFloat(g);
            k = k < lo ? lo : (k > hi ? hi : k);
            if (m(floatOfKey(k)) >= i) { // gallop down to a key with m < i
                hi = k;
                for (uint32_t step = kGallopStart; hi > lo; step <<= 2) {
                    const uint32_t probe = (hi - lo > step) ? hi - step : lo;
                    if (m(floatOfKey(probe)) >= i) {
                        hi = probe;
                        if (probe == lo)
                            break;
                    } else {
                        lo = probe + 1;
                        break;
                    }
                }
            } else { // gallop up to a key with m >= i
                lo = k + 1;
                for (uint32_t step = kGallopStart; lo < hi; step <<= 2) {
                    const uint32_t probe = (hi - lo > step) ? lo + step : hi;
                    if (m(floatOfKey(probe)) >= i) {
                        hi = probe;
                        break;
                    }
                    lo = probe + 1;
                }
            }
        }
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (m(floatOfKey(mid)) >= i)
                hi = mid;
            else
                lo = mid + 1;
        }
        T[i] = floatOfKey(lo);
        lowKey = lo;
    }
    return T;
}

} // namespace

bool gainMapChooseMathPrimaries(int basePrimaries, int altPrimaries, int * mathPrimaries)
{
    if (basePrimaries == altPrimaries) {
        *mathPrimaries = basePrimaries;
        return true;
    }
    double baseToAlt[9], altToBase[9];
    if (!gainMapPrimariesMatrix(basePrimaries, altPrimaries, baseToAlt) || !gainMapPrimariesMatrix(altPrimaries, basePrimaries, altToBase))
        return false;
    float baseMin = 0, altMin = 0;
    for (int c = 0; c < 3; ++c) {
        float v[3] = { 0, 0, 0 };
        v[c] = 1.0f;
        convertFloat3(v, altToBase);
        for (int i = 0; i < 3; ++i)
            baseMin = (baseMin < v[i]) ? baseMin : v[i];
        v[0] = v[1] = v[2] = 0;
        v[c] = 1.0f;
        convertFloat3(v, baseToAlt);
        for (int i = 0; i < 3; ++i)
            altMin = (altMin < v[i]) ? altMin : v[i];
    }
    *mathPrimaries = (altMin <= baseMin) ? basePrimaries : altPrimaries;
    return true;
}

void gainMapYCoefficients(int primaries, float coeffs[3])
{
    const float * p = primariesOf(primaries);
    const float rX = p[0], rY = p[1], gX = p[2], gY = p[3], bX = p[4], bY = p[5], wX = p[6], wY = p[7];
    const float rZ = 1.0f - (rX + rY), gZ = 1.0f - (gX + gY), bZ = 1.0f - (bX + bY), wZ = 1.0f - (wX + wY);
    const float kr = (rY * (wX * (gY * bZ - bY * gZ) + wY * (bX * gZ - gX * bZ) + wZ * (gX * bY - bX * gY))) /
                     (wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ)));
    const float kb = (bY * (wX * (rY * gZ - gY * rZ) + wY * (gX * rZ - rX * gZ) + wZ * (rX * gY - gX * rY))) /
                     (wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ)));
    coeffs[0] = kr, coeffs[2] = kb, coeffs[1] = 1.0f - coeffs[0] - coeffs[2];
}

bool gainMapDoubleToFraction(double v, int32_t * n, uint32_t * d)
{
    uint32_t positive;
    if (!doubleToFractionImpl(fabs(v), INT32_MAX, &positive, d))
        return false;
    *n = (int32_t)positive;
    if (v < 0)
        *n *= -1;
    return true;
}
bool gainMapDoubleToUnsignedFraction(double v, uint32_t * n, uint32_t * d)
{
    return doubleToFractionImpl(v, UINT32_MAX, n, d);
}

GainMapChannelRange gainMapChannelRange(float sign, float minRatio, float maxRatio, size_t numPixels)
{
    GainMapChannelRange R;
    R.sign = sign, R.minRatio = minRatio, R.maxRatio = maxRatio;
    const float a = valueOfRatio(sign, minRatio), b = valueOfRatio(sign, maxRatio);
    R.lo = (a < b) ? a : b, R.hi = (a < b) ? b : a;
    const float bucketSize = 0.01f, maxOutliersRatio = 0.001f;
    R.maxOutliersOnEachSide = (int)roundHalfUp(numPixels * maxOutliersRatio / 2.0f);
    if ((R.hi - R.lo) <= (bucketSize * 2) || R.maxOutliersOnEachSide == 0)
        return R;
    const int byWidth = (int)ceilf((R.hi - R.lo) / bucketSize);
    R.numBuckets = byWidth < 10000 ? byWidth : 10000;
    return R;
}

std::vector<float> gainMapBucketSteps(const GainMapChannelRange & R, uint32_t * entries)
{
    uint32_t n = 1;
    while (n < (uint32_t)R.numBuckets)
        n <<= 1;
    *entries = n;
    const int nb = R.numBuckets;
    return monotoneSteps(
        R.minRatio, R.maxRatio, (uint32_t)nb, n,
        [&](float r) -> uint32_t {
            const int b = bucketOfValue(valueOfRatio(R.sign, r), R.lo, R.hi, nb);
            return (uint32_t)(R.sign > 0 ? b : nb - 1 - b);
        },
        [&](uint32_t i) -> float { // bucket b starts near value lo + (b - 0.5) / n * (hi - lo); m = b or nb - 1 - b
            const float b = R.sign > 0 ? (float)i - 0.5f : (float)(nb - 1) - (float)i + 0.5f;
            return exp2f(R.sign * (R.lo + b / (float)nb * (R.hi - R.lo)));
        });
}

void gainMapRangeWithoutOutliers(const GainMapChannelRange & R, const uint32_t * histogram, float * rangeMin, float * rangeMax)
{
    *rangeMin = R.lo, *rangeMax = R.hi;
    const int n = R.numBuckets;
    auto bucketToValue = [&](int idx) -> float { return idx * (R.hi - R.lo) / n + R.lo; }; // avifBucketIdxToValue, :369-372
    int leftOutliers = 0;
    for (int i = 0; i < n; ++i) {
        leftOutliers += (int)histogram[i];
        if (leftOutliers > R.maxOutliersOnEachSide)
            break;
        if (histogram[i] == 0)
            *rangeMin = bucketToValue(i + 1);
    }
    int rightOutliers = 0;
    for (int i = n - 1; i >= 0; --i) {
        rightOutliers += (int)histogram[i];
        if (rightOutliers > R.maxOutliersOnEachSide)
            break;
        if (histogram[i] == 0)
            *rangeMax = bucketToValue(i);
    }
}

std::vector<float> gainMapCodeSteps(const GainMapChannelRange & R, float minLog2, float maxLog2, float gamma, uint32_t depth)
{
    const uint32_t n = 1u << depth, maxCode = n - 1;
    const float maxF = (float)maxCode;
    const float range = (maxLog2 - minLog2 > 0.0f) ? maxLog2 - minLog2 : 0.0f;
    return monotoneSteps(
        R.minRatio, R.maxRatio, n, n,
        [&](float r) -> uint32_t {
            float v = valueOfRatio(R.sign, r);
            v = clampf(v, minLog2, maxLog2);
            v = powf((v - minLog2) / range, gamma);
            v = fminf(1.0f, fmaxf(0.0f, v));
            const uint32_t code = (uint32_t)(0.5f + v * maxF);
            return R.sign > 0 ? code : maxCode - code;
        },
        [&](uint32_t i) -> float { // code c starts near ((c - 0.5) / maxCode)^(1 / gamma) of the way from minLog2 to maxLog2
            const float c = R.sign > 0 ? (float)i - 0.5f : maxF - (float)i + 0.5f;
            return exp2f(R.sign * (minLog2 + powf(c / maxF, 1.0f / gamma) * range));
        });
}

} // namespace avifhip
