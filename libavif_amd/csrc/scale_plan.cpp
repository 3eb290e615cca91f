// scale_plan.cpp -- schedules of the plane scaler (scale_plan.h).  Integer-only host code; every branch cites the piece
// of the vendored libyuv scaler (third_party/libyuv/source/...) it restates.
#include "scale_plan.h"

namespace avifhip {

namespace {

enum Filter : int { NONE = 0, LINEAR = 1, BILINEAR = 2, BOX = 3 };

inline int atLeast1(int v)
{
    return v < 1 ? 1 : v;
}
inline int fixedDiv(int num, int div) // scale_common.c:472-474
{
    return (int)(((int64_t)num << 16) / div);
}
inline int fixedDiv1(int num, int div) // scale_common.c:477-479
{
    return (int)((((int64_t)num << 16) - 0x00010001) / (div - 1));
}
inline int centered(int d, int s) // CENTERSTART, scale_common.c:482
{
    return (d < 0) ? -((-d >> 1) + s) : ((d >> 1) + s);
}

// ScaleFilterReduce, scale_common.c:428-469, starting from kFilterBox (src/scale.c:21)
Filter reducedFilter(int sw, int sh, int dw, int dh)
{
    Filter f = (dw * 2 >= sw || dh * 2 >= sh) ? BILINEAR : BOX;
    if (f == BILINEAR) {
        if (sh == 1 || dh == sh || dh * 3 == sh)
            f = LINEAR;
        if (sw == 1)
            f = NONE;
    }
    if (f == LINEAR && (sw == 1 || dw == sw || dw * 3 == sw))
        f = NONE;
    return f;
}

struct Stepping
{
    int x = 0, y = 0, dx = 0, dy = 0;
};

// ScaleSlope, scale_common.c:484-553
Stepping steppingFor(int sw, int sh, int dw, int dh, Filter f)
{
    Stepping s;
    if (dw == 1 && sw >= 32768)
        dw = sw;
    if (dh == 1 && sh >= 32768)
        dh = sh;
    if (f == BOX) {
        s.dx = fixedDiv(sw, dw), s.dy = fixedDiv(sh, dh);
        return s;
    }
    if (f == NONE) {
        s.dx = fixedDiv(sw, dw), s.dy = fixedDiv(sh, dh);
        s.x = centered(s.dx, 0), s.y = centered(s.dy, 0);
        return s;
    }
    if (dw <= sw) {
        s.dx = fixedDiv(sw, dw);
        s.x = centered(s.dx, -32768);
    } else if (sw > 1 && dw > 1) {
        s.dx = fixedDiv1(sw, dw);
    }
    if (f == LINEAR) {
        s.dy = fixedDiv(sh, dh);
        s.y = s.dy >> 1;
    } else if (dh <= sh) {
        s.dy = fixedDiv(sh, dh);
        s.y = centered(s.dy, -32768);
    } else if (sh > 1 && dh > 1) {
        s.dy = fixedDiv1(sh, dh);
    }
    return s;
}

// scale_any.c:19-85: near / far source index of the 2x upsamplers along one axis; `lastIsEdge`: the last destination
// index is unfiltered (always for columns; for rows only when the destination height is even, scale.c:525-527)
void upsample2Axis(int srcN, int dstN, bool lastIsEdge, std::vector<int32_t> & nearIdx, std::vector<int32_t> & farIdx)
{
    for (int k = 0; k < dstN; ++k) {
        const int n = k >> 1;
        int f = (k & 1) ? n + 1 : n - 1;
        if (k == 0 || (k == dstN - 1 && lastIsEdge) || f < 0 || f > srcN - 1)
            f = n;
        nearIdx[k] = n, farIdx[k] = f;
    }
}

void filteredColumns(ScaleSchedule & S, int x, int dx, int sw)
{
    for (size_t i = 0; i < S.colA.size(); ++i, x += dx) {
        const int xi = x >> 16;
        S.colA[i] = xi > sw - 1 ? sw - 1 : xi;
        S.colB[i] = x & 0xffff;
    }
}

} // namespace

static ScaleSchedule makeScaleScheduleImpl(int sw, int sh, int dw, int dh, bool wide);

ScaleSchedule makeScaleSchedule(int sw, int sh, int dw, int dh, bool wide)
{
    ScaleSchedule S = makeScaleScheduleImpl(sw, sh, dw, dh, wide);
    if (S.mode == SCALE_BOX && !wide && !S.colB.empty() && !S.rowB.empty()) {
        const int n = S.colB[0];
        bool exact = (n == 4 || n == 8) && sw == n * dw && sh == n * dh;
        for (size_t i = 0; i < S.colA.size() && exact; ++i)
            exact = S.colA[i] == n * (int)i && S.colB[i] == n;
        for (size_t j = 0; j < S.rowA.size() && exact; ++j)
            exact = S.rowA[j] == n * (int)j && S.rowB[j] == n;
        S.exactBox = exact ? n : 0;
    }
    return S;
}

static ScaleSchedule makeScaleScheduleImpl(int sw, int sh, int dw, int dh, bool wide)
{
    ScaleSchedule S;
    S.colA.assign(dw, 0), S.colB.assign(dw, 0);
    S.rowA.assign(dh, 0), S.rowB.assign(dh, 0), S.rowF.assign(dh, 0);
    const Filter f = reducedFilter(sw, sh, dw, dh);
    const bool copy = dw == sw && dh == sh;
    const bool vertical = dw == sw && f != BOX;
    const bool box = f == BOX && dh * 2 < sh;
    // ScalePlane_12 (16-bit samples) looks for the 2x cases before everything else (scale.c:966-977); ScalePlane and
    // ScalePlane_16 only after the copy / vertical / box cases (:851-884)
    const bool up2Allowed = wide || !(copy || vertical || box);
    const bool up2w = (dw + 1) / 2 == sw, up2h = (dh + 1) / 2 == sh;
    if (up2Allowed && up2w && f == LINEAR) { // ScalePlaneUp2_Linear and twins, scale.c:464-495: nearest rows
        S.mode = SCALE_UP2;
        upsample2Axis(sw, dw, true, S.colA, S.colB);
        if (dh == 1) {
            S.rowA[0] = S.rowB[0] = (sh - 1) / 2;
        } else {
            const int dy = fixedDiv(sh - 1, dh - 1);
            int y = (1 << 15) - 1;
            for (int j = 0; j < dh; ++j, y += dy)
                S.rowA[j] = S.rowB[j] = y >> 16;
        }
        return S;
    }
    if (up2Allowed && up2w && up2h && (f == BILINEAR || f == BOX)) { // ScalePlaneUp2_Bilinear and twins, :500-528
        S.mode = SCALE_UP2;
        S.doubling = true;
        upsample2Axis(sw, dw, true, S.colA, S.colB);
        upsample2Axis(sh, dh, !(dh & 1), S.rowA, S.rowB);
        return S;
    }
    if (copy) {
        for (int i = 0; i < dw; ++i)
            S.colA[i] = i;
        for (int j = 0; j < dh; ++j)
            S.rowA[j] = j;
        return S;
    }
    if (vertical) { // ScalePlaneVertical, scale_common.c:348-386 via scale.c:857-873
        S.mode = SCALE_DOWN;
        int y = 0, dy = 0;
        if (dh <= sh) {
            dy = fixedDiv(sh, dh);
            y = centered(dy, -32768);
        } else if (sh > 1 && dh > 1) {
            dy = fixedDiv1(sh, dh);
        }
        const int maxY = (sh > 1) ? ((sh - 1) << 16) - 1 : 0;
        for (int i = 0; i < dw; ++i)
            S.colA[i] = i; // colB = 0: a zero fraction leaves the column blend an identity
        for (int j = 0; j < dh; ++j, y += dy) {
            if (y > maxY)
                y = maxY;
            S.rowA[j] = y >> 16;
            S.rowF[j] = (f != NONE) ? ((y >> 8) & 255) : 0;
            S.rowB[j] = S.rowF[j] ? S.rowA[j] + 1 : S.rowA[j];
        }
        return S;
    }
    if (box) { // ScalePlaneBox / _16, scale.c:153-256
        S.mode = SCALE_BOX;
        Stepping s = steppingFor(sw, sh, dw, dh, BOX);
        const int maxY = sh << 16;
        for (int j = 0; j < dh; ++j) {
            const int iy = s.y >> 16;
            s.y += s.dy;
            if (s.y > maxY)
                s.y = maxY;
            S.rowA[j] = iy;
            S.rowB[j] = atLeast1((s.y >> 16) - iy);
        }
        if (s.dx & 0xffff) { // ScaleAddCols2: boxes of minboxwidth or minboxwidth + 1 columns
            for (int i = 0; i < dw; ++i) {
                const int ix = s.x >> 16;
                s.x += s.dx;
                S.colA[i] = ix;
                S.colB[i] = atLeast1((s.x >> 16) - ix);
            }
        } else { // ScaleAddCols1: whole-number step
            const int bw = atLeast1(s.dx >> 16);
            int ix = s.x >> 16;
            for (int i = 0; i < dw; ++i, ix += bw)
                S.colA[i] = ix, S.colB[i] = bw;
        }
        return S;
    }
    if (f != NONE && dh > sh) { // ScalePlaneBilinearUp / _16, scale.c:384-459
        S.mode = SCALE_UP;
        Stepping s = steppingFor(sw, sh, dw, dh, f);
        filteredColumns(S, s.x, s.dx, sw);
        const int maxY = (sh - 1) << 16;
        int y = s.y > maxY ? maxY : s.y;
        // the reference keeps two buffers of horizontally filtered rows and refills the older one whenever the integer row
        // advances; `held` follows which source row each buffer contains, `front` which buffer is "rowptr"
        int yi = y >> 16;
        int next = yi; // source row the reference's `src` pointer addresses
        int held[2], front = 0;
        held[0] = next;
        if (sh > 1)
            ++next;
        held[1] = next;
        if (sh > 2)
            ++next;
        int last = yi;
        for (int j = 0; j < dh; ++j, y += s.dy) {
            yi = y >> 16;
            if (yi != last) {
                if (y > maxY) {
                    y = maxY;
                    yi = y >> 16;
                    next = yi;
                }
                if (yi != last) {
                    held[front] = next;
                    front ^= 1;
                    last = yi;
                    if ((y + 65536) < maxY)
                        ++next;
                }
            }
            S.rowA[j] = held[front];
            S.rowB[j] = held[front ^ 1];
            S.rowF[j] = (f == LINEAR) ? 0 : ((y >> 8) & 255);
        }
        return S;
    }
    if (f != NONE) { // ScalePlaneBilinearDown / _16, scale.c:259-381
        S.mode = SCALE_DOWN;
        Stepping s = steppingFor(sw, sh, dw, dh, f);
        filteredColumns(S, s.x, s.dx, sw);
        const int maxY = (sh - 1) << 16;
        int y = s.y > maxY ? maxY : s.y;
        for (int j = 0; j < dh; ++j) {
            S.rowA[j] = y >> 16;
            S.rowF[j] = (f == LINEAR) ? 0 : ((y >> 8) & 255);
            S.rowB[j] = S.rowF[j] ? S.rowA[j] + 1 : S.rowA[j];
            y += s.dy;
            if (y > maxY)
                y = maxY;
        }
        return S;
    }
    // ScalePlaneSimple / _16, scale.c:770-826
    Stepping s = steppingFor(sw, sh, dw, dh, NONE);
    const bool doubling = sw * 2 == dw && s.x < 0x8000; // ScaleColsUp2
    for (int i = 0; i < dw; ++i, s.x += s.dx)
        S.colA[i] = doubling ? (i >> 1) : (s.x >> 16);
    for (int j = 0; j < dh; ++j, s.y += s.dy)
        S.rowA[j] = s.y >> 16;
    return S;
}

} // namespace avifhip
