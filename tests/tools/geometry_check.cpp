// geometry_check.cpp -- test tool (not shipped).  Walks the launch geometry of the wave-private tiled kernels on the CPU with the
// PRODUCT's own index arithmetic (libavif_amd/csrc/tile_geom.h: pkGeometry on the host side, pkTileOf / pkPlaceOf / blockRemap as the
// kernels evaluate them -- constexpr, so the same functions compile for the host) and counts how often every wave-tile of a job is
// visited: exactly once each, nothing outside, for every combination of image size and tuning the launchers can form.
// Built by tests/test_host_plans.py with `hipcc -x hip --cuda-host-only` (host code only; no GPU needed).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "tile_geom.h"

using namespace avifhip;
using namespace avifhip::tile;

namespace {

// 0 = every wave-tile of a w4 x h2 job visited exactly once; otherwise a code saying what went wrong
int checkPk(uint32_t w4, uint32_t h2, uint32_t count, uint32_t pkStrips, uint32_t wavesXLog2, uint32_t chunkRows, std::vector<uint8_t> & seenTile,
            std::vector<uint8_t> & seenPlace, bool transposed = false, uint32_t shiftStrips = 0)
{
    TileLaunch L;
    memset(&L, 0, sizeof(L));
    L.count = count, L.pkStrips = pkStrips, L.wavesXLog2 = wavesXLog2, L.chunkRows = chunkRows;
    L.shiftStrips = shiftStrips; // quarter turns of single images: the tile grid starts that many strips above the rectangle
    L.transposed = transposed; // quarter turns: tiles numbered down the columns, one tile column per XCD chunk
    uint32_t nsw = 0, blocks = 0;
    PkGeom g;
    pkGeometry(L, w4, h2, &nsw, &g, &blocks);
    if (nsw != 2 && nsw != 4)
        return 1;
    if (blocks < g.nTiles || blocks > 0x7fffffffu)
        return 2;
    const uint32_t bands = (w4 + 255u) / 256u, strips = h2 / 2, stripRuns = (strips + nsw - 1) / nsw;
    seenTile.assign(g.nTiles, 0);
    seenPlace.assign((size_t)bands * stripRuns, 0);
    for (uint32_t b = 0; b < blocks; ++b) {
        const uint32_t tile = pkTileOf(b, g);
        if (tile >= g.nTiles)
            continue; // padding of the chunked order: the workgroup leaves at once
        if (seenTile[tile]++)
            return 3; // two workgroups took the same tile
        for (uint32_t wave = 0; wave < 4; ++wave) {
            const PkPlace p = pkPlaceOf(tile, wave, g, nsw);
            if (p.band * 256u >= w4 || 2u * p.strip0 >= h2)
                continue; // the kernels' "no work for this wave" exit
            if (p.strip0 % nsw)
                return 4;
            const size_t at = (size_t)(p.strip0 / nsw) * bands + p.band;
            if (p.band >= bands || at >= seenPlace.size())
                return 5;
            if (seenPlace[at]++)
                return 6; // two waves own the same strips
        }
    }
    for (uint8_t s : seenTile)
        if (s != 1)
            return 7;
    for (uint8_t s : seenPlace)
        if (s != 1)
            return 8; // strips nobody converts
    return 0;
}

} // namespace

extern "C" {

// tile_shared.h linkHalo: a job's neighbours in the order its own planes were handed to the kernels (`swapped`: the V plane feeds the
// pixel's first colour channel), absent neighbours replaced by the job's own planes, the sides as bits.  0 = as specified.
int geomCheckHaloLink(int swapped, int above, int below, int left, int right)
{
    static uint8_t arena[64];
    const uint8_t *p1[9], *p2[9];
    for (int d = 0; d < 9; ++d)
        p1[d] = arena + 2 * d, p2[d] = arena + 2 * d + 1; // eighteen distinct addresses
    TileArgs T;
    memset(&T, 0, sizeof(T));
    T.u = swapped ? p2[0] : p1[0], T.v = swapped ? p1[0] : p2[0];
    T.planesSwapped = swapped ? 1u : 0u; // (distillArgs' own record: linkHalo no longer compares pointers -- a tile whose U and V planes are one buffer)
    linkHalo(T, p1, p2, above != 0, below != 0, left != 0, right != 0);
    const uint32_t sides = (above ? HALO_ABOVE : 0u) | (below ? HALO_BELOW : 0u) | (left ? HALO_LEFT : 0u) | (right ? HALO_RIGHT : 0u);
    if (T.haloSides != sides)
        return 1;
    for (int d = 0; d < 9; ++d) {
        const int v = d / 3, h = d % 3;
        const bool there = (v == 0 || (v == 1 ? above : below)) && (h == 0 || (h == 1 ? left : right));
        const uint8_t * wantU = there ? (swapped ? p2[d] : p1[d]) : T.u;
        const uint8_t * wantV = there ? (swapped ? p1[d] : p2[d]) : T.v;
        if (T.halo.at[d].u != wantU || T.halo.at[d].v != wantV)
            return 2 + d;
    }
    return 0;
}


// ... and with the job's own U and V planes being ONE buffer (a gray image stored as 4:2:0 with shared chroma): the neighbours, whose planes
// are distinct, must still follow the recorded order
int geomCheckHaloLinkSharedPlanes(int swapped)
{
    static uint8_t arena[64];
    const uint8_t *p1[9], *p2[9];
    for (int d = 0; d < 9; ++d)
        p1[d] = arena + 2 * d, p2[d] = arena + 2 * d + 1;
    p2[0] = p1[0];
    TileArgs T;
    memset(&T, 0, sizeof(T));
    T.u = T.v = p1[0];
    T.planesSwapped = swapped ? 1u : 0u;
    linkHalo(T, p1, p2, true, true, true, true);
    for (int d = 1; d < 9; ++d)
        if (T.halo.at[d].u != (swapped ? p2[d] : p1[d]) || T.halo.at[d].v != (swapped ? p1[d] : p2[d]))
            return 2 + d;
    return 0;
}

// batches along the rows of a canvas (tile_geom.h PkGeom::canvasColumns): every (job, tile) of every job exactly once over the launch's grid
int geomCheckCanvasOrder(uint32_t w4, uint32_t h2, uint32_t columns, uint32_t rows, uint32_t pkStrips, uint32_t wavesXLog2)
{
    TileLaunch L;
    memset(&L, 0, sizeof(L));
    static const TileArgs table[1] = {};
    L.table = table, L.count = columns * rows, L.canvasColumns = columns;
    L.pkStrips = pkStrips, L.wavesXLog2 = wavesXLog2, L.chunkRows = 1;
    uint32_t nsw = 0, blocks = 0;
    PkGeom g;
    pkGeometry(L, w4, h2, &nsw, &g, &blocks);
    if (columns > 1 && g.canvasColumns != columns)
        return 1;
    const dim3 grid = pkBatchGrid(g, blocks, L.count);
    std::vector<uint32_t> seen((size_t)L.count * g.nTiles, 0);
    for (uint32_t bz = 0; bz < grid.z; ++bz)
        for (uint32_t by = 0; by < grid.y; ++by)
            for (uint32_t bx = 0; bx < grid.x; ++bx) {
                const BatchWhere w = pkBatchWhereOf(bx, by, bz, g);
                if (w.job >= L.count)
                    return 2;
                if (w.tile >= g.nTiles)
                    continue; // (padding of the chunked order: leaves at once)
                ++seen[(size_t)w.job * g.nTiles + w.tile];
            }
    for (uint32_t v : seen)
        if (v != 1)
            return 3;
    return 0;
}

// a turned launch whose tile grid starts `shiftStrips` strips above the rectangle (tile_impl.h launchSoloMapped): every strip still once
int geomCheckPkShifted(uint32_t w4, uint32_t h2, uint32_t pkStrips, uint32_t shiftStrips)
{
    std::vector<uint8_t> a, b;
    return checkPk(w4, h2, 1, pkStrips, 0, 1, a, b, true, shiftStrips);
}

int geomCheckPk(uint32_t w4, uint32_t h2, uint32_t count, uint32_t pkStrips, uint32_t wavesXLog2, uint32_t chunkRows)
{
    std::vector<uint8_t> a, b;
    const int rowMajor = checkPk(w4, h2, count, pkStrips, wavesXLog2, chunkRows, a, b);
    return rowMajor ? rowMajor : checkPk(w4, h2, count, pkStrips, wavesXLog2, chunkRows, a, b, true);
}

// every tuning (strips 0/2/4, waves side by side 1/2/4, chunk rows 0..15) x job counts {1, 3, 64} over the widths around every listed band count
// and every even height up to maxH2; returns the number of failing combinations and describes the first one in `first`
// (w4, h2, count, pkStrips, wavesXLog2, chunkRows, code)
uint64_t geomSweepPk(uint32_t maxH2, uint32_t * first, uint64_t * casesOut)
{
    static const uint32_t bandCounts[] = { 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64 };
    static const uint32_t chunkRows[] = { 0, 1, 2, 3, 8, 15 };
    static const uint32_t counts[] = { 1, 3, 64 };
    std::vector<uint8_t> a, b;
    uint64_t bad = 0, cases = 0;
    for (uint32_t n : bandCounts)
        for (uint32_t w4 : { 256u * n - 252u, 256u * n - 4u, 256u * n })
            for (uint32_t h2 = 2; h2 <= maxH2; h2 += 2)
                for (uint32_t count : counts)
                    for (uint32_t strips : { 0u, 2u, 4u })
                        for (uint32_t wxl = 0; wxl <= 2; ++wxl)
                            for (uint32_t cr : chunkRows) {
                                int code = checkPk(w4, h2, count, strips, wxl, cr, a, b);
                                if (!code && wxl == 0 && cr <= 1) { // the quarter-turn order (waves stacked, one tile column per chunk or raster)
                                    code = checkPk(w4, h2, count, strips, wxl, cr, a, b, true);
                                    ++cases;
                                }
                                ++cases;
                                if (code && !bad++) {
                                    const uint32_t f[7] = { w4, h2, count, strips, wxl, cr, (uint32_t)code };
                                    memcpy(first, f, sizeof(f));
                                }
                            }
    *casesOut = cases;
    return bad;
}

// blockRemap (the cooperative kernels' order) is a permutation of [0, n) for every grid size up to maxN, with and without the XCD bands
uint64_t geomSweepRemap(uint32_t maxN)
{
    std::vector<uint8_t> seen;
    uint64_t bad = 0;
    for (uint32_t n = 1; n <= maxN; ++n)
        for (int bands = 0; bands < 2; ++bands) {
            seen.assign(n, 0);
            bool ok = true;
            for (uint32_t b = 0; b < n && ok; ++b) {
                const uint32_t r = blockRemap(b, n, bands != 0);
                ok = r < n && !seen[r]++;
            }
            bad += !ok;
        }
    return bad;
}

} // extern "C"
