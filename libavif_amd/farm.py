"""Tile farm: independent units of reformat work (tiles of an AVIF grid, frames of a sequence) spread over the GPUs of
one node -- one process per GPU, contiguous blocks of units per rank, NO data-path collective.

Grid semantics (reference src/read.c:1823-1877 stitches decoded tiles into one canvas which the caller then converts
as a whole): a tile job converts its rectangle of the stitched canvas with the chroma edge rules of
src/reformat.c:768,784 evaluated against the CANVAS, so the union of all tile jobs equals the whole-canvas
conversion byte for byte, seams included.  The rectangle converter is injected: `HipRectConverter` (the product:
device-resident canvas, one batched launch per rank) or, in the CPU-only tests, the oracle's rectangle entry point.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Callable, Iterable, List, Sequence, Tuple

from . import abi

Rect = Tuple[int, int, int, int]  # x, y, width, height


def grid_rects(canvas_w: int, canvas_h: int, tile_w: int, tile_h: int) -> List[Rect]:
    """Tiles of an AVIF grid in row-major order; the last column/row is cropped to the canvas
    (ISO 23008-12 grid derivation, reference src/read.c:1823-1877)."""
    if tile_w <= 0 or tile_h <= 0:
        raise ValueError("tile size must be positive")
    rects = []
    for y in range(0, canvas_h, tile_h):
        for x in range(0, canvas_w, tile_w):
            rects.append((x, y, min(tile_w, canvas_w - x), min(tile_h, canvas_h - y)))
    return rects


def shard(n_units: int, rank: int, world: int) -> List[int]:
    """Units of this rank: a contiguous, balanced block (sizes differ by at most one).  Contiguous because the tiles of a grid
    are listed row by row: a rank's block is then whole tile rows (64 tiles on 8 ranks: one row each), which the library moves
    between host and device as long full-width rows instead of many short ones ("contiguous blocks to keep host
    staging sequential")."""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def validate_rects(rects: Iterable[Rect], yuv_format: int) -> None:
    """Rectangle origins must sit on chroma sample boundaries (src/avif.c:335-337)."""
    sx, sy = abi.chroma_shifts(yuv_format)
    for (x, y, w, h) in rects:
        if (x & sx) or (y & sy):
            raise ValueError(f"rectangle origin ({x},{y}) is not aligned to the chroma grid")
        if w <= 0 or h <= 0:
            raise ValueError("empty rectangle")


def convert_shard(canvas: abi.HostYUV, rgb_canvas: abi.HostRGB, rects: Sequence[Rect], rank: int, world: int,
                  convert_rects: Callable[[abi.HostYUV, abi.HostRGB, Sequence[Rect]], None]) -> List[int]:
    """Converts this rank's tiles of the grid into rgb_canvas (other tiles are left untouched); returns their indices."""
    validate_rects(rects, canvas.struct.yuvFormat)
    mine = shard(len(rects), rank, world)
    if mine:
        convert_rects(canvas, rgb_canvas, [rects[t] for t in mine])
    return mine


class HipRectConverter:
    """The product path: avifhipImageYUVToRGBRects -- the library uploads only what this rank's rectangles need (their plane
    samples plus the one-sample chroma halo), converts them with the canvas's edge rules, and downloads only the rectangles
    into the host canvas, uploads / kernels / downloads of successive rectangles overlapping.  `bytes_up` / `bytes_down` hold
    what the last call moved over the host link."""

    def __init__(self):
        from . import device, native

        self.device, self.native = device, native
        self.lib = native.load()
        if self.lib.avifhipDeviceCount() <= 0:
            raise native.AvifHipError("HipRectConverter: no HIP device visible (there is no CPU fallback)")

    def __call__(self, canvas: abi.HostYUV, rgb_canvas: abi.HostRGB, rects: Sequence[Rect]) -> None:
        n = len(rects)
        crops = (abi.avifCropRect * n)(*[abi.avifCropRect(x, y, w, h) for (x, y, w, h) in rects])
        self.native.check(self.lib.avifhipImageYUVToRGBRects(canvas.struct, rgb_canvas.struct, crops, n), "avifhipImageYUVToRGBRects")
        up, down = C.c_uint64(0), C.c_uint64(0)
        self.lib.avifhipLastTransferBytes(C.byref(up), C.byref(down))
        self.bytes_up, self.bytes_down = int(up.value), int(down.value)


def planned_transfers(canvas: abi.HostYUV, rgb_canvas: abi.HostRGB, rects: Sequence[Rect]) -> Tuple[int, int]:
    """(bytes up, bytes down) avifhipImageYUVToRGBRects moves over the host link for these rectangles; needs no GPU."""
    from . import native

    lib = native.load()
    n = len(rects)
    crops = (abi.avifCropRect * n)(*[abi.avifCropRect(x, y, w, h) for (x, y, w, h) in rects])
    up, down = C.c_uint64(0), C.c_uint64(0)
    native.check(lib.avifhipPlanRectTransfers(canvas.struct, rgb_canvas.struct, crops, n, C.byref(up), C.byref(down)), "avifhipPlanRectTransfers")
    return int(up.value), int(down.value)


def timed_region(run: Callable[[], None], sync: Callable[[], None], dist=None) -> float:
    """Wall time of run() bracketed by a barrier + device sync on both sides, MAX over ranks (bench contract)."""
    sync()
    if dist is not None and dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    run()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.is_initialized():
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    return elapsed
