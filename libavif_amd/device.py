"""Device-resident images: avifImage / avifRGBImage structs whose buffers live in HBM.

Buffers are allocated through the library's own C ABI (avifhipDeviceAlloc), rows padded to 256 bytes, so the
tests and the benchmark need neither torch nor HIP headers for the data path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import abi, native


def _pitch(width_bytes: int, align: int = 256) -> int:
    return (width_bytes + align - 1) // align * align


class DeviceBuffer:
    def __init__(self, nbytes: int):
        lib = native.load()
        self.nbytes = max(int(nbytes), 1)
        self.ptr = lib.avifhipDeviceAlloc(self.nbytes)
        if not self.ptr:
            raise native.AvifHipError(f"device allocation of {nbytes} bytes failed: {lib.avifhipLastError().decode()}")

    def upload(self, host: np.ndarray) -> None:
        host = np.ascontiguousarray(host)
        native.check(native.load().avifhipCopyToDevice(self.ptr, host.ctypes.data, host.nbytes), "H2D copy")

    def download(self, nbytes: Optional[int] = None) -> np.ndarray:
        n = self.nbytes if nbytes is None else nbytes
        out = np.empty(n, dtype=np.uint8)
        native.check(native.load().avifhipCopyToHost(out.ctypes.data, self.ptr, n), "D2H copy")
        return out

    def memset(self, value: int) -> None:
        native.check(native.load().avifhipDeviceMemset(self.ptr, value, self.nbytes), "device memset")

    def free(self) -> None:
        if self.ptr:
            native.load().avifhipDeviceFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceYUV:
    """Device twin of a HostYUV (same geometry / CICP), planes uploaded with 256-byte row pitch."""

    def __init__(self, host: abi.HostYUV, upload: bool = True, tight: bool = False):
        self.host = host
        hs = host.struct
        self.struct = abi.avifImage()
        C.memmove(C.byref(self.struct), C.byref(hs), C.sizeof(abi.avifImage))
        bps = 2 if hs.depth > 8 else 1
        cw, ch = abi.chroma_dims(hs.width, hs.height, hs.yuvFormat)
        geo = [(hs.width, hs.height), (cw, ch), (cw, ch), (hs.width, hs.height)]
        self.buffers: list = [None] * 4
        self.pitch = [0] * 4
        self.geo = geo
        for p in range(4):
            src = host.alpha if p == 3 else host.planes[p]
            if src is None:
                if p < 3:
                    self.struct.yuvPlanes[p] = None
                    self.struct.yuvRowBytes[p] = 0
                else:
                    self.struct.alphaPlane = None
                    self.struct.alphaRowBytes = 0
                continue
            w, h = geo[p]
            pitch = w * bps if tight else _pitch(w * bps)
            buf = DeviceBuffer(pitch * h)
            if upload:
                staged = np.zeros((h, pitch), dtype=np.uint8)
                staged[:, : w * bps] = src[:h, : w * bps]
                buf.upload(staged)
            self.buffers[p] = buf
            self.pitch[p] = pitch
            if p < 3:
                self.struct.yuvPlanes[p] = buf.ptr
                self.struct.yuvRowBytes[p] = pitch
            else:
                self.struct.alphaPlane = buf.ptr
                self.struct.alphaRowBytes = pitch

    def upload(self) -> None:
        """Copies the host twin's planes into the device buffers again (new samples in the same buffers)."""
        hs = self.host.struct
        bps = 2 if hs.depth > 8 else 1
        for p in range(4):
            src = self.host.alpha if p == 3 else self.host.planes[p]
            if src is None or self.buffers[p] is None:
                continue
            w, h = self.geo[p]
            staged = np.zeros((h, self.pitch[p]), dtype=np.uint8)
            staged[:, : w * bps] = src[:h, : w * bps]
            self.buffers[p].upload(staged)

    def download_into_host(self) -> None:
        """Copies the device planes back into the host twin (RGB->YUV results)."""
        hs = self.host.struct
        bps = 2 if hs.depth > 8 else 1
        for p in range(4):
            dst = self.host.alpha if p == 3 else self.host.planes[p]
            if dst is None or self.buffers[p] is None:
                continue
            w, h = self.geo[p]
            raw = self.buffers[p].download(self.pitch[p] * h).reshape(h, self.pitch[p])
            dst[:h, : w * bps] = raw[:, : w * bps]


class DeviceRGB:
    def __init__(self, host: abi.HostRGB, upload: bool = False, tight: bool = False):
        self.host = host
        hs = host.struct
        self.struct = abi.avifRGBImage()
        C.memmove(C.byref(self.struct), C.byref(hs), C.sizeof(abi.avifRGBImage))
        self.width_bytes = hs.width * abi.rgb_pixel_size(hs.format, hs.depth)
        self.pitch = self.width_bytes if tight else _pitch(self.width_bytes)
        self.buffer = DeviceBuffer(self.pitch * hs.height)
        if upload:
            staged = np.zeros((hs.height, self.pitch), dtype=np.uint8)
            staged[:, : self.width_bytes] = host.pixels[:, : self.width_bytes]
            self.buffer.upload(staged)
        self.struct.pixels = self.buffer.ptr
        self.struct.rowBytes = self.pitch

    def upload(self) -> None:
        hs = self.host.struct
        staged = np.zeros((hs.height, self.pitch), dtype=np.uint8)
        staged[:, : self.width_bytes] = self.host.pixels[:, : self.width_bytes]
        self.buffer.upload(staged)

    def download_into_host(self) -> None:
        hs = self.host.struct
        raw = self.buffer.download(self.pitch * hs.height).reshape(hs.height, self.pitch)
        self.host.pixels[:, : self.width_bytes] = raw[:, : self.width_bytes]
