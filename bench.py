#!/usr/bin/env python3
"""bench.py -- megapixels/s of libavif's YUV->RGB reformat hot path on MI355X.

A "step" is one pass of the hot path over one synthetic 8K frame: 7680x4320 8-bit YUV 4:2:0, BT.709 limited
range -> RGBA8 with bilinear chroma upsampling (BASELINE.json configs[1]), planes and pixels resident in HBM, converted
with the API defaults (rgb.avoidLibYUV = 0): the reference's INTEGER path, i.e. byte-identical to what a libavif built
with libyuv computes (I420ToARGBMatrixFilter, kFilterBilinear).  The fp32 path (avoidLibYUV = 1, byte-identical to a
libavif built without libyuv) is timed next to it and reported inside "roofline" as "fp32_path".
Steps cycle over several distinct frames so the working set (>700 MB) exceeds the 256 MB Infinity Cache: every
byte of every step comes from and goes to HBM.  Frames are independent units of work (sequence frames / grid
tiles), so consecutive steps are issued round-robin on a few HIP streams and overlap each other's head and tail.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU): every rank converts its own frames,
there is no data-path collective ("scaling": "weak"); ranks only meet in the barrier around the timed region and
the MAX over rank times.

The "roofline" object describes the dominant kernel alone: algorithmic bytes per launch (5.5 B/pixel: each input
sample read once, each output byte written once) divided by the kernel's average duration, measured with HIP events
on the launch stream over back-to-back launches that cycle over the same distinct frames (single stream, no
overlap between launches).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WIDTH, HEIGHT = 7680, 4320
FRAMES_IN_FLIGHT = 4  # distinct frame buffers cycled by the timed loop (4 x 182 MB > Infinity Cache)
STREAMS = 2           # independent frames overlap head/tail on this many HIP streams
ALGORITHMIC_BYTES_PER_PIXEL = 5.5  # 1.5 B read (Y + U/4 + V/4) + 4 B written (RGBA8), SURVEY.md 8d
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=STREAMS)
    ap.add_argument("--arithmetic", choices=("integer", "fp32"), default="integer",
                    help="integer: API defaults, libyuv's fixed point (default); fp32: rgb.avoidLibYUV = 1, libavif's built-in path")
    return ap.parse_args()


def cpu_baseline(abi, synth, seconds: float):
    """The reference's own CPU path on the host cores, single thread (the reference forces 1 thread for 4:2:0
    bilinear, src/reformat.c:1684-1688), on the same 8K workload; bounded to ~`seconds` of CPU work."""
    ref_path = ROOT / "oracle" / "_ref" / "libavif_ref.so"
    port_path = ROOT / "oracle" / "liboracle.so"
    img = abi.make_yuv(WIDTH, HEIGHT, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, abi.AVIF_MATRIX_COEFFICIENTS_BT709)
    synth.fill_yuv(img, 0x12345678)
    rgb = abi.make_rgb(WIDTH, HEIGHT, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=True)
    if ref_path.exists():
        lib, kind, fn_name = C.CDLL(os.fspath(ref_path), mode=os.RTLD_LOCAL), "reference", "avifImageYUVToRGB"
    elif port_path.exists():
        lib, kind, fn_name = C.CDLL(os.fspath(port_path)), "port", "oracleImageYUVToRGB"
    else:
        return None
    fn = getattr(lib, fn_name)
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(abi.avifImage), C.POINTER(abi.avifRGBImage)]
    frames, t_total, best = 0, 0.0, float("inf")
    while t_total < seconds and frames < 40:
        t0 = time.perf_counter()
        if fn(img.struct, rgb.struct) != 0:
            return None
        dt = time.perf_counter() - t0
        t_total += dt
        best = min(best, dt)
        frames += 1
    mp = WIDTH * HEIGHT / 1e6
    out_extra = {}
    # Beside it, when the image ships one: a libavif BUILT WITH LIBYUV (Pillow's bundled binary) on the same frame -- the CPU
    # counterpart of the integer path the headline measures.  Reported as an extra field; `value` stays the from-source reference.
    try:
        import glob as _glob

        import PIL as _pil

        cands = _glob.glob(os.path.join(os.path.dirname(_pil.__file__) + ".libs", "libavif*.so*")) + \
            _glob.glob(os.path.join(os.path.dirname(os.path.dirname(_pil.__file__)), "pillow.libs", "libavif*.so*"))
        if cands:
            plib = C.CDLL(cands[0], mode=os.RTLD_LOCAL)
            pfn = plib.avifImageYUVToRGB
            pfn.restype, pfn.argtypes = C.c_int, [C.POINTER(abi.avifImage), C.POINTER(abi.avifRGBImage)]
            prgb = abi.make_rgb(WIDTH, HEIGHT, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False)
            pbest = float("inf")
            for _ in range(8):
                t0 = time.perf_counter()
                if pfn(img.struct, prgb.struct) != 0:
                    raise RuntimeError("conversion failed")
                pbest = min(pbest, time.perf_counter() - t0)
            out_extra["libyuv_build"] = {"value": round(mp / pbest, 1), "unit": "megapixels/s", "cores": 1,
                                         "sample": "best of 8 x the same frame, libavif 1.4.1 + libyuv 1922 (Pillow's binary), API defaults"}
    except Exception:
        pass
    return {**out_extra, "value": round(mp * frames / t_total, 2), "unit": "megapixels/s", "cores": 1, "kind": kind,
            "sample": f"{frames} x 7680x4320 8-bit 4:2:0 BT.709 limited -> RGBA8 bilinear frames, libavif built-in float path "
                      f"(the reference compiled from its own sources has no libyuv: avoidLibYUV=1 arithmetic; maxThreads=1: the reference "
                      f"runs 4:2:0 bilinear single-threaded), {t_total:.1f} s of CPU; "
                      f"best frame {mp / best:.1f} MP/s",
            "best_value": round(mp / best, 2)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    dist = None
    torch = None
    # (AVIFHIP_BENCH_FORCE_DIST=1: take the multi-rank code path -- torch + RCCL process group, barriers, max-over-ranks --
    # with a single rank too, to exercise it on a one-GPU box)
    if world > 1 or os.environ.get("AVIFHIP_BENCH_FORCE_DIST") == "1":
        # torch first: its bundled HIP runtime must be the one libavifhip.so binds to (same SONAME)
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on STDOUT when its communicator comes up; the contract is ONE JSON line there, so
        # stdout points at stderr while the process group initialises and runs its first collective
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from libavif_amd import abi, device, native, synth

    lib = native.load()
    if lib.avifhipDeviceCount() <= 0:
        raise SystemExit("bench.py: no HIP device visible -- there is no CPU fallback for the product path")
    native.check(lib.avifhipSetDevice(local_rank if world > 1 else 0), "avifhipSetDevice")
    lib.avifhipSetArithmetic(0)  # AVIFHIP_ARITHMETIC_AUTO: follow rgb.avoidLibYUV like a libavif built with libyuv
    integer = args.arithmetic == "integer"

    # ---- synthetic frames, resident in HBM before the timed region ----
    frames = []
    for f in range(FRAMES_IN_FLIGHT):
        img = abi.make_yuv(WIDTH, HEIGHT, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, abi.AVIF_MATRIX_COEFFICIENTS_BT709)
        synth.fill_yuv(img, 0x12345678 + rank * FRAMES_IN_FLIGHT + f)
        rgb = abi.make_rgb(WIDTH, HEIGHT, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR,
                           avoid_libyuv=not integer, allocate=False)
        dimg = device.DeviceYUV(img)
        drgb = device.DeviceRGB(rgb)
        frames.append((dimg, drgb))
        del img
    n_streams = max(1, min(args.streams, FRAMES_IN_FLIGHT))
    streams = [lib.avifhipStreamCreate() for _ in range(n_streams)]
    if any(not s for s in streams):
        raise SystemExit("bench.py: avifhipStreamCreate failed: " + lib.avifhipLastError().decode())

    convert = lib.avifhipImageYUVToRGBAsync
    calls = [(frames[k][0].struct, frames[k][1].struct, streams[k % n_streams]) for k in range(FRAMES_IN_FLIGHT)]

    def run(steps: int) -> None:
        for k in range(steps):
            a, b, s = calls[k % FRAMES_IN_FLIGHT]
            if convert(a, b, s) != 0:
                native.check(1, "avifhipImageYUVToRGBAsync")

    def device_sync() -> None:
        for s in streams:
            native.check(lib.avifhipSynchronize(s), "avifhipSynchronize")

    run(args.warmup)
    device_sync()
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
    t0 = time.perf_counter()
    run(args.steps)
    device_sync()
    if dist is not None:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_name = native.last_kernel()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    # ---- dominant kernel: average launch duration from HIP events on the launch stream ----
    # (a) cycling over the distinct frames: every launch streams from/to HBM; (b) one frame repeated: its 50 MB of
    # input stay in the 256 MB Infinity Cache.  (a) is the roofline figure.
    n, imgs, rgbs = _cycle_args(frames)
    def median(xs):
        xs = sorted(xs)
        return xs[len(xs) // 2]

    # median of 7 event-timed bursts of 40 launches (not the best one: the figure must agree with a profiler's average)
    kernel_ms_stream = median([lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 4, 40, None) for _ in range(7)])
    kernel_ms_same = median([lib.avifhipTimeYUVToRGB(frames[0][0].struct, frames[0][1].struct, 4, 40, None) for _ in range(7)])

    # the other arithmetic family on the same frames (same buffers, only rgb.avoidLibYUV flipped), kernel timing only
    for _, drgb in frames:
        drgb.struct.avoidLibYUV = 1 if integer else 0
    other_ms_stream = median([lib.avifhipTimeYUVToRGBCycle(n, imgs, rgbs, 4, 40, None) for _ in range(7)])
    other_kernel = native.last_kernel()
    for _, drgb in frames:
        drgb.struct.avoidLibYUV = 0 if integer else 1

    mp_per_step = WIDTH * HEIGHT / 1e6
    value = mp_per_step * args.steps * world / elapsed
    alg_bytes = ALGORITHMIC_BYTES_PER_PIXEL * WIDTH * HEIGHT
    achieved = alg_bytes / (kernel_ms_stream * 1e-3) / 1e9

    out = {
        "metric": "megapixels/sec YUV420->RGBA (8K)",
        "value": round(value, 1),
        "unit": "megapixels/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i32" if integer else "f32",  # the arithmetic type the path computes in: libyuv's fixed point / libavif's fp32
        "data": "synthetic",
        "config": {
            "workload": "7680x4320 8-bit YUV420 BT.709 limited -> RGBA8, bilinear chroma upsampling, HBM-resident, "
                        f"{FRAMES_IN_FLIGHT} distinct frames cycled per rank on {n_streams} HIP streams",
            "arithmetic": ("libavif API defaults (avoidLibYUV=0): libyuv fixed point, byte-identical to a libavif built with libyuv" if integer
                           else "avoidLibYUV=1: libavif built-in fp32 path, byte-identical to a libavif built without libyuv"),
            "kernel": kernel_name,
            "frames_per_step": 1,
            "parallelism": f"frames sharded over {world} rank(s), no collective",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": None,
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "kernel_ms_hbm_streaming": round(kernel_ms_stream, 5),
            "kernel_ms_same_frame": round(kernel_ms_same, 5),
            "frac_same_frame": round(alg_bytes / (kernel_ms_same * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "read_only_GBps": round(1.5 * WIDTH * HEIGHT / (kernel_ms_stream * 1e-3) / 1e9, 1),
            ("fp32_path" if integer else "integer_path"): {
                "kernel": other_kernel,
                "kernel_ms_hbm_streaming": round(other_ms_stream, 5),
                "achieved": round(alg_bytes / (other_ms_stream * 1e-3) / 1e9, 1),
                "frac": round(alg_bytes / (other_ms_stream * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            },
        },
    }
    traffic_file = ROOT / "profiles" / "pmc_traffic.json"
    if traffic_file.exists():
        try:
            tj = json.loads(traffic_file.read_text())
            # the counters were collected for one kernel family: use them only for that family
            if ("TileFxKernel" in tj.get("kernel", "")) == ("fixed" in kernel_name):
                out["roofline"]["traffic"] = tj.get("traffic_bytes_per_launch")
        except Exception:
            pass

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(abi, synth, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    for s in streams:
        lib.avifhipStreamDestroy(s)
    if dist is not None:
        dist.destroy_process_group()


def _cycle_args(frames):
    from libavif_amd import abi

    n = len(frames)
    imgs = (C.POINTER(abi.avifImage) * n)(*[C.pointer(f[0].struct) for f in frames])
    rgbs = (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(f[1].struct) for f in frames])
    return n, imgs, rgbs


if __name__ == "__main__":
    main()
