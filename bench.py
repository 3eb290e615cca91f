#!/usr/bin/env python3
"""bench.py -- megapixels/s of libavif's YUV->RGB reformat hot path on MI355X.

A "step" is one pass of the hot path over one synthetic 8K frame: 7680x4320 8-bit YUV 4:2:0, BT.709 limited
range -> RGBA8 with bilinear chroma upsampling (BASELINE.json configs[1]), planes and pixels resident in HBM, converted
with the API defaults (rgb.avoidLibYUV = 0): the reference's INTEGER path, i.e. byte-identical to what a libavif built
with libyuv computes (I420ToARGBMatrixFilter, kFilterBilinear).  The fp32 path (avoidLibYUV = 1, byte-identical to a
libavif built without libyuv) is timed next to it and reported inside "roofline" as "fp32_path".

  python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R]

Timed region: W untimed warm-up steps, then EXACTLY K steps between a barrier + device synchronisation on both sides
(MAX over ranks); that region is repeated R times (default 9) and the MEDIAN is reported as ms_per_step / value -- a single
region of a few dozen 30-microsecond launches is noise.  Steps cycle over 4 distinct frames (730 MB) on one HIP stream
(--streams 2: round-robin on two; frames are independent units of work, so consecutive steps then overlap each other's head
and tail and ms_per_step drops below one kernel's own duration -- reported as "two_streams").
Before anything is timed the GPU is kept busy for --preheat-ms (default 300 ms): an idle MI355X sits at 95 MHz and
needs ~40 ms of work to reach its running clocks (tests/tools/spread_probe.py: the first bursts run 55 -> 30 us per launch).

N > 1: one rank per GPU over RCCL (torch.distributed.run, launched by the driver -- or by this script itself when it is
started plainly with --gpus N > 1); every rank converts its own frames, there is no data-path collective ("scaling":
"weak"); ranks only meet in the barriers around the timed regions.

The line also carries, as first-class fields measured in the same run: "roofline.cold" (the same kernel with 12 frames cycled, 2.2 GB:
nothing stays in the Infinity Cache), "fp32" (the built-in fp32 arithmetic, rgb.avoidLibYUV = 1, on the same frames) and "planes_4k"
(3840x2160 planes, both arithmetics): the north star asks for 4K and 8K planes and the reference compiled from its own sources computes
the fp32 arithmetic.

"configs" carries the other BASELINE.json configurations measured in the same run and the same way (kernels alone, HIP events on the launch
stream, the median of bursts; algorithmic bytes of SURVEY.md 8d): cfg1 (256x256 defaults), cfg3 (8K 10-bit 4:4:4 + alpha -> premultiplied
RGBA16), cfg4 (4K RGBA8 -> 4:2:0 + alpha), cfg5x64 (64 separately stored 1080p 10-bit tiles, one batched launch) and cfg5grid (the same tiles
into one canvas, seams included); "ceilings" the no-arithmetic byte-movement kernel on 4K and 1080p planes (what the chip sustains on jobs that
short); "gainmap" avifRGBImageApplyGainMap on a 4K image.  The timed region itself runs on ONE stream, so that `value`,
`ms_per_step` and `roofline.kernel_ms` describe the same thing; "two_streams" is the same region with consecutive frames on two streams.

  python bench.py --dry-run --gpus N   exercises the rank / aggregation code (process group, barriers, MAX over ranks, the single JSON
  line, cfg5's tile blocks) WITHOUT a GPU over gloo with a converter that only sleeps; the line says "data": "dry-run" and its numbers
  mean nothing (tests/test_bench_ranks.py).

"roofline" describes the dominant kernel alone: algorithmic bytes per launch (5.5 B/pixel: each input sample read once,
each output byte written once) divided by the kernel's average duration, measured with HIP events on the launch stream over
back-to-back single-stream launches that cycle over the same 4 frames.  With 4 frames the 200 MB of input planes can stay in
the 256 MB Infinity Cache while the outputs stream to HBM; "deep_streaming" repeats the timing over 12 frames (2.2 GB), where
nothing can.  "ceiling" is a kernel of the library that moves the same bytes with NO arithmetic (kernels_bench.hip), timed the
same way: what the chip itself sustains for this byte movement.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WIDTH, HEIGHT = 7680, 4320
FRAMES_IN_FLIGHT = 4  # distinct frame buffers cycled by the timed loop (4 x 182 MB > Infinity Cache)
DEEP_FRAMES = 12      # ... by the "deep_streaming" kernel timing (2.2 GB: inputs cannot stay in the Infinity Cache either)
SEQUENCE_FRAMES = 4   # frames per launch of the sequence rows (avifhipImageYUVToRGBBatchAsync over large frames: one launch of the single-image kernels)
DEEP_FRAMES_4K = 24   # 4K frames cycled by the cold rows of planes_4k (1.1 GB)
STREAMS = 1           # the timed region's streams (2: consecutive frames overlap head and tail; measured as well, reported as "two_streams")
ALGORITHMIC_BYTES_PER_PIXEL = 5.5  # 1.5 B read (Y + U/4 + V/4) + 4 B written (RGBA8), SURVEY.md 8d
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=9, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--preheat-ms", type=float, default=300.0, help="GPU work before anything is timed (clock ramp from idle)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=STREAMS)
    ap.add_argument("--no-second-stream-count", action="store_true",
                    help="skip the side measurement of the timed region on the other stream count (profiling runs: overlapping kernels of two streams "
                         "would inflate the profiler's per-kernel average of the one kernel the line is about)")
    ap.add_argument("--workload", choices=("cfg2", "cfg5"), default="cfg2",
                    help="cfg2 (default): 8K frames, one per step, frames sharded over the ranks (weak scaling); cfg5: ONE 15360x8640 canvas of 64 "
                         "10-bit tiles per step, its tiles sharded over the ranks (strong scaling, BASELINE.json configs[4])")
    ap.add_argument("--arithmetic", choices=("integer", "fp32"), default="integer",
                    help="integer: API defaults, libyuv's fixed point (default); fp32: rgb.avoidLibYUV = 1, libavif's built-in path")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the 8K frames of the headline (timed regions, kernel timings of both arithmetics warm / cold / same frame, the live ceiling): no 4K planes, no "
                         "other configurations, no gain map -- profiling runs: since round 5 one kernel instantiation serves 8K, 4K and 1080p frames, and a profiler's "
                         "per-kernel average over the whole bench would mix their durations and byte counts")
    ap.add_argument("--in-process", action="store_true",
                    help="ONE process drives all --gpus N devices through the library's own device farm (avifhipSetDeviceSet: one worker thread, context and host "
                         "link per GPU, no torch, no launcher): host-resident frames / canvas in, host-resident pixels out -- what an unmodified libavif over "
                         "seam A / seam B gets from an N-GPU node (strong scaling of one image).  AVIFHIP_BENCH_DEVICES=0,0 names the set explicitly (one-GPU boxes)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: gloo process group and a converter that sleeps -- exercises the multi-rank plumbing only (numbers are meaningless)")
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
    if kind == "reference":
        out_extra.update(cpu_rows_threaded(abi, synth, lib))
    return {**out_extra, "value": round(mp * frames / t_total, 2), "unit": "megapixels/s", "cores": 1, "kind": kind,
            "sample": f"{frames} x 7680x4320 8-bit 4:2:0 BT.709 limited -> RGBA8 bilinear frames, libavif built-in float path "
                      f"(the reference compiled from its own sources has no libyuv: avoidLibYUV=1 arithmetic; maxThreads=1: the reference "
                      f"runs 4:2:0 bilinear single-threaded), {t_total:.1f} s of CPU; "
                      f"best frame {mp / best:.1f} MP/s",
            "best_value": round(mp / best, 2)}


def cpu_rows_threaded(abi, synth, ref):
    """BASELINE.md section 3's threaded rows, measured in this run on this box's host cores with the reference compiled from its sources:
      threads8_cfg3      cfg3 (8K 10-bit 4:4:4 + alpha -> premultiplied RGBA16) with rgb.maxThreads = 8 -- legal for 4:4:4 (src/reformat.c:1680-1688), which
                         the reference then splits into row bands on 8 threads itself (src/reformat.c:1695-1747); and the same with maxThreads = 1
      all_cores_cfg5     cfg5's 64 tiles (1920x1080 10-bit 4:2:0 -> RGBA(10) bilinear), one single-threaded conversion per tile over a pool of host
                         threads (ctypes releases the GIL): the CPU analogue of the tile farm"""
    from concurrent.futures import ThreadPoolExecutor

    fn = ref.avifImageYUVToRGB
    out = {}
    try:
        img = abi.make_yuv(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, with_alpha=True)
        synth.fill_yuv(img, 0x12345678)
        mp = 7680 * 4320 / 1e6
        for threads, reps in ((8, 3), (1, 1)):
            rgb = abi.make_rgb(7680, 4320, 16, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=True, alpha_premultiplied=True, max_threads=threads)
            best = float("inf")
            for _ in range(reps):
                t0 = time.perf_counter()
                if fn(img.struct, rgb.struct) != 0:
                    raise RuntimeError("conversion failed")
                best = min(best, time.perf_counter() - t0)
            out["threads8_cfg3" if threads == 8 else "threads1_cfg3"] = {
                "value": round(mp / best, 1), "unit": "megapixels/s", "cores": threads, "ms": round(best * 1e3, 1),
                "sample": f"best of {reps} x 7680x4320 10-bit 4:4:4 BT.2020 full + alpha -> RGBA16 premultiplied, reference from source, rgb.maxThreads = {threads}"}
        del img, rgb
        cores = os.cpu_count() or 1
        tiles = []
        for k in range(64):
            timg = abi.make_yuv(1920, 1080, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
            if k < 4:
                synth.fill_yuv(timg, 0x12345678 + k)
            else:
                for p in range(3):
                    timg.planes[p][...] = tiles[k % 4][0].planes[p]
            trgb = abi.make_rgb(1920, 1080, 10, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=True)
            tiles.append((timg, trgb))
        workers = min(64, cores)
        best = float("inf")
        with ThreadPoolExecutor(workers) as pool:
            for _ in range(3):
                t0 = time.perf_counter()
                if any(r != 0 for r in pool.map(lambda t: fn(t[0].struct, t[1].struct), tiles)):
                    raise RuntimeError("conversion failed")
                best = min(best, time.perf_counter() - t0)
        out["all_cores_cfg5"] = {"value": round(64 * 1920 * 1080 / 1e6 / best, 1), "unit": "megapixels/s", "cores": workers, "host_cores": cores, "ms": round(best * 1e3, 1),
                                 "sample": "best of 3 x cfg5's 64 tiles (1920x1080 10-bit 4:2:0 -> RGBA(10) bilinear), reference from source, one single-threaded "
                                           f"conversion per tile on {workers} host threads"}
    except Exception as exc:  # the headline's baseline must not depend on these side rows
        out["threaded_rows_error"] = repr(exc)
    return out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run (one per GPU)
    and hand back their exit code.  Fails loudly when the node has fewer than N GPUs."""
    if "--dry-run" not in sys.argv:
        from libavif_amd import native

        have = native.load().avifhipDeviceCount()
        if have < n:
            raise SystemExit(f"bench.py: --gpus {n} requested but only {have} HIP device(s) are visible -- refusing to report a {n}-GPU number")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.fspath(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


class _stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: communication libraries print banners through C stdio, which is block-buffered
    when stdout is not a terminal -- without the flush on the way out the text would sit in the buffer and come out on the REAL stdout at
    exit, after the JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class DryRunLib:
    """--dry-run: stands in for libavifhip.so so that the rank / aggregation code runs where there is no GPU.  Converts nothing: every
    conversion call sleeps for about a kernel's duration, every timing helper returns a constant."""

    def __init__(self, world):
        self._world = world

    def avifhipDeviceCount(self):
        return self._world

    def avifhipLastError(self):
        return b"dry run"

    def avifhipLastTransferBytes(self, up, down):
        up._obj.value, down._obj.value = 0, 0

    def __getattr__(self, name):
        if name.startswith("avifhipTime"):
            return lambda *a: 0.03
        if name in ("avifhipImageYUVToRGBAsync", "avifhipImageYUVToRGBBatchAsync", "avifhipImageYUVToRGBRects"):
            def convert(*a):
                time.sleep(30e-6)
                return 0
            return convert
        if name == "avifhipStreamCreate":
            return lambda *a: 1
        if name.startswith("avifhip"):
            return lambda *a: 0
        raise AttributeError(name)


class _HostOnly:
    """--dry-run: what device.DeviceYUV / DeviceRGB hand to the timed loop, without device memory."""

    def __init__(self, host):
        self.struct = host.struct


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    args = parse_args()
    if args.in_process:
        print(json.dumps(run_in_process(args)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    dist = None
    torch = None
    # (AVIFHIP_BENCH_FORCE_DIST=1: take the multi-rank code path -- torch + RCCL process group, barriers, max-over-ranks --
    # with a single rank too, to exercise it on a one-GPU box)
    if args.dry_run:
        if world > 1:
            import torch  # noqa: F811
            import torch.distributed as dist  # noqa: F811

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            with _stdout_to_stderr():  # (gloo announces its connections on stdout, like RCCL its version)
                dist.init_process_group(backend="gloo")
                dist.barrier()
    elif world > 1 or os.environ.get("AVIFHIP_BENCH_FORCE_DIST") == "1":
        # torch first: its bundled HIP runtime must be the one libavifhip.so binds to (same SONAME)
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"bench.py: rank {rank} has no GPU (local rank {local_rank}, {torch.cuda.device_count()} device(s) visible)")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on STDOUT when its communicator comes up; the contract is ONE JSON line there, so
        # stdout points at stderr while the process group initialises and runs its first collective
        with _stdout_to_stderr():
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()

    from libavif_amd import abi, device, native, synth

    global WIDTH, HEIGHT
    if args.dry_run:
        lib = DryRunLib(world)
        native.last_kernel = lambda: "dry-run"
        WIDTH, HEIGHT = 256, 128  # nothing is converted: token frames
    else:
        lib = native.load()
    if lib.avifhipDeviceCount() <= 0:
        raise SystemExit("bench.py: no HIP device visible -- there is no CPU fallback for the product path")
    native.check(lib.avifhipSetDevice(local_rank if world > 1 else 0), "avifhipSetDevice")
    lib.avifhipSetArithmetic(0)  # AVIFHIP_ARITHMETIC_AUTO: follow rgb.avoidLibYUV like a libavif built with libyuv
    integer = args.arithmetic == "integer"
    if args.workload == "cfg5":
        out = run_cfg5(args, lib, rank, world, dist, torch)
        if rank == 0:
            print(json.dumps(out))
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- synthetic frames, resident in HBM before the timed region ----
    def make_frames(w, h, count):
        out = []
        for f in range(count):
            img = abi.make_yuv(w, h, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, abi.AVIF_MATRIX_COEFFICIENTS_BT709)
            synth.fill_yuv(img, 0x12345678 + (rank * DEEP_FRAMES + f) % 4)  # 4 distinct contents are plenty: the buffers are what is cycled
            rgb = abi.make_rgb(w, h, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR,
                               avoid_libyuv=not integer, allocate=False)
            out.append((_HostOnly(img), _HostOnly(rgb)) if args.dry_run else (device.DeviceYUV(img), device.DeviceRGB(rgb)))
            del img
        return out

    frames = make_frames(WIDTH, HEIGHT, DEEP_FRAMES)
    # the north star's second plane size (3840x2160): the first FRAMES_IN_FLIGHT are the cycled ones, all of them the cold rows'
    frames_4k_all = [] if args.headline_only else make_frames(WIDTH // 2, HEIGHT // 2, DEEP_FRAMES_4K)
    frames_4k = frames_4k_all[:FRAMES_IN_FLIGHT]
    n_streams = max(1, min(args.streams, FRAMES_IN_FLIGHT))
    streams = [lib.avifhipStreamCreate() for _ in range(max(n_streams, 2))]
    if any(not s for s in streams):
        raise SystemExit("bench.py: avifhipStreamCreate failed: " + lib.avifhipLastError().decode())

    convert = lib.avifhipImageYUVToRGBAsync

    def runner(count):
        calls = [(frames[k][0].struct, frames[k][1].struct, streams[k % count]) for k in range(FRAMES_IN_FLIGHT)]

        def run(steps: int) -> None:
            for k in range(steps):
                a, b, s = calls[k % FRAMES_IN_FLIGHT]
                if convert(a, b, s) != 0:
                    native.check(1, "avifhipImageYUVToRGBAsync")
        return run

    run = runner(n_streams)

    def device_sync() -> None:
        for s in streams:
            native.check(lib.avifhipSynchronize(s), "avifhipSynchronize")

    # ---- preheat: clocks up before anything is timed ----
    t_heat = time.perf_counter()
    while (time.perf_counter() - t_heat) * 1e3 < (0.0 if args.dry_run else args.preheat_ms):
        run(200)
        device_sync()

    # ---- the contract's timed region, `repeats` times ----
    region_s = timed_regions(run, device_sync, args.steps, args.warmup, args.repeats, dist, torch)
    kernel_name = native.last_kernel()
    elapsed = median(region_s)
    # the same region with consecutive frames on the other stream count (1 <-> 2): heads and tails of independent frames overlap on two
    other_streams = 2 if n_streams == 1 else 1
    other_s = None if args.no_second_stream_count else timed_regions(runner(other_streams), device_sync, args.steps, args.warmup, max(3, args.repeats // 3), dist, torch)

    # ---- kernels alone: average launch duration from HIP events on the launch stream, single stream, back to back ----
    n4, imgs4, rgbs4 = _cycle_args(frames[:FRAMES_IN_FLIGHT])
    nd, imgsd, rgbsd = _cycle_args(frames)
    nk, imgsk, rgbsk = _cycle_args(frames_4k) if frames_4k else (0, None, None)
    nkd, imgskd, rgbskd = _cycle_args(frames_4k_all) if frames_4k_all else (0, None, None)

    def burst(fn, *a):
        # 40 ms of the SAME kernel first: after a change of kernel the first ~10 ms of launches run up to 25 % slower (tests/tools/
        # sustain_probe.py, profiles/r03_sustain_probe.txt: the fp32 kernel 39.9, 34.8, 38.3 ... us per launch before it settles at 32-33),
        # then the median of 9 event-timed bursts of 40 launches (not the best one: the figure must agree with a profiler's average)
        def checked(ms):
            if ms < 0:
                raise SystemExit("bench.py: a timing call failed: " + lib.avifhipLastError().decode())
            return ms

        spent = 0.0
        while spent < 40.0:
            spent += max(checked(fn(*a, 0, 100, None)), 1e-3) * 100
        return median([checked(fn(*a, 4, 40, None)) for _ in range(9)])

    def set_arithmetic(use_integer: bool) -> None:
        for _, drgb in frames + frames_4k_all:
            drgb.struct.avoidLibYUV = 0 if use_integer else 1

    timings = {}
    for fam, use_integer in (("integer", True), ("fp32", False)):
        set_arithmetic(use_integer)
        t = {"warm": burst(lib.avifhipTimeYUVToRGBCycle, n4, imgs4, rgbs4)}
        t["kernel"] = native.last_kernel()
        t["cold"] = burst(lib.avifhipTimeYUVToRGBCycle, nd, imgsd, rgbsd)
        t["same"] = burst(lib.avifhipTimeYUVToRGB, frames[0][0].struct, frames[0][1].struct)
        t["4k"] = burst(lib.avifhipTimeYUVToRGBCycle, nk, imgsk, rgbsk) if nk else None
        # sequences: SEQUENCE_FRAMES frames per launch (milliseconds per LAUNCH), inputs cache-resident and not
        t["seq_warm"] = burst(lib.avifhipTimeYUVToRGBBatchCycle, n4, imgs4, rgbs4, SEQUENCE_FRAMES)
        t["seq_kernel"] = native.last_kernel()
        t["seq_cold"] = burst(lib.avifhipTimeYUVToRGBBatchCycle, nd, imgsd, rgbsd, SEQUENCE_FRAMES)
        t["4k_cold"] = burst(lib.avifhipTimeYUVToRGBCycle, nkd, imgskd, rgbskd) if nkd else None
        t["4k_seq_warm"] = burst(lib.avifhipTimeYUVToRGBBatchCycle, nk, imgsk, rgbsk, SEQUENCE_FRAMES) if nk else None
        t["4k_seq_cold"] = burst(lib.avifhipTimeYUVToRGBBatchCycle, nkd, imgskd, rgbskd, SEQUENCE_FRAMES) if nkd else None
        timings[fam] = t
    set_arithmetic(integer)
    main_fam, other_fam = ("integer", "fp32") if integer else ("fp32", "integer")
    kernel_ms_stream, kernel_ms_deep, kernel_ms_same = timings[main_fam]["warm"], timings[main_fam]["cold"], timings[main_fam]["same"]

    # the chip's ceiling for this byte movement: same bytes, same lane mapping, no arithmetic (overwrites the RGB buffers)
    ceil_ms_stream = burst(lib.avifhipTimeStreamCeiling, n4, imgs4, rgbs4)
    ceil_ms_deep = burst(lib.avifhipTimeStreamCeiling, nd, imgsd, rgbsd)
    ceil_ms_seq_cold = burst(lib.avifhipTimeStreamCeilingBatchCycle, nd, imgsd, rgbsd, SEQUENCE_FRAMES)
    ceil_seq_pattern = native.last_kernel()

    more = {} if (args.dry_run or rank != 0 or args.headline_only) else measured_elsewhere(lib, abi, device, native, synth, burst, frames_4k)
    set_arithmetic(integer)
    mp_per_step = WIDTH * HEIGHT / 1e6
    value = mp_per_step * args.steps * world / elapsed
    alg_bytes = ALGORITHMIC_BYTES_PER_PIXEL * WIDTH * HEIGHT

    def gbps(ms, pixels=WIDTH * HEIGHT):
        return ALGORITHMIC_BYTES_PER_PIXEL * pixels / (ms * 1e-3) / 1e9

    def block(ms, pixels=WIDTH * HEIGHT, **extra):
        """{kernel_ms, achieved GB/s, fraction of the HBM peak, megapixels/s of the kernel alone}"""
        return {"kernel_ms": round(ms, 5), "achieved": round(gbps(ms, pixels), 1), "frac": round(gbps(ms, pixels) / HBM_PEAK_GBPS, 4),
                "value": round(pixels / 1e6 / (ms * 1e-3), 1), **extra}

    achieved = gbps(kernel_ms_stream)
    px4k = (WIDTH // 2) * (HEIGHT // 2)

    def seq_block(ms_per_launch, pixels_per_frame, frames_cycled, what, **extra):
        """a sequence row: one launch converts SEQUENCE_FRAMES frames -- algorithmic bytes per launch = 5.5 B x the launch's pixels"""
        if ms_per_launch is None:
            return None
        px = pixels_per_frame * SEQUENCE_FRAMES
        return {"what": what, "frames_per_launch": SEQUENCE_FRAMES, "frames_cycled": frames_cycled, "kernel_ms": round(ms_per_launch, 5),
                "us_per_frame": round(1e3 * ms_per_launch / SEQUENCE_FRAMES, 3), "algorithmic_bytes_per_launch": int(ALGORITHMIC_BYTES_PER_PIXEL * px),
                "achieved": round(gbps(ms_per_launch, px), 1), "frac": round(gbps(ms_per_launch, px) / HBM_PEAK_GBPS, 4),
                "value": round(px / 1e6 / (ms_per_launch * 1e-3), 1), **extra}

    def sequences(fam):
        t = timings[fam]
        return {
            "what": f"avifhipImageYUVToRGBBatchAsync over {SEQUENCE_FRAMES} frames: ONE launch of the single-image kernel, grid z = frame, the frames' addresses in the "
                    "kernel arguments (no descriptor table, no upload, no event between launches)",
            "kernel": t["seq_kernel"],
            "inputs_cache_resident": seq_block(t["seq_warm"], WIDTH * HEIGHT, FRAMES_IN_FLIGHT, f"{FRAMES_IN_FLIGHT} frames cycled (L3-resident planes: not an HBM figure)"),
            "cold": seq_block(t["seq_cold"], WIDTH * HEIGHT, DEEP_FRAMES, f"{DEEP_FRAMES} frames cycled (2.2 GB): every byte comes from and goes to HBM"),
        }
    out = {
        "metric": "megapixels/sec YUV420->RGBA (8K)",
        "value": round(value, 1),
        "unit": "megapixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i16" if integer else "f32",  # the arithmetic the path computes in: libyuv's fixed point in packed int16 lanes / libavif's fp32
        "data": "dry-run (no GPU, nothing converted: numbers are meaningless)" if args.dry_run else "synthetic",
        "value_basis": f"median of {len(region_s)} timed regions of {args.steps} steps each (min {1e3 * min(region_s) / args.steps:.5f}, max "
                       f"{1e3 * max(region_s) / args.steps:.5f} ms/step), host clock around back-to-back launches on {n_streams} HIP stream(s)"
                       + (" -- one stream: ms_per_step is one kernel's duration plus what the launches leave between kernels, and roofline.kernel_ms "
                          "(events on that stream) describes the same kernel" if n_streams == 1 else
                          ", so consecutive frames overlap head and tail and ms_per_step can be below roofline.kernel_ms (one kernel alone, single stream)")
                       + f"; the input planes of the {FRAMES_IN_FLIGHT} cycled frames can stay in the Infinity Cache -- roofline.cold is the figure without that help",
        "two_streams" if n_streams == 1 else "one_stream": None if other_s is None else {
            "what": f"the same timed region with consecutive frames issued round-robin on {other_streams} HIP stream(s)"
                    + (": independent frames overlap each other's head and tail" if other_streams == 2 else ""),
            "ms_per_step": round(1e3 * median(other_s) / args.steps, 5),
            "value": round(mp_per_step * args.steps * world / median(other_s), 1),
            "regions": len(other_s),
        },
        "config": {
            "workload": "7680x4320 8-bit YUV420 BT.709 limited -> RGBA8, bilinear chroma upsampling, HBM-resident, "
                        f"{FRAMES_IN_FLIGHT} distinct frames cycled per rank on {n_streams} HIP streams",
            "arithmetic": ("libavif API defaults (avoidLibYUV=0): libyuv fixed point, byte-identical to a libavif built with libyuv" if integer
                           else "avoidLibYUV=1: libavif built-in fp32 path, byte-identical to a libavif built without libyuv"),
            "kernel": kernel_name,
            "frames_per_step": 1,
            "parallelism": f"frames sharded over {world} rank(s), no collective",
            "preheat_ms": args.preheat_ms,
            "repeats": len(region_s),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            # the same kernel when nothing is cache-resident (12 frames cycled): what a decoder that streams frames sees -- `cold` below has the details
            "frac_cold": round(gbps(kernel_ms_deep) / HBM_PEAK_GBPS, 4),
            "traffic": None,
            "traffic_source": None,
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "kernel_ms": round(kernel_ms_stream, 5),
            "kernel_ms_inputs_cache_resident": round(kernel_ms_stream, 5),  # = kernel_ms: 4 frames cycled, their 200 MB of planes fit the 256 MB Infinity Cache
            "frames_cycled": FRAMES_IN_FLIGHT,
            "kernel_ms_same_frame": round(kernel_ms_same, 5),
            "frac_same_frame": round(gbps(kernel_ms_same) / HBM_PEAK_GBPS, 4),
            "ceiling": {
                "what": "same bytes, same lane-to-byte mapping, no arithmetic (avifhipTimeStreamCeiling), same timing method",
                "kernel_ms": round(ceil_ms_stream, 5),
                "frac_of_peak": round(gbps(ceil_ms_stream) / HBM_PEAK_GBPS, 4),
                "conversion_vs_ceiling": round(ceil_ms_stream / kernel_ms_stream, 4),
            },
            "cold": block(kernel_ms_deep, what=f"{DEEP_FRAMES} frames cycled (2.2 GB): neither planes nor pixels can stay in the 256 MB Infinity Cache",
                          frames_cycled=DEEP_FRAMES, ceiling_kernel_ms=round(ceil_ms_deep, 5),
                          ceiling_frac_of_peak=round(gbps(ceil_ms_deep) / HBM_PEAK_GBPS, 4), conversion_vs_ceiling=round(ceil_ms_deep / kernel_ms_deep, 4)),
            # the HBM regime with the launch's ramp and tail shared by SEQUENCE_FRAMES frames: the figure a decoder of image sequences sees
            "cold_batched": dict(sequences(main_fam)["cold"], kernel=timings[main_fam]["seq_kernel"], traffic=None, traffic_source=None,
                                 ceiling={"what": "same bytes, same frames per launch, no arithmetic (avifhipTimeStreamCeilingBatchCycle: the fastest of the mover's patterns)",
                                          "kernel_ms": round(ceil_ms_seq_cold, 5), "pattern": ceil_seq_pattern,
                                          "frac_of_peak": round(gbps(ceil_ms_seq_cold, WIDTH * HEIGHT * SEQUENCE_FRAMES) / HBM_PEAK_GBPS, 4),
                                          "conversion_vs_ceiling": round(ceil_ms_seq_cold / timings[main_fam]["seq_cold"], 4)}),
        },
        # the other arithmetic and the other plane size, measured in this run with the same method as roofline.kernel_ms
        "fp32": block(timings["fp32"]["warm"], kernel=timings["fp32"]["kernel"], cold=block(timings["fp32"]["cold"]), sequence=sequences("fp32"),
                      what="rgb.avoidLibYUV = 1: libavif's built-in fp32 arithmetic (what the reference compiled from its own sources computes), same 8K frames"),
        "integer": block(timings["integer"]["warm"], kernel=timings["integer"]["kernel"], cold=block(timings["integer"]["cold"]), sequence=sequences("integer"),
                         what="API defaults: libyuv's fixed point (what a stock libavif computes), same 8K frames"),
        "planes_4k": None if args.headline_only else {
            "what": f"3840x2160 planes, same configuration, kernel alone ({int(ALGORITHMIC_BYTES_PER_PIXEL * px4k)} B per frame): {FRAMES_IN_FLIGHT} frames cycled "
                    f"(183 MB: L3-resident, not an HBM figure), `cold` {DEEP_FRAMES_4K} frames cycled (1.1 GB), `sequence` {SEQUENCE_FRAMES} frames per launch",
            **{fam: block(timings[fam]["4k"], px4k, cold=block(timings[fam]["4k_cold"], px4k),
                          sequence={"inputs_cache_resident": seq_block(timings[fam]["4k_seq_warm"], px4k, FRAMES_IN_FLIGHT, f"{FRAMES_IN_FLIGHT} frames cycled (L3-resident)"),
                                    "cold": seq_block(timings[fam]["4k_seq_cold"], px4k, DEEP_FRAMES_4K, f"{DEEP_FRAMES_4K} frames cycled (1.1 GB)")})
               for fam in ("integer", "fp32")},
        },
    }
    out.update(more)
    # (names of round 2's line, kept for the profile tooling)
    out["roofline"]["deep_streaming"] = out["roofline"]["cold"]
    out["roofline"]["fp32_path" if integer else "integer_path"] = {k: out[other_fam][k] for k in ("kernel", "kernel_ms", "achieved", "frac")}
    traffic_file = ROOT / "profiles" / "pmc_traffic.json"
    if traffic_file.exists() and not args.dry_run:
        try:
            tj = json.loads(traffic_file.read_text())
            # the counters were collected for one kernel in a separate rocprofv3 --pmc run: use them only when that kernel is the one reported
            if tj.get("kernel_family", "") == kernel_name:
                out["roofline"]["traffic"] = tj.get("traffic_bytes_per_launch")
                out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command; not measured in this run)"
        except Exception:
            pass

    seq_file = ROOT / "profiles" / "pmc_traffic_sequence.json"
    if seq_file.exists() and not args.dry_run:
        try:
            sj = json.loads(seq_file.read_text()).get("cfg2seq" if integer else "cfg2seq_fp32", {})
            cb = out["roofline"]["cold_batched"]
            if sj.get("kernel_family", "") == cb["kernel"] and sj.get("frames_per_launch") == SEQUENCE_FRAMES and sj.get("traffic_bytes_per_launch"):
                cb["traffic"] = sj["traffic_bytes_per_launch"]
                cb["traffic_source"] = ("profiles/pmc_traffic_sequence.json (tests/tools/seq_evidence.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                        "`stream_sweep.py run cfg2seq`, the same launches and nothing else; not measured in this run)")
        except Exception:
            pass

    # ---- N > 1: the other two multi-GPU shapes in the same line (one driver command captures all three; N = 1 prints today's line) ----
    #   strong_scaling_cfg5      ONE 15360x8640 canvas of 64 tiles per step, its tiles sharded over the ranks (what `--workload cfg5` prints)
    #   in_process_host_to_host  ONE process (rank 0) driving all N devices through the library's device farm, host image in, host pixels out
    #                            (what `--in-process` prints: an unmodified libavif over seam A / seam B with AVIFHIP_DEVICES=all); the other ranks wait
    if (world > 1 or os.environ.get("AVIFHIP_BENCH_ALL_BLOCKS") == "1") and not args.headline_only and os.environ.get("AVIFHIP_BENCH_SIDE_BLOCKS", "1") != "0":
        import copy
        import threading

        # The headline above is measured; the side blocks below have never met more than one physical GPU (no node was to be had).
        # A rank that hangs in one of them must not cost the line: past AVIFHIP_BENCH_SIDE_TIMEOUT seconds rank 0 prints the line without
        # them and every rank leaves.  (AVIFHIP_BENCH_SIDE_BLOCKS=0 skips them.)
        side_done = threading.Event()
        side_timeout = float(os.environ.get("AVIFHIP_BENCH_SIDE_TIMEOUT", "240"))
        fallback_line = json.dumps(dict(out, strong_scaling_cfg5={"error": f"side blocks did not finish within {side_timeout:.0f} s"}, in_process_host_to_host=None, cpu_baseline=None))

        def side_watchdog():
            if not side_done.wait(side_timeout):
                if rank == 0:
                    print(fallback_line, flush=True)
                os._exit(0)

        threading.Thread(target=side_watchdog, daemon=True).start()

        side = copy.copy(args)
        side.steps, side.warmup, side.repeats, side.preheat_ms = min(args.steps, 50), min(args.warmup, 10), min(args.repeats, 3), min(args.preheat_ms, 100.0)
        try:
            cfg5 = run_cfg5(side, lib, rank, world, dist, torch)
            out["strong_scaling_cfg5"] = {k: cfg5[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "config", "roofline", "host_to_host")}
        except Exception as exc:  # (the headline must not depend on a side block; a failing rank would hang the others in a collective: re-raise there)
            if dist is not None:
                raise
            out["strong_scaling_cfg5"] = {"error": repr(exc)}
        if dist is not None:
            dist.barrier()
        if rank == 0 and not args.dry_run:
            try:
                side.gpus = world
                block = run_in_process(side)
                out["in_process_host_to_host"] = {k: block[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "data", "config", "host_link")}
                native.check(lib.avifhipSetDevice(local_rank if world > 1 else 0), "avifhipSetDevice")
            except (Exception, SystemExit) as exc:
                out["in_process_host_to_host"] = {"error": repr(exc)}
        else:
            out["in_process_host_to_host"] = None
        if dist is not None:
            dist.barrier()
        side_done.set()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            out["cpu_baseline"] = cpu_baseline(abi, synth, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    for s in streams:
        lib.avifhipStreamDestroy(s)
    if dist is not None:
        dist.destroy_process_group()


def measured_elsewhere(lib, abi, device, native, synth, burst, frames_4k):
    """The other BASELINE.json configurations, the byte-movement ceilings of the small plane sizes and the gain-map application, each measured
    like roofline.kernel_ms: the kernel(s) alone, HIP events on the launch stream, 40 ms of the same launches first, then the median of 9 bursts.
    Algorithmic bytes: SURVEY.md 8d (every input sample read once, every output byte written once)."""
    BIL = abi.AVIF_CHROMA_UPSAMPLING_BILINEAR

    def row(ms, alg_bytes, pixels, what, **extra):
        gb = alg_bytes / (ms * 1e-3) / 1e9
        return {"what": what, "kernel_ms": round(ms, 5), "algorithmic_bytes_per_launch": int(alg_bytes), "achieved": round(gb, 1),
                "frac": round(gb / HBM_PEAK_GBPS, 4), "value": round(pixels / 1e6 / (ms * 1e-3), 1), "unit": "megapixels/s", **extra}

    def ceiling(ceil_ms, conv_ms, alg_bytes):
        """The byte-movement ceiling of a configuration's own shape (avifhipTimeStreamCeiling*: same planes, same pixels, same buffers cycled the same
        way, no arithmetic; the fastest of the mover's tile shapes / orders), and where the conversion stands against it."""
        gb = alg_bytes / (ceil_ms * 1e-3) / 1e9
        return {"what": "same bytes, same buffers cycled the same way, no arithmetic (kernels_bench.hip streamMoveKernel: the fastest of its tile shapes / orders)",
                "kernel_ms": round(ceil_ms, 5), "frac_of_peak": round(gb / HBM_PEAK_GBPS, 4), "conversion_vs_ceiling": round(ceil_ms / conv_ms, 4),
                "pattern": native.last_kernel()}

    def y2r(w, h, depth, fmt, rng, mc, rgb_depth, alpha=False, premult=False, avoid=False, seed=0x12345678):
        img = abi.make_yuv(w, h, depth, fmt, rng, mc, with_alpha=alpha)
        synth.fill_yuv(img, seed)
        rgb = abi.make_rgb(w, h, rgb_depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, alpha_premultiplied=premult, avoid_libyuv=avoid, allocate=False)
        return device.DeviceYUV(img), device.DeviceRGB(rgb)

    configs = {}
    # cfg1: BASELINE.json configs[0], the reference's own CPU-runnable case -- 256x256 8-bit 4:2:0 BT.601 full range -> RGBA8, API defaults
    pair = y2r(256, 256, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_FULL, 6, 8)
    ms = burst(lib.avifhipTimeYUVToRGB, pair[0].struct, pair[1].struct)
    configs["cfg1"] = row(ms, 5.5 * 256 * 256, 256 * 256, "256x256 8-bit 4:2:0 BT.601 full -> RGBA8, API defaults; one launch (0.36 MB: launch-bound)",
                          kernel=native.last_kernel())
    # cfg3: 8K 10-bit 4:4:4 BT.2020 full + alpha plane -> RGBA16 premultiplied (530 MB per frame: two frames cycled, nothing is cache-resident)
    pairs = [y2r(7680, 4320, 10, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 9, 16, alpha=True, premult=True, seed=0x12345678 + k) for k in range(2)]
    imgs = (C.POINTER(abi.avifImage) * 2)(*[C.pointer(q[0].struct) for q in pairs])
    rgbs = (C.POINTER(abi.avifRGBImage) * 2)(*[C.pointer(q[1].struct) for q in pairs])
    ms = burst(lib.avifhipTimeYUVToRGBCycle, 2, imgs, rgbs)
    configs["cfg3"] = row(ms, 16.0 * 7680 * 4320, 7680 * 4320, "7680x4320 10-bit 4:4:4 BT.2020 full + alpha -> RGBA16, alpha premultiplied in the same kernel; 2 frames cycled",
                          kernel=native.last_kernel())
    configs["cfg3"]["ceiling"] = ceiling(burst(lib.avifhipTimeStreamCeiling, 2, imgs, rgbs), ms, 16.0 * 7680 * 4320)
    del pairs, imgs, rgbs
    # cfg4: 4K RGBA8 -> 8-bit 4:2:0 BT.709 limited + alpha plane (6.5 B/pixel), the encode direction; same frame and 8 frames cycled (431 MB)
    enc = []
    for k in range(8):
        rgb = abi.make_rgb(3840, 2160, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
        synth.fill_rgb(rgb, 0x12345678 + k % 2, opaque=True)
        img = abi.make_yuv(3840, 2160, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, with_alpha=True)
        enc.append((device.DeviceYUV(img, upload=False), device.DeviceRGB(rgb, upload=True)))
    imgs = (C.POINTER(abi.avifImage) * 8)(*[C.pointer(q[0].struct) for q in enc])
    rgbs = (C.POINTER(abi.avifRGBImage) * 8)(*[C.pointer(q[1].struct) for q in enc])
    ms_same = burst(lib.avifhipTimeRGBToYUV, enc[0][0].struct, enc[0][1].struct)
    ms = burst(lib.avifhipTimeRGBToYUVCycle, 8, imgs, rgbs)
    px4k = 3840 * 2160
    configs["cfg4"] = row(ms, 6.5 * px4k, px4k, "3840x2160 RGBA8 -> 8-bit 4:2:0 BT.709 limited + alpha plane (avifImageRGBToYUV); 8 frames cycled",
                          kernel=native.last_kernel(), same_frame=row(ms_same, 6.5 * px4k, px4k, "the same frame every launch (54 MB: cache-resident)"))
    configs["cfg4"]["ceiling"] = ceiling(burst(lib.avifhipTimeStreamCeilingRGBToYUV, 8, imgs, rgbs), ms, 6.5 * px4k)
    # ... and as an image sequence on its way into an encoder: avifhipImageRGBToYUVBatchAsync, SEQUENCE_FRAMES frames per launch of the same kernel
    ms_seq = burst(lib.avifhipTimeRGBToYUVBatchCycle, 8, imgs, rgbs, SEQUENCE_FRAMES)
    configs["cfg4"]["sequence"] = row(ms_seq, 6.5 * px4k * SEQUENCE_FRAMES, px4k * SEQUENCE_FRAMES,
                                      f"avifhipImageRGBToYUVBatchAsync: {SEQUENCE_FRAMES} frames per launch (grid z = frame, addresses in the kernel arguments), 8 frames cycled",
                                      kernel=native.last_kernel(), frames_per_launch=SEQUENCE_FRAMES, us_per_frame=round(1e3 * ms_seq / SEQUENCE_FRAMES, 3))
    configs["cfg4"]["sequence"]["ceiling"] = ceiling(burst(lib.avifhipTimeStreamCeilingRGBToYUVBatchCycle, 8, imgs, rgbs, SEQUENCE_FRAMES), ms_seq, 6.5 * px4k * SEQUENCE_FRAMES)
    configs["cfg4"]["same_frame"]["ceiling"] = ceiling(burst(lib.avifhipTimeStreamCeilingRGBToYUV, 1, imgs, rgbs), ms_same, 6.5 * px4k)
    del enc, imgs, rgbs
    # cfg5: 64 separately stored 1920x1080 10-bit 4:2:0 tiles -> RGBA (10 bits in 16-bit containers, API defaults), 11 B/pixel
    tiles = []
    for t in range(64):
        img = abi.make_yuv(1920, 1080, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1)
        synth.fill_yuv(img, 0x12345678 + t)
        tiles.append(device.DeviceYUV(img))
    timgs = (C.POINTER(abi.avifImage) * 64)(*[C.pointer(t.struct) for t in tiles])

    def tile_outputs():
        outs = [device.DeviceRGB(abi.make_rgb(1920, 1080, 10, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False, allocate=False)) for _ in range(64)]
        return outs, (C.POINTER(abi.avifRGBImage) * 64)(*[C.pointer(o.struct) for o in outs])

    outs_a, rgbs_a = tile_outputs()
    px_tiles = 64 * 1920 * 1080
    ms = burst(lib.avifhipTimeYUVToRGBBatch, 64, timgs, rgbs_a, None)
    kernel = native.last_kernel()
    ceil_x64 = ceiling(burst(lib.avifhipTimeStreamCeilingBatch, 64, timgs, rgbs_a), ms, 11.0 * 64 * 1920 * 1080)
    # a decoder that rotates its output buffers sends a fresh descriptor table with every batch; one that reuses them launches on the table
    # the device still holds (avifhipTableUploadCount): both regimes
    outs_b, rgbs_b = tile_outputs()
    n_alt, alt = 100, []
    for rep in range(4):  # (the first pass is not timed: it is the new buffers' first touch)
        uploads0 = lib.avifhipTableUploadCount()
        t0 = time.perf_counter()
        for k in range(n_alt):
            native.check(lib.avifhipImageYUVToRGBBatchAsync(64, timgs, rgbs_b if k & 1 else rgbs_a, None, None), "avifhipImageYUVToRGBBatchAsync")
        native.check(lib.avifhipSynchronize(None), "avifhipSynchronize")
        if rep:
            alt.append((time.perf_counter() - t0) / n_alt * 1e3)
    ms_alt = median(alt)
    configs["cfg5x64"] = row(ms, 11.0 * px_tiles, px_tiles, "64 separately stored 1920x1080 10-bit 4:2:0 tiles -> 64 RGBA (10 bits in 16-bit containers) images, ONE batched "
                             "launch per step (avifhipImageYUVToRGBBatchAsync); the same buffers every step: the descriptor table stays on the device",
                             kernel=kernel, ceiling=ceil_x64, rotating_outputs={"what": "two sets of output buffers alternated: every batch uploads its descriptor table; host clock "
                                                              f"around {n_alt} back-to-back calls, median of {len(alt)} passes", "ms_per_batch": round(ms_alt, 5),
                                                              "table_uploads_per_batch": round((lib.avifhipTableUploadCount() - uploads0) / n_alt, 2)})
    del outs_a, outs_b, rgbs_a, rgbs_b
    # ... and into ONE 15360x8640 canvas with the chroma filter reaching across the seams (avifhipGridYUVToRGBAsync), what avifdec's grid path does
    canvas = device.DeviceRGB(abi.make_rgb(15360, 8640, 10, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False, allocate=False))
    grid = native.avifhipGrid(8, 8, 15360, 8640)
    ms = burst(lib.avifhipTimeGridYUVToRGB, C.byref(grid), timgs, None, 0, canvas.struct)
    configs["cfg5grid"] = row(ms, 11.0 * px_tiles, px_tiles, "the same 64 tiles -> one 15360x8640 RGBA canvas, converted where they lie, seams as on the stitched canvas "
                              "(avifhipGridYUVToRGBAsync: every kernel of the call)", kernel=native.last_kernel())

    def canvas_views(cv, depth, pixel_bytes):
        """avifRGBImage views of the 64 tile rectangles of a canvas: the destinations of the byte-movement ceiling's 64 jobs"""
        views = []
        for t in range(64):
            v = abi.make_rgb(1920, 1080, depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False, allocate=False)
            v.struct.pixels = cv.buffer.ptr + (t // 8) * 1080 * cv.struct.rowBytes + (t % 8) * 1920 * pixel_bytes
            v.struct.rowBytes = cv.struct.rowBytes
            views.append(v)
        return views, (C.POINTER(abi.avifRGBImage) * 64)(*[C.pointer(v.struct) for v in views])

    views10, vrgbs10 = canvas_views(canvas, 10, 8)
    configs["cfg5grid"]["ceiling"] = ceiling(burst(lib.avifhipTimeStreamCeilingBatch, 64, timgs, vrgbs10), ms, 11.0 * px_tiles)
    # ... -> RGBA8, where the default arithmetic runs the packed 16-bit kernels: their seam-aware build reads the chroma across the seams itself,
    # ONE launch per canvas (7 B/pixel); AVIFHIP_GRID_SEAM_PASS=1 brings the tile batch + seam pass of rounds 1-3 back for comparison

    def with_seam_pass(fn):
        os.environ["AVIFHIP_GRID_SEAM_PASS"] = "1"
        try:
            return fn()
        finally:
            del os.environ["AVIFHIP_GRID_SEAM_PASS"]

    canvas8 = device.DeviceRGB(abi.make_rgb(15360, 8640, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False, allocate=False))
    ms8 = burst(lib.avifhipTimeGridYUVToRGB, C.byref(grid), timgs, None, 0, canvas8.struct)
    kernel8 = native.last_kernel()
    ms8_pass = with_seam_pass(lambda: burst(lib.avifhipTimeGridYUVToRGB, C.byref(grid), timgs, None, 0, canvas8.struct))
    configs["cfg5grid"]["rgba8"] = row(ms8, 7.0 * px_tiles, px_tiles, "the same tiles -> one RGBA8 canvas: tiles and seams in ONE launch", kernel=kernel8,
                                       with_seam_pass=row(ms8_pass, 7.0 * px_tiles, px_tiles, "tile batch, then the seam kernel (two launches)"))
    views8, vrgbs8 = canvas_views(canvas8, 8, 4)
    configs["cfg5grid"]["rgba8"]["ceiling"] = ceiling(burst(lib.avifhipTimeStreamCeilingBatch, 64, timgs, vrgbs8), ms8, 7.0 * px_tiles)
    del views10, vrgbs10, views8, vrgbs8
    del canvas, canvas8, tiles, timgs
    # a phone photograph: 4032x3024 8-bit 4:2:0 stored as 8 x 6 tiles of 512x512 (the last row cropped) -> RGBA8, the same buffers call after call
    ptiles = []
    for t in range(48):
        img = abi.make_yuv(512, 512, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_FULL, 6)
        synth.fill_yuv(img, 0x2468 + t)
        ptiles.append(device.DeviceYUV(img))
    pimgs = (C.POINTER(abi.avifImage) * 48)(*[C.pointer(t.struct) for t in ptiles])
    pcanvas = device.DeviceRGB(abi.make_rgb(4032, 3024, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=BIL, avoid_libyuv=False, allocate=False))
    pgrid = native.avifhipGrid(6, 8, 4032, 3024)
    px_photo = 4032 * 3024
    msp = burst(lib.avifhipTimeGridYUVToRGB, C.byref(pgrid), pimgs, None, 0, pcanvas.struct)
    kernelp = native.last_kernel()
    msp_pass = with_seam_pass(lambda: burst(lib.avifhipTimeGridYUVToRGB, C.byref(pgrid), pimgs, None, 0, pcanvas.struct))
    configs["photo_grid"] = row(msp, 5.5 * px_photo, px_photo, "4032x3024 8-bit 4:2:0 as 8 x 6 tiles of 512x512 -> one RGBA8 canvas (avifhipGridYUVToRGBAsync), ONE launch",
                                kernel=kernelp, with_seam_pass=row(msp_pass, 5.5 * px_photo, px_photo, "tile batch, then the seam kernel (two launches)"))
    del ptiles, pimgs, pcanvas

    # the chip's ceiling for short jobs: the no-arithmetic byte-movement kernel on 4K and 1080p 8-bit 4:2:0 -> RGBA8 frames (4 frames cycled)
    nk, imgsk, rgbsk = _cycle_args(frames_4k)
    ceil4k = burst(lib.avifhipTimeStreamCeiling, nk, imgsk, rgbsk)
    small = [y2r(1920, 1080, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, 1, 8, seed=0x12345678 + k) for k in range(4)]
    n1, imgs1, rgbs1 = _cycle_args(small)
    ceil1080 = burst(lib.avifhipTimeStreamCeiling, n1, imgs1, rgbs1)
    conv1080 = burst(lib.avifhipTimeYUVToRGBCycle, n1, imgs1, rgbs1)
    px1080 = 1920 * 1080
    ceilings = {
        "what": "same bytes, same lane-to-byte mapping, no arithmetic (avifhipTimeStreamCeiling), 8-bit 4:2:0 -> RGBA8, 4 frames cycled: what the chip sustains on a job this short",
        "planes_4k": row(ceil4k, 5.5 * px4k, px4k, "3840x2160 (45.6 MB per launch)"),
        "planes_1080p": row(ceil1080, 5.5 * px1080, px1080, "1920x1080 (11.4 MB per launch)",
                            conversion=row(conv1080, 5.5 * px1080, px1080, "the conversion kernel on the same frames (API defaults)", kernel=native.last_kernel())),
    }
    del small

    # avifRGBImageApplyGainMap: 3840x2160 RGBA8 sRGB / BT.709 base -> RGBA10 PQ / BT.2020, 8-bit 4:4:4 gain map of the same size
    base = abi.make_rgb(3840, 2160, 8, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False)
    synth.fill_rgb(base, 0x4242)
    gimg = abi.make_yuv(3840, 2160, 8, abi.AVIF_PIXEL_FORMAT_YUV444, abi.AVIF_RANGE_FULL, 6)
    synth.fill_yuv(gimg, 0x99)
    gm = abi.avifGainMap()
    for i in range(3):
        gm.gainMapMin[i].n, gm.gainMapMin[i].d = 0, 1
        gm.gainMapMax[i].n, gm.gainMapMax[i].d = 3, 1
        gm.gainMapGamma[i].n, gm.gainMapGamma[i].d = 1, 1
        gm.baseOffset[i].n, gm.baseOffset[i].d = 1, 64
        gm.alternateOffset[i].n, gm.alternateOffset[i].d = 1, 64
    gm.baseHdrHeadroom.n, gm.baseHdrHeadroom.d, gm.alternateHdrHeadroom.n, gm.alternateHdrHeadroom.d = 0, 1, 3, 1
    gm.useBaseColorSpace = 1
    dbase, dgimg = device.DeviceRGB(base, upload=True), device.DeviceYUV(gimg)
    gm.image = C.pointer(dgimg.struct)
    dout = device.DeviceRGB(abi.make_rgb(3840, 2160, 10, abi.AVIF_RGB_FORMAT_RGBA, avoid_libyuv=False, allocate=False))
    clli, diag = abi.avifContentLightLevelInformationBox(), abi.avifDiagnostics()
    ms_kernel = burst(lambda w, n, st: lib.avifhipTimeRGBImageApplyGainMap(dbase.struct, 1, 13, C.byref(gm), 3.0, 9, 16, dout.struct, w, max(n, 2), st))
    kernel = native.last_kernel()
    calls = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(20):
            native.check(lib.avifhipRGBImageApplyGainMapAsync(dbase.struct, 1, 13, C.byref(gm), 3.0, 9, 16, dout.struct, C.byref(clli), C.byref(diag), None),
                         "avifhipRGBImageApplyGainMapAsync")
        native.check(lib.avifhipSynchronize(None), "avifhipSynchronize")  # (round 6: the statistics travel behind each kernel; this call turns them into clli)
        calls.append((time.perf_counter() - t0) / 20 * 1e3)
    # ... and without light levels (clli = NULL): nothing of the answer depends on the pixels then (the fast kernel's precondition rules NaNs out), so
    # the asynchronous entry point returns with its work enqueued -- calls follow each other at the device's pace, one synchronisation at the end
    calls_async = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(20):
            native.check(lib.avifhipRGBImageApplyGainMapAsync(dbase.struct, 1, 13, C.byref(gm), 3.0, 9, 16, dout.struct, None, C.byref(diag), None),
                         "avifhipRGBImageApplyGainMapAsync")
        native.check(lib.avifhipSynchronize(None), "avifhipSynchronize")
        calls_async.append((time.perf_counter() - t0) / 20 * 1e3)
    gain_bytes = (4 + 3 + 8) * px4k  # base pixels + gain-map planes + tone-mapped pixels
    gainmap = row(ms_kernel, gain_bytes, px4k, "avifRGBImageApplyGainMap, 3840x2160 RGBA8 sRGB/BT.709 -> RGBA10 PQ/BT.2020, 8-bit 4:4:4 gain map: the apply kernel alone, which "
                  "converts the gain map's planes itself since round 5 (base pixels 4 + gain-map planes 3 + tone-mapped pixels 8 B/pixel)", kernel=kernel,
                  whole_call={"what": "the whole call WITH light levels (apply with the gain map's YUV -> RGB inside; since round 6 the statistics travel into pinned memory behind the kernel and "
                                      "avifhipSynchronize turns them into clli: the call returns with its work enqueued), host clock around 20 back-to-back calls and one "
                                      "synchronisation, median of 7; algorithmic bytes base 4 + gain-map planes 3 + output 8 B/pixel",
                              "ms_per_call": round(median(calls), 5), "algorithmic_bytes_per_call": int(gain_bytes),
                              "frac": round(gain_bytes / (median(calls) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "maxCLL": int(clli.maxCLL), "maxPALL": int(clli.maxPALL)},
                  whole_call_without_light_levels={"what": "the same call with clli = NULL: no statistics to wait for, the call returns with its kernel enqueued; "
                                                           "host clock around 20 back-to-back calls and one synchronisation, median of 7",
                                                   "ms_per_call": round(median(calls_async), 5),
                                                   "frac": round(gain_bytes / (median(calls_async) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
    # avifRGBImageComputeGainMap (the encode side), device-resident (round 6: avifhipRGBImageComputeGainMapAsync): 3840x2160 RGBA8 sRGB / BT.709 base + RGBA10 PQ /
    # BT.2020 alternate -> 8-bit 4:4:4 gain map + metadata.  The call waits for its own first passes (the metadata is a function of every pixel): host clock
    # around back-to-back calls, median of 5 bursts of 10
    try:
        sys.path.insert(0, os.fspath(ROOT / "tests"))
        import gainmap_cases as G

        cc = G.ComputeCase(3840, 2160, alt_primaries=9, seed=3)
        cbase, calt = G.make_compute_inputs(cc)
        dcb, dca = device.DeviceRGB(cbase, upload=True), device.DeviceRGB(calt, upload=True)
        dgm = device.DeviceYUV(abi.make_yuv(cc.w, cc.h, cc.gm_depth, cc.gm_format, cc.gm_range, cc.gm_matrix), upload=False)
        cgm = abi.avifGainMap()
        cgm.image = C.pointer(dgm.struct)
        t = lib.avifhipTimeRGBImageComputeGainMap
        if t(dcb.struct, 1, 13, dca.struct, 9, 16, C.byref(cgm), 3, 10, None) <= 0:
            raise RuntimeError(lib.avifhipLastError().decode())
        ms_compute = median([t(dcb.struct, 1, 13, dca.struct, 9, 16, C.byref(cgm), 1, 10, None) for _ in range(5)])
        compute_bytes = (4 + 8 + 3) * px4k
        gainmap["compute"] = {
            "what": "avifhipRGBImageComputeGainMapAsync, 3840x2160 RGBA8 sRGB/BT.709 + RGBA10 PQ/BT.2020 -> 8-bit 4:4:4 gain map + metadata, everything device-resident: "
                    "channel minima, ratios, outlier histogram, codes, RGBA -> YUV (five kernels, three waits for the host: offsets, histogram ranges, code steps), host clock "
                    "around 10 back-to-back calls, median of 5; the kernels' own durations: profiles/r06_gainmap_compute.txt",
            "ms_per_call": round(ms_compute, 5), "algorithmic_bytes_per_call": int(compute_bytes),
            "achieved": round(compute_bytes / (ms_compute * 1e-3) / 1e9, 1), "frac": round(compute_bytes / (ms_compute * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "value": round(px4k / 1e6 / (ms_compute * 1e-3), 1), "unit": "megapixels/s"}
    except Exception as exc:  # (the headline must not depend on this side row)
        gainmap["compute"] = {"error": repr(exc)}
    return {"configs": configs, "ceilings": ceilings, "gainmap": gainmap}


def timed_regions(run_steps, sync, steps, warmup, repeats, dist, torch):
    """The contract's timed region, `repeats` times: warm-up steps, barrier + sync, EXACTLY `steps` steps, sync, MAX over ranks."""
    on_gpu = dist is not None and dist.get_backend() != "gloo"  # (--dry-run meets over gloo, without a device)
    region_s = []
    for _ in range(max(1, repeats)):
        run_steps(warmup)
        sync()
        if dist is not None:
            if on_gpu:
                torch.cuda.synchronize()
            dist.barrier()
        t0 = time.perf_counter()
        run_steps(steps)
        sync()
        if on_gpu:
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            dist.barrier()
        region_s.append(elapsed)
    return region_s


def run_cfg5(args, lib, rank, world, dist, torch):
    """BASELINE.json configs[4]: an 8 x 8 AVIF grid of 1920x1080 10-bit 4:2:0 tiles (one 15360x8640 canvas), YUV -> RGB with the API
    defaults (RGBA at the image's depth: 10 bits in 16-bit containers, which libyuv declines -- the fp32 path), bilinear.  The tiles
    are sharded over the ranks (contiguous blocks of the row-major tile list: whole tile rows; no collective): STRONG scaling of one canvas.  A step = one canvas.
      value         : tiles and canvas resident in each rank's HBM, the rank's tiles converted by one batched launch
      host_to_host  : the same canvas from and to HOST memory through avifhipImageYUVToRGBRects (each rank uploads only what
                      its tiles need, downloads only its rectangles): the number an application sees, PCIe-bound"""
    from libavif_amd import abi, device, farm, native, synth

    W, H, TW, TH = 15360, 8640, 1920, 1080
    rects = farm.grid_rects(W, H, TW, TH)
    mine = farm.shard(len(rects), rank, world)
    dry = args.dry_run
    # (--dry-run: nothing is converted -- a token canvas, the tile list and the sharding are the real ones)
    canvas = abi.make_yuv(W, H, 10, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, abi.AVIF_MATRIX_COEFFICIENTS_BT709, allocate=not dry)
    if not dry:
        synth.fill_yuv(canvas, 0x12345678)  # the same decoded canvas on every rank
    rgb_host = abi.make_rgb(W, H, 10, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False, allocate=not dry)
    dimg = _HostOnly(canvas) if dry else device.DeviceYUV(canvas)
    drgb = _HostOnly(rgb_host) if dry else device.DeviceRGB(rgb_host)
    n = len(mine)
    imgs = (C.POINTER(abi.avifImage) * max(n, 1))(*[C.pointer(dimg.struct)] * n)
    rgbs = (C.POINTER(abi.avifRGBImage) * max(n, 1))(*[C.pointer(drgb.struct)] * n)
    crops = (abi.avifCropRect * max(n, 1))(*[abi.avifCropRect(*rects[t]) for t in mine])
    host_crops = (abi.avifCropRect * max(n, 1))(*[abi.avifCropRect(*rects[t]) for t in mine])

    def run_device(steps):
        for _ in range(steps):
            if n:
                native.check(lib.avifhipImageYUVToRGBBatchAsync(n, imgs, rgbs, crops, None), "avifhipImageYUVToRGBBatchAsync")

    def sync():
        native.check(lib.avifhipSynchronize(None), "avifhipSynchronize")

    def run_host(steps):
        for _ in range(steps):
            if n:
                native.check(lib.avifhipImageYUVToRGBRects(canvas.struct, rgb_host.struct, host_crops, n), "avifhipImageYUVToRGBRects")

    t_heat = time.perf_counter()
    while (time.perf_counter() - t_heat) * 1e3 < (0.0 if dry else args.preheat_ms):
        run_device(20)
        sync()
    steps = min(args.steps, 200)
    warmup = min(args.warmup, 20)
    region_s = timed_regions(run_device, sync, steps, warmup, args.repeats, dist, torch)
    kernel_name = native.last_kernel()
    elapsed = median(region_s)
    host_steps = 3
    host_s = timed_regions(run_host, lambda: None, host_steps, 1, max(3, args.repeats // 3), dist, torch)
    host_elapsed = median(host_s)
    up, down = C.c_uint64(0), C.c_uint64(0)
    lib.avifhipLastTransferBytes(C.byref(up), C.byref(down))
    mp = W * H / 1e6
    alg_bytes_rank = 11.0 * sum(rects[t][2] * rects[t][3] for t in mine)  # 3 B in (10-bit 4:2:0) + 8 B out per pixel
    ms_step = 1e3 * elapsed / steps
    achieved = alg_bytes_rank / (ms_step * 1e-3) / 1e9 if n else 0.0
    return {
        "metric": "megapixels/sec YUV420->RGBA (8x8 grid of 1080p 10-bit tiles, one canvas sharded over the GPUs)",
        "value": round(mp * steps / elapsed, 1),
        "unit": "megapixels/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(ms_step, 5),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "dry-run (no GPU, nothing converted: numbers are meaningless)" if dry else "synthetic",
        "value_basis": f"median of {len(region_s)} timed regions of {steps} canvases each (min {1e3 * min(region_s) / steps:.4f}, max {1e3 * max(region_s) / steps:.4f} "
                       f"ms/canvas), MAX over ranks; canvas and pixels resident in every rank's HBM",
        "config": {
            "workload": "15360x8640 canvas = 8x8 grid of 1920x1080 10-bit YUV420 BT.709 limited tiles -> RGBA (10 bits in 16-bit containers, API defaults), bilinear, "
                        f"contiguous blocks of tiles per rank ({n} on rank 0), one batched launch per rank and step",
            "kernel": kernel_name,
            "tiles_per_rank": [len(farm.shard(len(rects), r, world)) for r in range(world)],
            "parallelism": f"tiles of one canvas sharded over {world} rank(s), no collective",
            "preheat_ms": args.preheat_ms,
            "repeats": len(region_s),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": None,
            "algorithmic_bytes_per_launch": int(alg_bytes_rank),
            "kernel_ms": round(ms_step, 5),
            "note": "rank 0's batched launch: its tiles' algorithmic bytes (11 B/pixel) over the step time (back-to-back launches, host clock)",
        },
        "host_to_host": {
            "what": "the same canvas from HOST planes to HOST pixels through avifhipImageYUVToRGBRects, staging included; MAX over ranks",
            "ms_per_canvas": round(1e3 * host_elapsed / host_steps, 3),
            "megapixels_per_s": round(mp * host_steps / host_elapsed, 1),
            "rank0_bytes_up": int(up.value),
            "rank0_bytes_down": int(down.value),
            "rank0_link_GBps": round((up.value + down.value) / (host_elapsed / host_steps) / 1e9, 1),
        },
        "cpu_baseline": None,
    }


def run_in_process(args):
    """`--in-process`: ONE process, the library's own device farm (include/avifhip.h avifhipSetDeviceSet; libavif_amd/csrc/api_farm.cpp) over --gpus N
    devices.  A step = one host-resident image converted by ONE synchronous call (avifhipImageYUVToRGB -- what libavif's seam A / seam B hand
    over), its rows shared out over the devices, every device on its own host link: STRONG scaling of one image, host to host.  This is NOT the
    contract's HBM-resident `value` (that is the default run): the line says so in "metric" and "data"."""
    from libavif_amd import abi, native, synth

    lib = native.load()
    have = lib.avifhipDeviceCount()
    if have <= 0:
        raise SystemExit("bench.py: no HIP device visible -- there is no CPU fallback for the product path")
    named = os.environ.get("AVIFHIP_BENCH_DEVICES")
    devices = [int(x) for x in named.split(",")] if named else list(range(args.gpus))
    if not named and have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {have} HIP device(s) are visible -- refusing to report a {args.gpus}-GPU number "
                         "(AVIFHIP_BENCH_DEVICES=0,0 names a set explicitly)")
    native.check(lib.avifhipSetDeviceSet((C.c_int * len(devices))(*devices), len(devices)), "avifhipSetDeviceSet")
    lib.avifhipSetArithmetic(0)
    if args.workload == "cfg5":
        w, h, depth, what = 15360, 8640, 10, "15360x8640 stitched canvas of the 8x8 grid of 1920x1080 10-bit YUV420 BT.709 limited tiles -> RGBA (10 bits in 16-bit containers, API defaults), bilinear"
    else:
        w, h, depth, what = WIDTH, HEIGHT, 8, "7680x4320 8-bit YUV420 BT.709 limited -> RGBA8, bilinear chroma upsampling, API defaults (integer path)"
    img = abi.make_yuv(w, h, depth, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, abi.AVIF_MATRIX_COEFFICIENTS_BT709)
    synth.fill_yuv(img, 0x12345678)
    rgb = abi.make_rgb(w, h, depth, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR, avoid_libyuv=False)
    steps, warmup = max(1, min(args.steps, 10)), max(1, min(args.warmup, 2))

    def run(k):
        for _ in range(k):
            native.check(lib.avifhipImageYUVToRGB(img.struct, rgb.struct), "avifhipImageYUVToRGB")

    run(2)  # (the workers' contexts, device twins and download helpers are built by the first call)
    regions = []
    for _ in range(max(3, args.repeats // 3)):
        run(warmup)
        t0 = time.perf_counter()
        run(steps)
        regions.append(time.perf_counter() - t0)
    elapsed = median(regions)
    workers = []
    for k in range(lib.avifhipLastFarmWorkers()):
        dev, b, e, up, down = C.c_int(-1), C.c_uint32(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        native.check(lib.avifhipLastFarmTransferBytes(k, C.byref(dev), C.byref(b), C.byref(e), C.byref(up), C.byref(down)), "avifhipLastFarmTransferBytes")
        workers.append({"device": dev.value, "rows": [b.value, e.value], "bytes_up": up.value, "bytes_down": down.value})
    up, down = C.c_uint64(0), C.c_uint64(0)
    lib.avifhipLastTransferBytes(C.byref(up), C.byref(down))
    mp = w * h / 1e6
    ms = 1e3 * elapsed / steps
    native.check(lib.avifhipSetDeviceSet(None, 0), "avifhipSetDeviceSet")
    return {
        "metric": "megapixels/sec YUV420->RGBA, HOST to HOST through one synchronous call, rows shared over the GPUs of one process",
        "value": round(mp * steps / elapsed, 1), "unit": "megapixels/s", "n_gpus": len(devices), "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i16" if depth == 8 else "f32",
        "data": "synthetic, host-resident in and out (pageable memory): staging over the host links is inside the timed region",
        "value_basis": f"median of {len(regions)} regions of {steps} calls (min {1e3 * min(regions) / steps:.3f}, max {1e3 * max(regions) / steps:.3f} ms per call), host clock",
        "config": {"workload": what, "kernel": native.last_kernel(), "devices": devices, "in_process": True,
                   "parallelism": f"one process, {len(workers) or 1} worker thread(s), one per entry of the device set; no collective, halo rows uploaded per device"},
        "host_link": {"bytes_up": up.value, "bytes_down": down.value, "GBps": round((up.value + down.value) / (ms * 1e-3) / 1e9, 1), "workers": workers},
        "roofline": None, "cpu_baseline": None,
    }


def _cycle_args(frames):
    from libavif_amd import abi

    n = len(frames)
    imgs = (C.POINTER(abi.avifImage) * n)(*[C.pointer(f[0].struct) for f in frames])
    rgbs = (C.POINTER(abi.avifRGBImage) * n)(*[C.pointer(f[1].struct) for f in frames])
    return n, imgs, rgbs


if __name__ == "__main__":
    main()
