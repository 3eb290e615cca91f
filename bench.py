#!/usr/bin/env python3
"""bench.py -- megapixels/s of libavif's YUV->RGB reformat hot path on MI355X.

A "step" is one pass of the hot path over one synthetic 8K frame: 7680x4320 8-bit YUV 4:2:0, BT.709 limited
range -> RGBA8 with bilinear chroma upsampling (BASELINE.json configs[1]), planes and pixels resident in HBM.
Steps cycle over several distinct frames so the working set (>700 MB) exceeds the 256 MB Infinity Cache.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU): frames are independent units
(grid tiles / sequence frames), every rank converts its own frames, there is no data-path collective
("scaling": "weak"); ranks only meet in the barrier around the timed region and the MAX over rank times.
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
FRAMES_IN_FLIGHT = 4  # distinct frame buffers cycled by the timed loop
ALGORITHMIC_BYTES_PER_PIXEL = 5.5  # 1.5 B read (Y + U/4 + V/4) + 4 B written (RGBA8), SURVEY.md 8d
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arith", choices=["float", "libyuv"], default="float")
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
    return {"value": round(mp * frames / t_total, 2), "unit": "megapixels/s", "cores": 1, "kind": kind,
            "sample": f"{frames} x 7680x4320 8-bit 4:2:0 BT.709 limited -> RGBA8 bilinear frames, libavif built-in float path "
                      f"(avoidLibYUV=1, maxThreads=1), {t_total:.1f} s of CPU; best frame {mp / best:.1f} MP/s",
            "best_value": round(mp / best, 2)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    dist = None
    torch = None
    if world > 1:
        # torch first: its bundled HIP runtime must be the one libavifhip.so binds to (same SONAME)
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from libavif_amd import abi, device, native, synth

    lib = native.load()
    if lib.avifhipDeviceCount() <= 0:
        raise SystemExit("bench.py: no HIP device visible -- there is no CPU fallback for the product path")
    native.check(lib.avifhipSetDevice(local_rank if world > 1 else 0), "avifhipSetDevice")
    lib.avifhipSetArithmetic(1 if args.arith == "float" else 2)

    # ---- synthetic frames, resident in HBM before the timed region ----
    frames = []
    for f in range(FRAMES_IN_FLIGHT):
        img = abi.make_yuv(WIDTH, HEIGHT, 8, abi.AVIF_PIXEL_FORMAT_YUV420, abi.AVIF_RANGE_LIMITED, abi.AVIF_MATRIX_COEFFICIENTS_BT709)
        synth.fill_yuv(img, 0x12345678 + rank * FRAMES_IN_FLIGHT + f)
        rgb = abi.make_rgb(WIDTH, HEIGHT, 8, abi.AVIF_RGB_FORMAT_RGBA, upsampling=abi.AVIF_CHROMA_UPSAMPLING_BILINEAR,
                           avoid_libyuv=(args.arith == "float"), allocate=False)
        dimg = device.DeviceYUV(img)
        drgb = device.DeviceRGB(rgb)
        frames.append((dimg, drgb))
        del img

    def step(k: int) -> None:
        dimg, drgb = frames[k % FRAMES_IN_FLIGHT]
        native.check(lib.avifhipImageYUVToRGBAsync(dimg.struct, drgb.struct, None), "avifhipImageYUVToRGBAsync")

    def fence() -> None:
        native.check(lib.avifhipSynchronize(None), "avifhipSynchronize")
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    for k in range(args.warmup):
        step(k)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    native.check(lib.avifhipSynchronize(None), "avifhipSynchronize")
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
    dimg, drgb = frames[0]
    kernel_ms = lib.avifhipTimeYUVToRGB(dimg.struct, drgb.struct, 5, 50, None)
    # cycle over all frames too (cold Infinity Cache per launch), timed by events in one go
    cyc_ms = []
    for dimg_k, drgb_k in frames:
        cyc_ms.append(lib.avifhipTimeYUVToRGB(dimg_k.struct, drgb_k.struct, 0, 1, None))
    kernel_ms_cold = sum(cyc_ms) / len(cyc_ms)

    mp_per_step = WIDTH * HEIGHT / 1e6
    total_mp = mp_per_step * args.steps * world
    value = total_mp / elapsed
    alg_bytes = ALGORITHMIC_BYTES_PER_PIXEL * WIDTH * HEIGHT
    # roofline uses the per-launch time inside the streaming loop (distinct frames), i.e. wall / steps
    per_launch_ms = 1000.0 * elapsed / args.steps
    achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9

    out = {
        "metric": "megapixels/sec YUV420->RGBA (8K)",
        "value": round(value, 1),
        "unit": "megapixels/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(per_launch_ms, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.arith == "float" else "i32",
        "data": "synthetic",
        "config": {
            "workload": "7680x4320 8-bit YUV420 BT.709 limited -> RGBA8, bilinear chroma upsampling, HBM-resident, "
                        f"{FRAMES_IN_FLIGHT} distinct frames cycled per rank",
            "arithmetic": "libavif built-in fp32 path, byte-exact" if args.arith == "float" else "libyuv fixed-point, byte-exact",
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
            "kernel_ms_streaming": round(per_launch_ms, 5),
            "kernel_ms_same_frame_events": round(kernel_ms, 5),
            "kernel_ms_cold_frame_events": round(kernel_ms_cold, 5),
            "read_only_GBps": round(1.5 * WIDTH * HEIGHT / (per_launch_ms * 1e-3) / 1e9, 1),
        },
    }
    traffic_file = ROOT / "profiles" / "pmc_traffic.json"
    if traffic_file.exists():
        try:
            out["roofline"]["traffic"] = json.loads(traffic_file.read_text()).get("traffic_bytes_per_launch")
        except Exception:
            pass

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(abi, synth, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
