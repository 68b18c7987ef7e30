#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X accelerate backend.

Workload (BASELINE.json configs[1]): 8192x8192 RGBA Q16 BlurImage(radius=0,
sigma=10) — the 79-tap separable Gaussian blur with a Quantum-rounded
intermediate — on device-resident images.  One *step* = one BlurImage call on
one 8192x8192 image per rank (independent images shard across ranks with no
collective: weak scaling).  Metric: Mpixels/s of output, whole job.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 with the `roofline` (dominant kernel, hipEvent
timed on the launch stream) and `cpu_baseline` (the compiled reference's own
OpenMP BlurImage on this host, bounded sample, rank 0 / N=1 only) objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=8192, help="image edge (default: the C2 config)")
    ap.add_argument("--sigma", type=float, default=10.0)
    ap.add_argument("--precision", choices=["exact", "fast"], default="fast",
                    help="fast: f32 accumulation, results within +-1 Quantum level of the reference (the "
                         "tolerance BASELINE.json's north_star states); exact: fp64 in the CPU's operation "
                         "order, bit-identical")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (resize) measurements")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo rehearses the N>1 control "
                         "flow on a box with fewer GPUs than ranks: ranks then share devices)")
    return ap.parse_args()


def profile_begin():
    """Start recording a hipEvent pair around every kernel the library launches, on the
    stream it launches them on (asynchronous: nothing waits until the records are read)."""
    from imagemagick_amd import _lib
    lib = _lib.load()
    lib.MhResetProfileRecords()
    lib.MhSetProfileEnabled(1)


def profile_end():
    """Per-kernel {count, avg_ms, min_ms, max_ms} since profile_begin()."""
    import torch
    from imagemagick_amd import _lib
    lib = _lib.load()
    torch.cuda.synchronize()
    lib.MhSetProfileEnabled(0)
    recs = (_lib.MhKernelProfileRecord * 32)()
    n = lib.MhGetProfileRecords(recs, 32)
    out = {}
    for i in range(min(n, 32)):
        r = recs[i]
        out[r.kernel_name.decode()] = {"count": int(r.count), "avg_ms": r.total_ms / max(r.count, 1),
                                       "min_ms": r.min_ms, "max_ms": r.max_ms}
    lib.MhResetProfileRecords()
    return out


def kernel_profile(im, fn, reps):
    """Average per-launch duration (ms) of every kernel `fn` launches, from the
    library's hipEvent records on the launch stream."""
    import ctypes
    import torch
    from imagemagick_amd import _lib
    lib = _lib.load()
    lib.MhResetProfileRecords()
    lib.MhSetProfileEnabled(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    lib.MhSetProfileEnabled(0)
    recs = (_lib.MhKernelProfileRecord * 32)()
    n = lib.MhGetProfileRecords(recs, 32)
    out = {}
    for i in range(min(n, 32)):
        r = recs[i]
        out[r.kernel_name.decode()] = {"count": int(r.count), "avg_ms": r.total_ms / max(r.count, 1),
                                       "min_ms": r.min_ms, "max_ms": r.max_ms}
    lib.MhResetProfileRecords()
    return out


def cpu_baseline(sigma):
    """The reference's own CPU/OpenMP BlurImage (oracle/_ref) on a bounded
    sample of the same workload: same distribution, same sigma, smaller frame."""
    import numpy as np
    from oracle import ref
    if not ref.available(False):
        return None
    threads = os.cpu_count() or 1
    ref.set_thread_limit(threads)
    rng = np.random.default_rng(42)
    edge = 1024
    spent = 0.0
    best = None
    while True:
        px = rng.integers(0, 65536, (edge, edge, 4), dtype=np.uint16)
        img = ref.RefImage(px)
        out = img.blur(0.0, sigma)
        sec = out.last_seconds
        spent += sec
        best = (edge, sec)
        del out, img
        # grow the sample until one call takes a few seconds, within a ~30 s budget
        if sec > 4.0 or spent + 4.5 * sec > 30.0 or edge >= 8192:
            break
        edge *= 2
    edge, sec = best
    return {"value": round(edge * edge / sec / 1e6, 3), "unit": "Mpixels/s",
            "cores": int(ref.thread_limit()), "kind": "reference",
            "sample": "%dx%d RGBA Q16 BlurImage(0,%g), reference MagickCore OpenMP path, "
                      "1 call, %.2f s" % (edge, edge, sigma, sec)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    import imagemagick_amd as im
    im.load()
    im.set_precision(im.PRECISION_FAST if args.precision == "fast" else im.PRECISION_EXACT)

    n = args.size
    gen = torch.Generator(device="cuda").manual_seed(42 + rank)
    # uniform uint16 in every channel incl. alpha: exercises the alpha-weighted path
    src = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda",
                        dtype=torch.int16).view(torch.uint16)
    if os.environ.get("MAGICKHIP_BENCH_ZERO"):     # diagnostics only: data-dependent clocking
        src.zero_()
    image = im.Image(src)
    out_holder = {}

    def step():
        out_holder["out"] = im.blur_image(image, 0.0, args.sigma)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    profile_begin()          # hipEvent pairs around every kernel of the timed steps (async records)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = profile_end()
    if distributed:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        pixels = float(n) * n
        ms_per_step = elapsed / args.steps * 1e3
        value = world * pixels * args.steps / elapsed / 1e6
        # dominant kernel: hipEvent-timed on the launch stream, over the timed steps themselves
        conv = {k: v for k, v in prof.items() if k.startswith("conv_") or k == "blur_fused"}
        dominant = max(conv, key=lambda k: conv[k]["avg_ms"]) if conv else None
        roofline = None
        if dominant:
            # algorithmic bytes of one pass: read the frame once, write it once
            bytes_per_launch = 2.0 * pixels * 4 * 2
            ms = conv[dominant]["avg_ms"]
            achieved = bytes_per_launch / (ms * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(dominant)
                except Exception:
                    traffic = None
            # also report the arithmetic rate.  FAST forms the sums on the f16 matrix cores
            # (convolve_mfma.hip: three f16 products per multiply-add, 112-wide Toeplitz band for 79
            # taps), EXACT on the fp64 vector ALU.  Algorithmic flops of one pass:
            # pixels * 4 channels * taps * 2.
            taps = 79 if abs(args.sigma - 10.0) < 1e-9 else None
            alu = None
            if taps:
                flops = pixels * 4 * taps * 2.0
                mfma = args.precision == "fast" and os.environ.get("MAGICKHIP_NO_MFMA") is None
                peak = 2500.0 if mfma else (157.3 if args.precision == "fast" else 78.6)
                alu = {"achieved_tflops": round(flops / (ms * 1e-3) / 1e12, 2), "peak_tflops": peak,
                       "frac": round(flops / (ms * 1e-3) / 1e12 / peak, 4),
                       "unit": "f16 MFMA dense" if mfma else ("f32 vector" if args.precision == "fast"
                                                             else "f64 vector"),
                       "note": "algorithmic multiply-adds only; the matrix-core path executes "
                               "3 x 112/79 = 4.3x as many (hi/lo operand split, band padding)"
                               if mfma else
                               "algorithmic multiply-adds only (alpha weighting, conversions and the "
                               "epilogue are extra work, not counted)"}
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "avg_ms": round(ms, 4),
                        "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in prof.items()},
                        "alu": alu}
        result = {
            "metric": "Mpixels/sec GaussianBlur sigma=10, 8K RGBA Q16",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.precision == "exact" else
                     ("f32" if os.environ.get("MAGICKHIP_NO_MFMA") else "f16x2 products, f32 accumulate"),
            "data": "synthetic",
            "tolerance": "bit-identical to the reference CPU path" if args.precision == "exact" else
                         "each pass within +-1 Quantum level of the reference CPU pass on the same input; the two-pass "
                         "blur within +-1 wherever the intermediate alpha exceeds a few levels "
                         "(tests/test_gpu_parity.py, DESIGN.md section 2)",
            "config": {"workload": "%dx%d RGBA Q16 BlurImage(radius=0,sigma=%g): 79-tap row pass + "
                                   "79-tap column pass, Quantum-rounded intermediate, edge clamp, "
                                   "alpha-weighted colour channels; one independent image per GPU"
                                   % (n, n, args.sigma),
                       "precision": args.precision, "images_per_step": world},
            "roofline": roofline,
        }
        if not args.no_extra and world == 1:
            result["extra"] = extra_measurements(im, torch, args)
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args.sigma)
            except Exception as exc:  # the baseline is a report, never a reason to lose the line
                result["cpu_baseline"] = {"error": str(exc)}
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def timed(torch, fn, reps):
    fn()            # two warm-up calls: the result of call n is released only after call n+1
    fn()            # allocated its own, so the caching allocator needs two blocks before it is warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def extra_measurements(im, torch, args):
    """Secondary numbers reported next to the headline (not `value`): the other precision
    mode of the blur, the C3 Lanczos 4x resize (8192^2 -> 32768^2, float Quantum), one image of
    the C4 batch (sRGB->Lab + ContrastStretch) and the two C5 operators."""
    extra = {}
    try:
        n = args.size
        gen = torch.Generator(device="cuda").manual_seed(1)
        src = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda",
                            dtype=torch.int16).view(torch.uint16)
        image = im.Image(src)
        other = im.PRECISION_EXACT if args.precision == "fast" else im.PRECISION_FAST
        im.set_precision(other)
        sec = timed(torch, lambda: im.blur_image(image, 0.0, args.sigma), 5)
        extra["blur_%s_Mpixels_per_s" % ("exact" if args.precision == "fast" else "fast")] = \
            round(n * n / sec / 1e6, 1)
        im.set_precision(im.PRECISION_FAST if args.precision == "fast" else im.PRECISION_EXACT)
        # GaussianBlurImage: the 2-D Gaussian kernel, separated in FAST mode
        if args.precision == "fast":
            sec = timed(torch, lambda: im.gaussian_blur_image(image, 0.0, args.sigma), 5)
            extra["gaussian_blur_2d_kernel_Mpixels_per_s"] = round(n * n / sec / 1e6, 1)
        # reference point for the roofline: what a plain device copy of the same frame reaches
        # (read 537 MB + write 537 MB, torch's copy kernel)
        mirror = torch.empty_like(src)
        sec = timed(torch, lambda: mirror.copy_(src), 10)
        extra["device_copy_GBps"] = round(2.0 * src.numel() * 2 / sec / 1e9, 1)
        del mirror, src, image
        torch.cuda.empty_cache()
        # The kernels' speed depends on the data (the same instruction stream runs ~20 % faster on
        # an all-zero frame: clocks, i.e. power): the headline uses uniform noise, the worst case;
        # a smooth frame (low-frequency waves + 1 % noise, varying alpha) is closer to a photograph
        yy, xx = torch.meshgrid(torch.arange(n, device="cuda", dtype=torch.float32),
                                torch.arange(n, device="cuda", dtype=torch.float32), indexing="ij")
        smooth = torch.stack([torch.sin(xx * 0.003 + c) * torch.cos(yy * 0.002 - c) for c in range(4)], dim=2)
        smooth = (smooth * 0.45 + 0.5) * 65535.0 + torch.randn((n, n, 4), generator=gen, device="cuda") * 600.0
        smooth = smooth.clamp_(0, 65535).to(torch.int32).to(torch.int16).view(torch.uint16).contiguous()
        del yy, xx
        image = im.Image(smooth)
        sec = timed(torch, lambda: im.blur_image(image, 0.0, args.sigma), 5)
        extra["blur_smooth_frame_Mpixels_per_s"] = round(n * n / sec / 1e6, 1)
        # the same call on a host (pixel-cache) buffer: upload + both passes + download, what a
        # single un-chained operator costs through the MagickCore shim
        host = smooth.view(torch.int16).cpu().numpy().view("uint16")
        host_image = im.Image(host)
        im.blur_image(host_image, 0.0, args.sigma)
        t0 = time.perf_counter()
        for _ in range(2):
            im.blur_image(host_image, 0.0, args.sigma)
        extra["blur_host_buffers_Mpixels_per_s"] = round(2 * n * n / (time.perf_counter() - t0) / 1e6, 1)
        del smooth, image, host, host_image
        torch.cuda.empty_cache()
        # C3: 8192^2 -> 32768^2 Lanczos, float Quantum (17.2 GB result)
        m = 8192
        srcf = torch.rand((m, m, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
        imgf = im.Image(srcf)
        holder = {}

        def resize():
            holder["o"] = None
            holder["o"] = im.resize_image(imgf, 4 * m, 4 * m, "Lanczos")
        sec = timed(torch, resize, 3)
        prof = kernel_profile(im, resize, 2)
        out_px = 16.0 * m * m
        extra["resize_lanczos4x_f32_Mpixels_per_s"] = round(out_px / sec / 1e6, 1)
        extra["resize_kernels_ms"] = {k: round(v["avg_ms"], 3) for k, v in prof.items()}
        # algorithmic bytes: horizontal pass reads 8192x32768 and writes 32768x32768 float RGBA
        if "resize_horizontal" in prof:
            b = (m * 4.0 * m + 16.0 * m * m) * 16
            extra["resize_horizontal_GBps"] = round(b / (prof["resize_horizontal"]["avg_ms"] * 1e-3) / 1e9, 1)
        if "resize_fused" in prof:
            # fused V+H kernel: reads the 8192^2 source, writes the 32768^2 result (float RGBA)
            b = (1.0 * m * m + 16.0 * m * m) * 16
            extra["resize_fused_GBps"] = round(b / (prof["resize_fused"]["avg_ms"] * 1e-3) / 1e9, 1)
        if "resize_vertical" in prof:
            b = (1.0 * m * m + 4.0 * m * m) * 16
            extra["resize_vertical_GBps"] = round(b / (prof["resize_vertical"]["avg_ms"] * 1e-3) / 1e9, 1)
        holder.clear()
        del srcf, imgf
        torch.cuda.empty_cache()
        # C4 (one image of the batch): 4096^2 RGBA Q16 sRGB->Lab + ContrastStretch 2%x1%
        k = 4096
        src4 = torch.randint(-32768, 32768, (k, k, 4), generator=gen, device="cuda",
                             dtype=torch.int16).view(torch.uint16)
        work = src4.clone()

        def c4():
            work.copy_(src4)
            img4 = im.Image(work)
            im.transform_image_colorspace(img4, "Lab")
            im.contrast_stretch_image(img4, 0.02 * k * k, k * k - 0.01 * k * k)
        sec = timed(torch, c4, 5)
        prof = kernel_profile(im, c4, 3)
        extra["c4_lab_contrast_stretch_4096_Mpixels_per_s"] = round(k * k / sec / 1e6, 1)
        extra["c4_kernels_ms"] = {kk: round(v["avg_ms"], 3) for kk, v in prof.items()}
        del src4, work
        # C5: 16384^2 RGBA Q16 Dilate Disk:15, then UnsharpMask(0x10+1+0.02)
        k = 16384
        src5 = torch.randint(-32768, 32768, (k, k, 4), generator=gen, device="cuda",
                             dtype=torch.int16).view(torch.uint16)
        img5 = im.Image(src5)

        def dilate():
            holder["o"] = im.morphology_image(img5, "Dilate", 1, "Disk:15")
        sec = timed(torch, dilate, 2)
        extra["c5_dilate_disk15_16384_Mpixels_per_s"] = round(k * k / sec / 1e6, 1)

        def unsharp():
            holder["o"] = im.unsharp_mask_image(img5, 0.0, 10.0, 1.0, 0.02)
        sec = timed(torch, unsharp, 2)
        extra["c5_unsharp_0x10_16384_Mpixels_per_s"] = round(k * k / sec / 1e6, 1)
        holder.clear()
    except Exception as exc:
        extra["error"] = str(exc)
    return extra


if __name__ == "__main__":
    main()
