#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X accelerate backend.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on):
8192x8192 RGBA Q16 BlurImage(radius=0, sigma=10) — the 79-tap separable Gaussian blur with its
Quantum-rounded intermediate — on device-resident images.  One *step* = one BlurImage call on
one 8192x8192 image per rank (independent images shard across ranks with no collective: weak
scaling).  `value` = Mpixels/s of output, whole job.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      the dominant kernel of the headline: algorithmic bytes / hipEvent-timed launch
  modes         both precision modes of the blur as first-class objects, each with its roofline
  resize        BASELINE's other half of the metric: C3, 8192^2 -> 32768^2 Lanczos, float Quantum
  configs       C4 (sRGB->Lab + ContrastStretch), C5 (Dilate Disk:15, UnsharpMask): Mpixels/s,
                one roofline object per kernel, and the reference's CPU figure beside each
  sustained     >= 2 s of back-to-back blur calls (power-limit behaviour)
  cpu_baseline  the compiled reference's OpenMP BlurImage on this host (bounded sample)

Other configurations as the step (`--config`): c4 = the batch of 512 images sharded over the
ranks (strong scaling, no collective), c5 = one 16384^2 image row-sharded over the ranks with
halo rows (Dilate Disk:15 + UnsharpMask; strong scaling), equalize = one 16384^2 image
row-sharded, EqualizeImage with the one all-reduce of the 65536 x channels table over RCCL.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6      # 1/2 of the guide's 157.3 TFLOP/s fp32 vector peak
I8_MFMA_PEAK_TOPS = 5000.0        # dense i8 matrix peak (= the fp8 figure of MI355X_MICROARCH.md; measured 4200)

# What each precision mode of the blur computes in, and what it promises
MODE_DTYPE = {
    "fast": "f16x2 products, f32 accumulate for the colour sums of both passes; the row pass's alpha (the "
            "column pass's weights) as u8 digit products on the i8 matrix cores: exact i32 sums, f64 rounding",
    "exact": "u8 digit products on the i8 matrix cores, exact i32 sums, f64 epilogue; the few pixels the "
             "error bound cannot decide recomputed in the reference's f64 operation order",
    "hdri": "f64 in the CPU's operation order, float Quantum",
}
MODE_TOLERANCE = {
    "fast": "within +-1 Quantum level of the reference CPU path on ANY input, by construction: the intermediate "
            "alpha is the reference's level bit for bit, the intermediate colour is NOT rounded (the reference's "
            "rounding moves the value the column pass rounds by at most 0.5 level), the f16 sums add < 0.1",
    "exact": "bit-identical to the reference CPU path",
    "hdri": "bit-identical to the reference CPU path (float Quantum)",
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "Mpixels/sec GaussianBlur σ=10 + Lanczos 4× resize, 8K RGBA; %HBM-roofline"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["c2", "c4", "c5", "equalize"], default="c2",
                    help="what one step is (default c2: the headline blur)")
    ap.add_argument("--size", type=int, default=0, help="image edge (default: the config's own)")
    ap.add_argument("--sigma", type=float, default=10.0)
    ap.add_argument("--precision", choices=["exact", "fast"], default="fast",
                    help="fast: f16x2 products on the matrix cores, f32 accumulation, results within +-1 "
                         "Quantum level of the reference (the tolerance BASELINE.json's north_star states); "
                         "exact: fp64 in the CPU's operation order, bit-identical")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline only: no modes / resize / configs")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from profiles/pmc_traffic.json instead of two rocprofv3 --pmc passes")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of back-to-back calls (0: skip)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N > 1 only: skip the short equalize / c5 / c4 runs reported beside the headline")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo rehearses the N>1 control "
                         "flow on a box with fewer GPUs than ranks: ranks then share devices)")
    return ap.parse_args()


# ------------------------------------------------------------------ profiling
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
    recs = (_lib.MhKernelProfileRecord * 48)()
    n = lib.MhGetProfileRecords(recs, 48)
    out = {}
    for i in range(min(n, 48)):
        r = recs[i]
        out[r.kernel_name.decode()] = {"count": int(r.count), "avg_ms": r.total_ms / max(r.count, 1),
                                       "min_ms": r.min_ms, "max_ms": r.max_ms}
    lib.MhResetProfileRecords()
    return out


def kernel_profile(im, fn, reps):
    """Average per-launch duration (ms) of every kernel `fn` launches, from the
    library's hipEvent records on the launch stream."""
    profile_begin()
    for _ in range(reps):
        fn()
    return profile_end()


def timed(torch, fn, reps):
    fn()            # two warm-up calls: the result of call n is released only after call n+1
    fn()            # allocated its own, so the caching allocator needs two blocks before it is warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


_TRAFFIC = None


def pmc_traffic(kernel):
    """HBM bytes per launch from the PMC passes kept under profiles/ (FETCH_SIZE doubled as the
    guide prescribes for gfx950, + WRITE_SIZE; tools/import_profiles.py writes the file)."""
    global _TRAFFIC
    if _TRAFFIC is None:
        try:
            _TRAFFIC = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            _TRAFFIC = {}
    return _TRAFFIC.get(kernel)


def live_traffic(mode, sigma, kernel_substring="blur_fused"):
    """HBM bytes per launch of the headline kernel, measured NOW: two `rocprofv3 --kernel-trace
    --pmc` passes (FETCH_SIZE, WRITE_SIZE: separate passes, no other trace domain) of
    tools/time_blur_modes.py on this GPU, corrected as profiles/pmc_traffic.json is (FETCH_SIZE in
    KiB and doubled on gfx950, WRITE_SIZE in KiB).  None when rocprofv3 is missing or fails; about
    five seconds."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return None
    totals = {}
    work = tempfile.mkdtemp(prefix="mh_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [tool, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pass", "--",
                   sys.executable, os.path.join(ROOT, "tools", "time_blur_modes.py"), mode, "8192", str(sigma), "4"]
            done = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180)
            if done.returncode != 0:
                return None
            values = []
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if kernel_substring in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        values.append(float(row["Counter_Value"]))
            if not values:
                return None
            totals[counter] = sum(values) / len(values)
        return int(round(totals["FETCH_SIZE"] * 1024.0 * 2.0 + totals["WRITE_SIZE"] * 1024.0))
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def roofline(kernel, algorithmic_bytes, avg_ms, traffic_key=None):
    achieved = algorithmic_bytes / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic(traffic_key or kernel), "avg_ms": round(avg_ms, 4),
            "algorithmic_bytes": int(algorithmic_bytes)}


def issue_roofline(mode, avg_ms):
    """What occupies a one-launch blur kernel besides HBM (its traffic is 1.03x the compulsory bytes), from the SQ
    counters kept under profiles/ (tools/collect_profiles.sh; per launch of the 8192^2 frame): how busy the vector
    pipe, the matrix pipe and the LDS are, the share of wave cycles spent waiting, the shader clock — and the
    instruction counts priced at their issue cost (a 16x16x32 f16 / 16x16x64 i8 matrix instruction occupies its
    SIMD's matrix pipe for 16 cycles, a vector instruction its pipe for 4: wave64 on 16 lanes; 1024 SIMDs at the
    nominal 2.4 GHz).  No pipe is saturated: the kernel is one 16-wave workgroup per CU (its LDS ring leaves no room
    for a second) walking in two barrier intervals of dependent chains, and the time is the chains' latency."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json"))):
        try:
            table = json.load(open(path))
        except Exception:
            continue
        if mode in table and "SQ_INSTS_VALU" in table[mode]:
            best = (path, table[mode])
    if best is None:
        return None
    path, c = best
    simd_cycles_per_ms = 1024.0 * 2.4e9 * 1e-3
    matrix_ms = c["SQ_INSTS_MFMA"] * 16.0 / simd_cycles_per_ms
    vector_ms = c["SQ_INSTS_VALU"] * 4.0 / simd_cycles_per_ms
    out = {"limiter": "latency: dependent chains between two workgroup barriers, one 16-wave workgroup per CU "
                      "(no pipe saturated; the LDS ring leaves no room for a second workgroup)",
           "kernel_ms": round(avg_ms, 4),
           "matrix_pipe": {"instructions": int(c["SQ_INSTS_MFMA"]), "cycles_per_instruction": 16,
                           "busy_ms": round(matrix_ms, 4), "frac": round(matrix_ms / avg_ms, 4)},
           "vector_issue": {"instructions": int(c["SQ_INSTS_VALU"]), "cycles_per_instruction": 4,
                            "busy_ms": round(vector_ms, 4), "frac": round(vector_ms / avg_ms, 4),
                            "lane_instructions_per_pixel": round(c["SQ_INSTS_VALU"] * 64.0 / (8192.0 * 8192.0), 1)},
           "perfect_overlap_ms": round(max(matrix_ms, vector_ms), 4),
           "source": os.path.relpath(path, ROOT)}
    busy = float(c.get("SQ_BUSY_CU_CYCLES") or 0)
    if busy > 0:
        counters = {"vector_pipe_busy": round(c.get("SQ_ACTIVE_INST_VALU", 0) / busy, 3),
                    "matrix_pipe_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4.0 * busy), 3),
                    "lds_busy": round(c.get("SQ_LDS_IDX_ACTIVE", 0) / busy, 3),
                    "lds_bank_conflict_share": c.get("SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE"),
                    "wait_any_share_of_wave_cycles": c.get("SQ_WAIT_ANY/SQ_WAVE_CYCLES"),
                    "wait_inst_lds_share_of_wave_cycles": round(c.get("SQ_WAIT_INST_LDS", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3)}
        clock = clock_under_load("blur_fused_hybrid" if mode == "fast" else "blur_fused_exact")
        if clock:
            counters["shader_clock_GHz"] = clock
        out["counters"] = counters
    return out


def clock_under_load(kernel):
    """The shader clock a kernel holds (GHz; GRBM_GUI_ACTIVE / duration, profiles/clock_under_load.json): the fp64
    vector unit beside 4 TB/s of stores runs C3's kernel at 1.8 GHz, not the nominal 2.4 the peaks are quoted at."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "clock_under_load.json"))).get(kernel)
    except Exception:
        return None


def add_measured_ceiling(node, copy_gbps):
    """frac_of_measured_copy beside every HBM `frac`: this box's own device-to-device copy rate
    (extra.device_copy_GBps) is what a kernel that only streams reaches here."""
    if isinstance(node, dict):
        if node.get("unit") == "GB/s" and "achieved" in node and "frac" in node:
            node["frac_of_measured_copy"] = round(node["achieved"] / copy_gbps, 4)
        for value in node.values():
            add_measured_ceiling(value, copy_gbps)
    elif isinstance(node, list):
        for value in node:
            add_measured_ceiling(value, copy_gbps)


def traffic_key(mode, kernel):
    """How profiles/pmc_traffic.json names a blur kernel: the fused kernels by their own name, the
    two-pass fp64 kernels by workload (exact: Q16, hdri: float Quantum)."""
    if kernel.startswith("blur_fused"):
        return kernel
    return {"exact": "exact:", "hdri": "hdri:"}.get(mode, "") + kernel


def kernel_rooflines(prof, bytes_by_kernel, prefix=""):
    out = {}
    for name, rec in prof.items():
        if name in bytes_by_kernel:
            out[name] = roofline(name, bytes_by_kernel[name], rec["avg_ms"], prefix + name)
        else:
            out[name] = {"avg_ms": round(rec["avg_ms"], 4)}
    return out


# ------------------------------------------------------------------ CPU baselines
def _ref():
    from oracle import ref
    if not ref.available(False):
        return None
    return ref


def cpu_baseline_blur(sigma, edge=8192, calls=3, im=None, torch=None):
    """The reference's own CPU/OpenMP BlurImage (oracle/_ref) on the full BASELINE frame: the
    median of `calls` calls on edge x edge RGBA Q16, same distribution as the GPU
    workload.  About 13 s per call on the 128-thread hosts of the pool; MAGICKHIP_BENCH_CPU_EDGE
    bounds the sample on a slower host."""
    import numpy as np
    ref = _ref()
    if ref is None:
        return None
    ref.set_thread_limit(os.cpu_count() or 1)
    edge = int(os.environ.get("MAGICKHIP_BENCH_CPU_EDGE", edge))
    rng = np.random.default_rng(42)
    px = rng.integers(0, 65536, (edge, edge, 4), dtype=np.uint16)
    image = ref.RefImage(px)
    seconds = []
    last = None
    for _ in range(calls):
        out = image.blur(0.0, sigma)
        seconds.append(out.last_seconds)
        last = out.numpy() if im is not None else None
        del out
        if sum(seconds) > 75.0:            # a slow host: what has been measured so far is the sample
            break
    sec = sorted(seconds)[len(seconds) // 2]
    shares = None
    if (im is not None) and (last is not None):
        # the share of samples on which each mode returns the reference's own level, same frame
        try:
            shares = {}
            dev = torch.from_numpy(px.view(np.int16)).to("cuda").view(torch.uint16)
            before = im.get_precision()
            for name, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
                im.set_precision(precision)
                got = im.blur_image(im.Image(dev), 0.0, sigma).numpy()
                diff = np.abs(got.astype(np.int32) - last.astype(np.int32))
                shares[name] = {"identical": round(float((diff == 0).mean()), 6), "max_abs_diff": int(diff.max())}
                del got, diff
            im.set_precision(before)
            del dev
            torch.cuda.empty_cache()
        except Exception as exc:
            shares = {"error": "%s: %s" % (type(exc).__name__, exc)}
    del last
    return {"identical_share": shares,
            "value": round(edge * edge / sec / 1e6, 3), "unit": "Mpixels/s",
            "cores": int(ref.thread_limit()), "kind": "reference",
            "sample": "%dx%d RGBA Q16 BlurImage(0,%g), reference MagickCore OpenMP path, median of %d calls "
                      "(%s s)" % (edge, edge, sigma, len(seconds), ", ".join("%.2f" % t for t in seconds))}


def cpu_baseline_configs():
    """The reference's OpenMP path on bounded samples of C3 / C4 / C5 (a few seconds each)."""
    import numpy as np
    ref = _ref()
    if ref is None:
        return {}
    threads = os.cpu_count() or 1
    ref.set_thread_limit(threads)
    ref.set_thread_limit(threads, True)
    rng = np.random.default_rng(43)
    out = {}
    try:
        px = rng.integers(0, 65536, (1024, 1024, 4), dtype=np.uint16)
        image = ref.RefImage(px)
        seconds = sorted(image.blur(0.0, 2.0).last_seconds for _ in range(5))
        out["c1_blur_1024_sigma2"] = {"value": round(1024.0 * 1024.0 / seconds[2] / 1e6, 2), "unit": "Mpixels/s",
                                      "ms": round(seconds[2] * 1e3, 3), "cores": int(ref.thread_limit(False)), "kind": "reference",
                                      "sample": "1024x1024 RGBA Q16 BlurImage(0x2), reference MagickCore OpenMP path, "
                                                "median of 5 calls"}
    except Exception as exc:
        out["c1_blur_1024_sigma2"] = {"error": str(exc)[:200]}

    def entry(value_px, sec, sample, hdri=False):
        return {"value": round(value_px / sec / 1e6, 3), "unit": "Mpixels/s",
                "cores": int(ref.thread_limit(hdri)), "kind": "reference",
                "sample": "%s, reference MagickCore OpenMP path, 1 call, %.2f s" % (sample, sec)}
    try:
        m = 8192                                   # (the whole C3 frame: 17 GB of result in the reference's pixel cache, ~11 s)
        src = (rng.random((m, m, 4), dtype=np.float32) * 65535.0).astype(np.float32)
        r = ref.RefImage(src).resize(4 * m, 4 * m, "Lanczos")
        out["c3_resize"] = entry(16.0 * m * m, r.last_seconds,
                                 "%dx%d -> %dx%d Lanczos ResizeImage, float Quantum RGBA" % (m, m, 4 * m, 4 * m), True)
        del r, src
        k = 4096
        px = rng.integers(0, 65536, (k, k, 4), dtype=np.uint16)
        img = ref.RefImage(px)
        img.colorspace("Lab")
        sec = img.last_seconds
        img.contrast_stretch(0.02 * k * k, k * k - 0.01 * k * k)
        sec += img.last_seconds
        out["c4_lab_contrast_stretch"] = entry(float(k) * k, sec, "%dx%d RGBA Q16 sRGB->Lab + ContrastStretch 2%%x1%%" % (k, k))
        del img
        d = 4096                                   # (a sixteenth of C5's pixels)
        px = rng.integers(0, 65536, (d, d, 4), dtype=np.uint16)
        r = ref.RefImage(px).morphology("Dilate", 1, "Disk:15")
        out["c5_dilate_disk15"] = entry(float(d) * d, r.last_seconds, "%dx%d RGBA Q16 Dilate Disk:15" % (d, d))
        del r
        r = ref.RefImage(px).set_artifact("convolve:scale", "!").morphology("Convolve", 1, "Disk:15")
        out["c5_convolve_disk15"] = entry(float(d) * d, r.last_seconds,
                                          "%dx%d RGBA Q16 Convolve Disk:15, convolve:scale='!'" % (d, d))
        del r
        # the reduction, whole frame (Q16 and float), and the reference's own device benchmark chain
        m = 8192
        px = rng.integers(0, 65536, (m, m, 4), dtype=np.uint16)
        r = ref.RefImage(px).resize(2048, 2048, "Lanczos")
        reduce_q16 = entry(2048.0 * 2048.0, r.last_seconds, "8192x8192 -> 2048x2048 Lanczos ResizeImage, Q16 RGBA (%.1f source Mpixel/s)"
                           % (float(m) * m / r.last_seconds / 1e6))
        del r
        r = ref.RefImage(px.astype(np.float32)).resize(2048, 2048, "Lanczos")
        reduce_hdri = entry(2048.0 * 2048.0, r.last_seconds, "8192x8192 -> 2048x2048 Lanczos ResizeImage, float Quantum RGBA "
                            "(%.1f source Mpixel/s)" % (float(m) * m / r.last_seconds / 1e6), True)
        del r, px
        out["resize_reduce"] = {"q16": reduce_q16, "hdri": reduce_hdri}
        chain = {}
        for label, px in (("xc_none", np.zeros((1536, 2048, 4), dtype=np.uint16)),
                          ("random", rng.integers(0, 65536, (1536, 2048, 4), dtype=np.uint16))):
            image = ref.RefImage(px)
            rounds = []
            for _ in range(3):                     # (RunOpenCLBenchmark: the first round is not timed)
                b = image.blur(10.0, 3.5)
                usm = b.unsharp(2.0, 2.0, 50.0, 10.0)
                rs = usm.resize(640, 480, "Lanczos")
                rounds.append(b.last_seconds + usm.last_seconds + rs.last_seconds)
            sec = sorted(rounds[1:])[0]
            chain[label] = entry(2048.0 * 1536.0, sec, "2048x1536 RGBA Q16 %s: BlurImage(10,3.5) -> UnsharpMaskImage(2,2,50,10) -> "
                                 "ResizeImage(640x480, Lanczos), best of 2 timed rounds" % label)
        out["reference_device_benchmark"] = chain
        u = 4096
        px = rng.integers(0, 65536, (u, u, 4), dtype=np.uint16)
        r = ref.RefImage(px).unsharp(0.0, 10.0, 1.0, 0.02)
        out["c5_unsharp"] = entry(float(u) * u, r.last_seconds, "%dx%d RGBA Q16 UnsharpMask(0x10+1+0.02)" % (u, u))
        del r
    except Exception as exc:   # a baseline is a report, never a reason to lose the line
        out["error"] = str(exc)
    return out


# ------------------------------------------------------------------ workloads
def random_q16(torch, gen, rows, cols):
    # uniform uint16 in every channel incl. alpha: exercises the alpha-weighted path
    return torch.randint(-32768, 32768, (rows, cols, 4), generator=gen, device="cuda",
                         dtype=torch.int16).view(torch.uint16)


def blur_mode(im, torch, image, sigma, mode, reps=24, ramp=0.3):
    """One precision mode of the blur as a first-class object: rate + roofline of its dominant
    kernel.  mode: fast | exact | hdri (float Quantum frame, EXACT).  Same clock ramp as the headline and >= 20 timed calls."""
    pixels = float(image.rows) * image.columns
    im.set_precision(im.PRECISION_EXACT if mode in ("exact", "hdri") else im.PRECISION_FAST)
    holder = {}

    def call():
        holder["o"] = im.blur_image(image, 0.0, sigma)
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp:
        call()
        torch.cuda.synchronize()
    sec = timed(torch, call, reps)
    prof = kernel_profile(im, call, max(2, reps // 2))
    conv = {k: v for k, v in prof.items() if k.startswith("conv_") or k.startswith("blur_fused")}
    dominant = max(conv, key=lambda k: conv[k]["avg_ms"])
    frame = pixels * (16.0 if mode == "hdri" else 8.0)
    # algorithmic bytes per launch: a pass (or the fused operator) reads the frame once and writes it once
    out = {"Mpixels_per_s": round(pixels / sec / 1e6, 1), "ms": round(sec * 1e3, 4), "calls": reps,
           "dtype": MODE_DTYPE[mode], "tolerance": MODE_TOLERANCE[mode],
           "launches": "one (row + column pass fused, intermediate in LDS)" if dominant.startswith("blur_fused")
                       else "two (row pass, column pass; intermediate through HBM)",
           "roofline": roofline(dominant, 2.0 * frame, conv[dominant]["avg_ms"], traffic_key(mode, dominant)),
           "operator_frac_of_compulsory_bytes": round(2.0 * frame / sec / 1e9 / HBM_PEAK_GBS, 4),
           "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in prof.items()}}
    ntaps = im.optimal_kernel_width_1d(0.0, sigma)
    if dominant.startswith("blur_fused") and mode in ("fast", "exact") and image.rows == 8192 and image.columns == 8192:
        issue = issue_roofline(mode, conv[dominant]["avg_ms"])
        if issue:
            out["issue_roofline"] = issue
    if dominant.startswith("blur_fused_exact"):
        out["compute_roofline"] = i8_roofline(pixels, ntaps, conv[dominant]["avg_ms"], dominant)
    elif mode in ("exact", "hdri"):
        # the fp64 kernels are bound by the vector unit, not by HBM: K taps x 4 channels of
        # v_fma_f64 per pixel per pass (v_fma_f64 issues at half the fp32 rate: 256 CUs x 4 SIMDs
        # x 16 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s)
        flops = pixels * 4.0 * ntaps * 2.0
        achieved = flops / (conv[dominant]["avg_ms"] * 1e-3) / 1e12
        out["compute_roofline"] = {"bound": "valu_f64", "kernel": dominant, "achieved": round(achieved, 2),
                                   "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(achieved / FP64_VALU_PEAK_TFLOPS, 4),
                                   "flops_per_launch": int(flops), "taps": int(ntaps)}
    holder.clear()
    return out


def i8_roofline(pixels, ntaps, avg_ms, kernel):
    """The matrix-core side of the exact-integer kernels: executed i8 multiply-adds per launch.
    A 16-output tile of 16 entries costs 14 digit products per band chunk (alpha-weighted RGBA), the
    first chunk 64 slots wide, the second (kernels of more than 49 taps) issued as 32 slots at the
    same instruction time; the f16 column pass of the FAST mode adds 3 products x ceil((K+15)/32)
    chunks of 32 slots."""
    exact_column = kernel == "blur_fused_exact"
    chunks = 1 if ntaps + 15 <= 64 else 2
    tiles = pixels * 4.0 / 256.0                      # 16 entries x 16 outputs per tile and pass
    instructions = tiles * 14.0 * chunks * (2.0 if exact_column else 1.0)
    seconds = avg_ms * 1e-3
    # an instruction slot is worth 16*16*64 multiply-adds whether it is issued as 64 or as 32 slots
    tops = instructions * 16.0 * 16.0 * 64.0 * 2.0 / seconds / 1e12
    return {"bound": "mfma_i8", "kernel": kernel, "achieved": round(tops, 1), "peak": I8_MFMA_PEAK_TOPS,
            "unit": "TOPS (instruction slots x 16x16x64)", "frac": round(tops / I8_MFMA_PEAK_TOPS, 4),
            "matrix_instructions_per_launch": int(instructions), "taps": int(ntaps),
            "note": "executed digit products, not algorithmic multiply-adds: 14 products per chunk replace "
                    "one fp64 multiply-add chain per tap; measured issue ceiling 4200 TOPS "
                    "(tools/ubench/mfma_i8_shapes.hip)"}


def resize_config(im, torch, gen):
    """C3: 8192^2 -> 32768^2 Lanczos, float Quantum RGBA (17.2 GB result)."""
    m = 8192
    srcf = torch.rand((m, m, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
    imgf = im.Image(srcf)
    holder = {}

    def resize():
        holder["o"] = None
        holder["o"] = im.resize_image(imgf, 4 * m, 4 * m, "Lanczos")
    out_px = 16.0 * m * m
    px16 = 16.0                                   # bytes per float RGBA pixel
    bytes_by_kernel = {
        "resize_vertical": (1.0 * m * m + 4.0 * m * m) * px16,       # 8192^2 in, 8192x32768 out
        "resize_horizontal": (4.0 * m * m + 16.0 * m * m) * px16,    # 8192x32768 in, 32768^2 out
        "resize_mfma": (1.0 * m * m + 16.0 * m * m) * px16,          # one launch: 8192^2 in, 32768^2 out
        "resize_stream": (1.0 * m * m + 16.0 * m * m) * px16,        # one launch on the vector pipe (FAST default)
        "resize_stream_careful": 0.0,                                # the (normally empty) launch behind it
    }
    compulsory = (1.0 * m * m + 16.0 * m * m) * px16

    def measure():
        sec = timed(torch, resize, 3)
        prof = kernel_profile(im, resize, 2)
        holder.clear()
        kernels = kernel_rooflines(prof, bytes_by_kernel)
        kernel_ms = sum(v["avg_ms"] for v in prof.values())
        dominant = max(prof, key=lambda k: prof[k]["avg_ms"])
        if kernels.get(dominant) is not None and clock_under_load(dominant):
            kernels[dominant]["shader_clock_GHz"] = clock_under_load(dominant)
        return {"Mpixels_per_s": round(out_px / sec / 1e6, 1), "ms": round(sec * 1e3, 3),
                "kernel_only_Mpixels_per_s": round(out_px / (kernel_ms * 1e-3) / 1e6, 1),
                "roofline": kernels.get(dominant),
                "operator_frac_of_compulsory_bytes": round(compulsory / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "kernels": kernels}
    out = {"workload": "8192x8192 -> 32768x32768 Lanczos ResizeImage, float Quantum RGBA (BASELINE configs[2])",
           "dtype": "f64 accumulation, float Quantum"}
    out.update(measure())                         # the precision mode of the line (--precision)
    # both modes side by side (EXACT: the reference's own operation order, bit-identical; FAST: fused
    # multiply-adds, within one float ULP)
    modes = {}
    try:
        for name, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
            if PRECISION_IS_FAST[0] == (name == "fast"):
                modes[name] = {k: out[k] for k in ("Mpixels_per_s", "ms", "operator_frac_of_compulsory_bytes")}
                modes[name]["kernels_ms"] = {k: v["avg_ms"] for k, v in out["kernels"].items()}
                continue
            im.set_precision(precision)
            other = measure()
            modes[name] = {k: other[k] for k in ("Mpixels_per_s", "ms", "operator_frac_of_compulsory_bytes")}
            modes[name]["kernels_ms"] = {k: v["avg_ms"] for k, v in other["kernels"].items()}
    finally:
        im.set_precision(im.PRECISION_FAST if PRECISION_IS_FAST[0] else im.PRECISION_EXACT)
    out["modes"] = modes
    return out


def c4_config(im, torch, gen):
    """C4: 4096^2 RGBA Q16 sRGB->Lab + ContrastStretch 2%x1% — one image (the chain as the one call
    MagickHipBatchImages makes, on fresh data every call: no copy inside the timed region) and the
    whole batch of 512 images through MagickHipBatchImages on this GPU (device-resident, in place)."""
    k = 4096
    reps = 6
    fresh = [random_q16(torch, gen, k, k) for _ in range(reps + 3 + 1)]
    turn = {"i": 0}

    def c4():
        # TransformImageColorspace + ContrastStretchImage as the one call a chain makes
        # (MagickHipBatchImages pairs them the same way): FAST converts and bins in one kernel
        img4 = im.Image(fresh[turn["i"] % len(fresh)])
        turn["i"] += 1
        im.transform_colorspace_contrast_stretch_image(img4, "Lab", 0.02 * k * k, k * k - 0.01 * k * k)
    sec = timed(torch, c4, reps)
    prof = kernel_profile(im, c4, 3)
    del fresh
    torch.cuda.empty_cache()
    frame = float(k) * k * 8.0
    bytes_by_kernel = {"colorspace_histogram": 2.0 * frame, "colorspace": 2.0 * frame, "histogram": frame,
                       "apply_lut": 2.0 * frame, "gray_check": frame}
    kernels = kernel_rooflines(prof, bytes_by_kernel, "c4:")
    kernel_ms = sum(v["avg_ms"] for v in prof.values())
    out = {"workload": "4096x4096 RGBA Q16 sRGB->Lab + ContrastStretch 2%x1% (one image of BASELINE configs[3])",
           "Mpixels_per_s": round(k * k / sec / 1e6, 1), "ms": round(sec * 1e3, 4),
           "kernel_only_ms": round(kernel_ms, 4),
           "operator_frac_of_compulsory_bytes": round(4.0 * frame / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "compulsory_bytes": int(4.0 * frame), "kernels": kernels}
    # the batch itself: 512 distinct resident images (68.7 GB), ONE pass of the chain over each, in place
    try:
        count = 512
        chain = [("colorspace", "Lab"), ("contraststretch", 0.02 * k * k, k * k - 0.01 * k * k)]
        warm = [im.Image(random_q16(torch, gen, k, k)) for _ in range(8)]
        im.batch_images(chain, warm, devices=1, streams_per_device=2)
        del warm
        block = torch.randint(-32768, 32768, (count, k, k, 4), generator=gen, device="cuda",
                              dtype=torch.int16).view(torch.uint16)
        images = [im.Image(block[i]) for i in range(count)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        report = im.batch_images(chain, images, devices=1, streams_per_device=2)
        torch.cuda.synchronize()
        sec = time.perf_counter() - t0
        out["batch"] = {"workload": "batch of 512 independent 4096x4096 RGBA Q16 images, sRGB->Lab + ContrastStretch "
                                    "2%x1%, MagickHipBatchImages on ONE GPU, device-resident, in place, one pass "
                                    "(BASELINE configs[3] is this batch over 8 GPUs)",
                        "images": count, "ms": round(sec * 1e3, 2), "ms_per_image": round(sec * 1e3 / count, 4),
                        "Mpixels_per_s": round(count * float(k) * k / sec / 1e6, 1),
                        "frac_of_compulsory_bytes": round(count * 4.0 * frame / sec / 1e9 / HBM_PEAK_GBS, 4),
                        "workers": report.get("workers") if isinstance(report, dict) else None}
        del images, block
    except Exception as exc:
        out["batch"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    torch.cuda.empty_cache()
    return out


PRECISION_IS_FAST = [True]      # set by main() from --precision


def c5_config(im, torch, gen):
    """C5: 16384^2 RGBA Q16 Dilate Disk:15, then UnsharpMask(0x10+1+0.02)."""
    k = 16384
    src5 = random_q16(torch, gen, k, k)
    img5 = im.Image(src5)
    holder = {}
    frame = float(k) * k * 8.0
    out = {}

    def dilate():
        holder["o"] = im.morphology_image(img5, "Dilate", 1, "Disk:15")
    sec = timed(torch, dilate, 2)
    prof = kernel_profile(im, dilate, 2)
    out["c5_dilate_disk15"] = {
        "workload": "16384x16384 RGBA Q16 MorphologyImage(Dilate, Disk:15) (BASELINE configs[4])",
        "Mpixels_per_s": round(k * k / sec / 1e6, 1), "ms": round(sec * 1e3, 3),
        "kernels": kernel_rooflines(prof, {"morph_rects": 2.0 * frame, "morph_convex": 2.0 * frame,
                                           "morph2d": 2.0 * frame}, "c5:")}
    holder.clear()
    # the same frame as float Quantum (HDRI, the reference's configure default): 4.3 GB in, 4.3 GB out
    try:
        srcf = src5.view(torch.int16).to(torch.float32)
        srcf = torch.where(srcf < 0, srcf + 65536.0, srcf)
        imgf = im.Image(srcf)

        def dilate_float():
            holder["o"] = None
            holder["o"] = im.morphology_image(imgf, "Dilate", 1, "Disk:15")
        sec = timed(torch, dilate_float, 2)
        prof = kernel_profile(im, dilate_float, 2)
        out["c5_dilate_disk15_hdri"] = {
            "workload": "16384x16384 RGBA float Quantum MorphologyImage(Dilate, Disk:15)",
            "Mpixels_per_s": round(k * k / sec / 1e6, 1), "ms": round(sec * 1e3, 3),
            "kernels": kernel_rooflines(prof, {"morph_rects": 4.0 * frame, "morph_convex": 4.0 * frame,
                                               "morph2d": 4.0 * frame}, "c5hdri:")}
        holder.clear()

        # ... and the MAC-bound variant on the float frame: its samples are integers (what a 16-bit file
        # decodes to), so the exact-integer i8 kernel takes it (convolve2d_exact.hip; a frame with one
        # fractional sample falls back to the generic kernel: tools/time_convolve2d_hdri.py)
        def convolve_float():
            holder["o"] = None
            holder["o"] = im.morphology_image(imgf, "Convolve", 1, "Disk:15", scale=(1.0, 1))
        sec = timed(torch, convolve_float, 2)
        prof = kernel_profile(im, convolve_float, 2)
        out["c5_convolve_disk15_hdri"] = {
            "workload": "16384x16384 RGBA float Quantum (integer samples) MorphologyImage(Convolve, Disk:15), "
                        "convolve:scale='!'",
            "Mpixels_per_s": round(k * k / sec / 1e6, 1), "ms": round(sec * 1e3, 3), "tolerance": "bit-identical",
            # (the kernel launched behind it for other frames returns at once on this one: not a roofline row)
            "kernels": kernel_rooflines({k: v for k, v in prof.items() if k not in ("morph2d", "conv2d_tie")},
                                        {"conv2d_exact": 4.0 * frame}, "c5hdri:")}
        holder.clear()
        del imgf, srcf
        torch.cuda.empty_cache()
    except Exception as exc:                                 # (memory: 3 x 4.3 GB beside the Q16 frames)
        out.setdefault("c5_dilate_disk15_hdri", {"error": str(exc)[:200]})

    def convolve():
        holder["o"] = im.morphology_image(img5, "Convolve", 1, "Disk:15", scale=(1.0, 1))
    sec = timed(torch, convolve, 2)
    prof = kernel_profile(im, convolve, 2)
    # the cells of a flat kernel are integer multiples of one unit: exact integer sums on the i8 matrix
    # cores (convolve2d_exact.hip), the same launch and the same bits in both precision modes
    im.set_precision(im.PRECISION_EXACT)
    try:
        sec_exact = timed(torch, convolve, 2)
    finally:
        im.set_precision(im.PRECISION_FAST if PRECISION_IS_FAST[0] else im.PRECISION_EXACT)
    macs = float(k) * k * 4.0 * 709.0
    # executed: 4 byte planes of alpha*p x 31 kernel rows x 2 chunks of 32 slots per 32 outputs
    executed = float(k) * k * 4.0 * 4.0 * 31.0 * 64.0
    out["c5_convolve_disk15"] = {
        "workload": "16384x16384 RGBA Q16 MorphologyImage(Convolve, Disk:15) with convolve:scale='!' — the MAC-bound "
                    "variant of BASELINE configs[4] (SURVEY 8d): 709 active cells per channel and pixel",
        "Mpixels_per_s": round(k * k / sec / 1e6, 1), "ms": round(sec * 1e3, 3), "ms_exact": round(sec_exact * 1e3, 3),
        "tolerance": "bit-identical in both modes (exact integer sums + tie check)",
        "kernels": kernel_rooflines(prof, {"conv2d_exact": 2.0 * frame, "conv2d_mfma": 2.0 * frame,
                                           "morph2d": 2.0 * frame}, "c5:"),
        "alu": {"bound": "mfma_i8", "unit": "TOP/s", "peak": I8_MFMA_PEAK_TOPS, "dtype": "i8",
                "achieved": round(2.0 * executed / sec / 1e12, 1),
                "frac": round(2.0 * executed / sec / 1e12 / I8_MFMA_PEAK_TOPS, 4),
                "algorithmic": round(2.0 * macs / sec / 1e12, 1),
                "useful_frac": round(2.0 * macs / sec / 1e12 / I8_MFMA_PEAK_TOPS, 4),
                "executed_frac": round(2.0 * executed / sec / 1e12 / I8_MFMA_PEAK_TOPS, 4),
                "note": "`frac` / `executed_frac` count EXECUTED work, `useful_frac` the algorithmic multiply-adds: "
                        "achieved = the i8 multiply-adds the banded form executes (four byte planes of alpha*p, 31 "
                        "kernel rows, two 32-slot chunks per 32 outputs: 22x the 709 algorithmic ones per sample, "
                        "which `algorithmic` counts) against the nominal dense i8 peak (bare instructions issue at 4200 TOP/s, "
                        "tools/ubench/mfma_i8_shapes.hip)"}}
    holder.clear()

    def unsharp():
        holder["o"] = im.unsharp_mask_image(img5, 0.0, 10.0, 1.0, 0.02)
    sec = timed(torch, unsharp, 2)
    prof = kernel_profile(im, unsharp, 2)
    # one launch: frame in, frame out (the unblurred pixel the epilogue needs is re-read out of
    # L2 / MALL).  Two-launch form (kernels wider than 81 taps): row pass frame in, frame out;
    # column pass: intermediate + original in, frame out
    out["c5_unsharp"] = {
        "workload": "16384x16384 RGBA Q16 UnsharpMaskImage(0x10+1.0+0.02) (BASELINE configs[4])",
        "Mpixels_per_s": round(k * k / sec / 1e6, 1), "ms": round(sec * 1e3, 3),
        "operator_frac_of_compulsory_bytes": round(2.0 * frame / sec / 1e9 / HBM_PEAK_GBS, 4),
        "kernels": kernel_rooflines(prof, {"unsharp_fused": 2.0 * frame, "unsharp_fused_exact_row": 2.0 * frame,
                                           "unsharp_fused_exact": 2.0 * frame, "conv_row": 2.0 * frame,
                                           "conv_column": 3.0 * frame, "unsharp_epilogue": 3.0 * frame}, "c5:")}
    holder.clear()
    return out


def extra_measurements(im, torch, args, image):
    """What the line reports next to the headline."""
    extra, result = {}, {}
    gen = torch.Generator(device="cuda").manual_seed(1)
    n = image.rows
    try:
        result["modes"] = {p: blur_mode(im, torch, image, args.sigma, p) for p in ("fast", "exact")}
        # float Quantum (the reference's configure default, HDRI): the same 8192^2 RGBA frame as floats
        hdri_pixels = image.pixels.view(torch.int16).to(torch.float32)
        hdri_pixels = torch.where(hdri_pixels < 0, hdri_pixels + 65536.0, hdri_pixels)
        result["modes"]["hdri"] = blur_mode(im, torch, im.Image(hdri_pixels), args.sigma, "hdri", reps=6)
        del hdri_pixels
        torch.cuda.empty_cache()
        im.set_precision(im.PRECISION_FAST if args.precision == "fast" else im.PRECISION_EXACT)
        if args.sustain > 0:
            # sustained rate: back-to-back calls for >= args.sustain seconds (clocks settle against
            # the power limit within the first tens of milliseconds)
            holder = {}
            calls, t0 = 0, time.perf_counter()
            while True:
                for _ in range(100):
                    holder["o"] = im.blur_image(image, 0.0, args.sigma)
                calls += 100
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t0
                if elapsed >= args.sustain:
                    break
            result["sustained"] = {"seconds": round(elapsed, 2), "calls": calls,
                                   "Mpixels_per_s": round(calls * float(n) * n / elapsed / 1e6, 1)}
            holder.clear()
        if args.precision == "fast":
            sec = timed(torch, lambda: im.gaussian_blur_image(image, 0.0, args.sigma), 5)
            extra["gaussian_blur_2d_kernel_Mpixels_per_s"] = round(n * n / sec / 1e6, 1)
            # GaussianBlurImage (a 79 x 79 kernel for sigma 10) bit-identical: two fp64 passes over
            # alpha-premultiplied doubles + tie check (convolve_separable.hip), Q16 and float Quantum
            try:
                im.set_precision(im.PRECISION_EXACT)
                sec = timed(torch, lambda: im.gaussian_blur_image(image, 0.0, args.sigma), 3)
                extra["gaussian_blur_2d_exact_Mpixels_per_s"] = round(n * n / sec / 1e6, 1)
                as_float = image.pixels.view(torch.int16).to(torch.float32)
                as_float = torch.where(as_float < 0, as_float + 65536.0, as_float)
                float_image = im.Image(as_float)
                sec = timed(torch, lambda: im.gaussian_blur_image(float_image, 0.0, args.sigma), 3)
                extra["gaussian_blur_2d_exact_hdri_Mpixels_per_s"] = round(n * n / sec / 1e6, 1)
                del float_image, as_float
            finally:
                im.set_precision(im.PRECISION_FAST)
            torch.cuda.empty_cache()
        # reference point for the roofline: what a plain device copy of the same frame reaches
        mirror = torch.empty_like(image.pixels)
        sec = timed(torch, lambda: mirror.copy_(image.pixels), 10)
        extra["device_copy_GBps"] = round(2.0 * image.pixels.numel() * 2 / sec / 1e9, 1)
        # ... and what a kernel that only WRITES reaches (C3's resize is 94 % stores: 17.2 of 18.25 GB)
        sec = timed(torch, lambda: mirror.fill_(7), 10)
        extra["device_fill_GBps"] = round(image.pixels.numel() * 2 / sec / 1e9, 1)
        del mirror
        # the same call on a host (pixel-cache) buffer: upload + kernels + download, what a single
        # un-chained operator costs through the MagickCore shim
        host = image.pixels.view(torch.int16).cpu().numpy().view("uint16")
        host_image = im.Image(host)
        im.blur_image(host_image, 0.0, args.sigma)
        t0 = time.perf_counter()
        for _ in range(2):
            im.blur_image(host_image, 0.0, args.sigma)
        # (every call faults in a fresh 537 MB result array: the kernel's page faults, not the link)
        extra["blur_host_buffers_new_result_Mpixels_per_s"] = round(2 * n * n / (time.perf_counter() - t0) / 1e6, 1)
        # ... and into a destination that already exists, which is MagickCore's situation: the
        # result image's pixel cache is allocated (and touched) by CloneImage before the operator runs
        host_out = host_image.like()
        host_out.pixels[:] = 0
        im.blur_image(host_image, 0.0, args.sigma, out=host_out)
        t0 = time.perf_counter()
        for _ in range(4):
            im.blur_image(host_image, 0.0, args.sigma, out=host_out)
        extra["blur_host_buffers_Mpixels_per_s"] = round(4 * n * n / (time.perf_counter() - t0) / 1e6, 1)
        # ... and with both pixel caches page-locked (MhHostAlloc: what the shim's allocator hands
        # MagickCore once acceleration is on): one DMA transfer per direction, no staging threads
        pinned_in = im.host_alloc(host.shape, host.dtype)
        pinned_in[:] = host
        pinned_image = im.Image(pinned_in)
        pinned_out = im.Image(im.host_alloc(host.shape, host.dtype))
        pinned_out.pixels[:] = 0
        im.blur_image(pinned_image, 0.0, args.sigma, out=pinned_out)
        t0 = time.perf_counter()
        for _ in range(4):
            im.blur_image(pinned_image, 0.0, args.sigma, out=pinned_out)
        extra["blur_pinned_host_buffers_Mpixels_per_s"] = round(4 * n * n / (time.perf_counter() - t0) / 1e6, 1)
        del pinned_image, pinned_out, pinned_in
        extra["host_link"] = ("PCIe: 57 GB/s either direction, 57 GB/s combined when both run (half duplex on "
                              "this box, tools/pcie_probe.py): 1.07 GB per 8192^2 call = 18.8 ms at best")
        del host_out, host_image
        torch.cuda.empty_cache()
        try:
            extra.update(shim_measurements(n, args.sigma, host))
        except Exception as exc:                          # the shim build is optional on a box
            extra["shim"] = "%s: %s" % (type(exc).__name__, exc)
        del host
        torch.cuda.empty_cache()
        result["resize"] = resize_config(im, torch, gen)
        torch.cuda.empty_cache()
        configs = {"c4_lab_contrast_stretch": c4_config(im, torch, gen)}
        torch.cuda.empty_cache()
        configs.update(c5_config(im, torch, gen))
        torch.cuda.empty_cache()
        torch.cuda.empty_cache()
        configs["c1_blur_1024_sigma2"] = c1_config(im, torch, gen)
        configs["resize_reduce"] = resize_reduce_config(im, torch, gen)
        configs["reference_device_benchmark"] = reference_device_benchmark_config(im, torch, gen)
        torch.cuda.empty_cache()
        configs["narrow_pixel_layouts"] = narrow_layouts_config(im, torch, gen)
        torch.cuda.empty_cache()
        result["configs"] = configs
        result["inputs"] = input_variants(im, torch, gen, args.sigma)
        torch.cuda.empty_cache()
        im.set_precision(im.PRECISION_FAST if args.precision == "fast" else im.PRECISION_EXACT)
    except Exception as exc:
        extra["error"] = "%s: %s" % (type(exc).__name__, exc)
    result["extra"] = extra
    if extra.get("device_copy_GBps"):
        add_measured_ceiling(result, extra["device_copy_GBps"])
    if extra.get("device_fill_GBps") and isinstance(result.get("resize"), dict):
        roof = result["resize"].get("roofline")
        if isinstance(roof, dict) and "achieved" in roof:
            roof["frac_of_measured_fill"] = round(roof["achieved"] / extra["device_fill_GBps"], 4)
    return result


def resize_reduce_config(im, torch, gen):
    """The commonest resize there is, a REDUCTION: 8192^2 -> 2048^2 Lanczos on Q16 and float Quantum RGBA, both
    modes.  Compulsory bytes = the source read + the result written; the work follows the SOURCE pixels
    (every one of them is weighted into the first filter's output), so both pixel rates are given."""
    m, t = 8192, 2048
    out = {"workload": "8192x8192 -> 2048x2048 Lanczos ResizeImage (reduction), RGBA, Q16 and float Quantum"}
    try:
        for label, frame, px_bytes in (("q16", random_q16(torch, gen, m, m), 8.0),
                                       ("hdri", torch.rand((m, m, 4), generator=gen, device="cuda",
                                                           dtype=torch.float32) * 65535.0, 16.0)):
            image = im.Image(frame)
            compulsory = (float(m) * m + float(t) * t) * px_bytes
            node = {}
            for name, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
                im.set_precision(precision)
                sec = timed(torch, lambda: im.resize_image(image, t, t, "Lanczos"), 8)
                prof = kernel_profile(im, lambda: im.resize_image(image, t, t, "Lanczos"), 3)
                kernel_ms = sum(v["avg_ms"] for v in prof.values())
                node[name] = {"ms": round(sec * 1e3, 4),
                              "Mpixels_per_s": round(float(t) * t / sec / 1e6, 1),
                              "source_Mpixels_per_s": round(float(m) * m / sec / 1e6, 1),
                              "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in prof.items()},
                              "roofline": {"bound": "hbm", "achieved": round(compulsory / (kernel_ms * 1e-3) / 1e9, 1),
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(compulsory / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           "algorithmic_bytes": int(compulsory), "avg_ms": round(kernel_ms, 4),
                                           "traffic": None}}
            out[label] = node
            del image, frame
            torch.cuda.empty_cache()
    finally:
        im.set_precision(im.PRECISION_FAST if PRECISION_IS_FAST[0] else im.PRECISION_EXACT)
    return out


def reference_device_benchmark_config(im, torch, gen):
    """The reference's OWN device benchmark (RunOpenCLBenchmark, opencl.c:1047-1110: what it times to rank
    devices): a 2048x1536 RGBA frame through BlurImage(10, 3.5) -> UnsharpMaskImage(2, 2, 50, 10) ->
    ResizeImage(640x480, Lanczos); one untimed round, then timed rounds.  The reference's frame is "xc:none"
    (transparent black); a random frame is timed beside it."""
    w, h = 2048, 1536
    out = {"workload": "2048x1536 RGBA Q16: BlurImage(10,3.5) -> UnsharpMaskImage(2,2,50,10) -> ResizeImage(640x480, Lanczos) "
                       "(RunOpenCLBenchmark, opencl.c:1066-1090)"}
    frames = {"xc_none": torch.zeros((h, w, 4), dtype=torch.int16, device="cuda").view(torch.uint16),
              "random": random_q16(torch, gen, h, w)}
    compulsory = 3 * 2.0 * w * h * 8.0 - 1.0 * w * h * 8.0 + 640.0 * 480.0 * 8.0   # blur r+w, unsharp r+w, resize r + its result
    try:
        for label, frame in frames.items():
            image = im.Image(frame)

            def chain():
                blurred = im.blur_image(image, 10.0, 3.5)
                sharpened = im.unsharp_mask_image(blurred, 2.0, 2.0, 50.0, 10.0)
                return im.resize_image(sharpened, 640, 480, "Lanczos")
            node = {}
            for name, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
                im.set_precision(precision)
                sec = timed(torch, chain, 20)
                prof = kernel_profile(im, chain, 3)
                kernel_ms = sum(v["avg_ms"] for v in prof.values())
                node[name] = {"ms": round(sec * 1e3, 4), "Mpixels_per_s": round(float(w) * h / sec / 1e6, 1),
                              "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in prof.items()},
                              "roofline": {"bound": "hbm", "achieved": round(compulsory / (kernel_ms * 1e-3) / 1e9, 1),
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(compulsory / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           "algorithmic_bytes": int(compulsory), "avg_ms": round(kernel_ms, 4),
                                           "traffic": None}}
            out[label] = node
    finally:
        im.set_precision(im.PRECISION_FAST if PRECISION_IS_FAST[0] else im.PRECISION_EXACT)
    return out


def c1_config(im, torch, gen):
    """C1 (BASELINE configs[0]): 1024^2 RGBA Q16 BlurImage(0x2) — the reference's own CPU-runnable case:
    GPU ms per call in both modes here; the reference's OpenMP time is filled in by the CPU leg."""
    img = im.Image(random_q16(torch, gen, 1024, 1024))
    out = {"workload": "1024x1024 RGBA Q16 BlurImage(0x2) (BASELINE configs[0])"}
    try:
        for name, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
            im.set_precision(precision)
            sec = timed(torch, lambda: im.blur_image(img, 0.0, 2.0), 50)
            out["ms_" + name] = round(sec * 1e3, 4)
            out["Mpixels_per_s_" + name] = round(1024.0 * 1024.0 / sec / 1e6, 1)
    finally:
        im.set_precision(im.PRECISION_FAST if PRECISION_IS_FAST[0] else im.PRECISION_EXACT)
    return out


def narrow_layouts_config(im, torch, gen):
    """Gray (one channel) and RGB (three channels, no alpha) Q16 frames — masks, scans, photographs — have 2- and
    6-byte pixels; the wide-pixel kernels take them as four row bands = four channels / with a fourth, empty channel
    (DESIGN 4.2.1).  8192^2, ms per call in the library's default mode, the frame's own form beside it, and the
    fraction of the 8 TB/s roofline on the compulsory bytes (frame read + frame written)."""
    n = 8192
    out = {"workload": "8192x8192 Q16, one-channel (gray) and three-channel (RGB) frames: Dilate Disk:15, "
                       "BlurImage(0x10), GaussianBlurImage(0x3), sRGB->Lab"}
    try:
        im.set_precision(im.PRECISION_FAST)
        rgba = random_q16(torch, gen, n, n)
        frames = {"gray": (rgba[:, :, :1].contiguous(), 2), "rgb": (rgba[:, :, :3].contiguous(), 6)}
        del rgba
        ops = {"gray": [("dilate_disk15", lambda i: im.morphology_image(i, "Dilate", 1, "Disk:15")),
                        ("blur_0x10", lambda i: im.blur_image(i, 0.0, 10.0)),
                        ("gaussian_blur_0x3", lambda i: im.gaussian_blur_image(i, 0.0, 3.0))],
               "rgb": [("dilate_disk15", lambda i: im.morphology_image(i, "Dilate", 1, "Disk:15"))]}
        for layout, (px, pixel_bytes) in frames.items():
            img = im.Image(px)
            for name, op in ops[layout]:
                entry = {}
                for form, switch in (("ms", None), ("ms_own_form", "1")):
                    im.set_option("MAGICKHIP_NO_GRAY_BANDS", switch)
                    im.set_option("MAGICKHIP_NO_RGB_PAD", switch)
                    op(img)
                    entry[form] = round(timed(torch, lambda: op(img), 5) * 1e3, 4)
                entry["frac_of_8TBps_on_compulsory_bytes"] = round(2.0 * n * n * pixel_bytes / (entry["ms"] * 1e-3) / 8e12, 4)
                out["%s_%s" % (layout, name)] = entry
            del img
        im.set_option("MAGICKHIP_NO_GRAY_BANDS", None)
        im.set_option("MAGICKHIP_NO_RGB_PAD", None)
        px = frames["rgb"][0]
        entry = {}
        for form, switch in (("ms", None), ("ms_own_form", "1")):
            im.set_option("MAGICKHIP_NO_FAST_LAB", switch)
            im.transform_image_colorspace(im.Image(px.clone()), "Lab")
            copies = [px.clone() for _ in range(3)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in copies:
                im.transform_image_colorspace(im.Image(c), "Lab")
            torch.cuda.synchronize()
            entry[form] = round((time.perf_counter() - t0) / 3 * 1e3, 4)
            del copies
        im.set_option("MAGICKHIP_NO_FAST_LAB", None)
        entry["frac_of_8TBps_on_compulsory_bytes"] = round(2.0 * n * n * 6 / (entry["ms"] * 1e-3) / 8e12, 4)
        out["rgb_srgb_to_lab"] = entry
    except Exception as exc:
        out["error"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
    finally:
        for name in ("MAGICKHIP_NO_GRAY_BANDS", "MAGICKHIP_NO_RGB_PAD", "MAGICKHIP_NO_FAST_LAB"):
            im.set_option(name, None)
        im.set_precision(im.PRECISION_FAST if PRECISION_IS_FAST[0] else im.PRECISION_EXACT)
    return out


def input_variants(im, torch, gen, sigma):
    """SURVEY 8(d)'s other input distributions: the BASELINE-size blur and one C4 image on an opaque frame
    (alpha = QuantumRange everywhere: what most real images are) and on a smooth gradient with low noise
    (the worst case for LDS-atomic contention in the histogram), next to the uniform-random headline."""
    n, k = 8192, 4096

    def frame(kind, edge):
        px = random_q16(torch, gen, edge, edge)
        if kind == "opaque":
            px[:, :, 3] = 65535
        elif kind == "smooth":
            y = torch.arange(edge, device="cuda", dtype=torch.float32).view(edge, 1, 1)
            x = torch.arange(edge, device="cuda", dtype=torch.float32).view(1, edge, 1)
            scale = torch.tensor([0.6, 0.7, 0.8, 0.9], device="cuda").view(1, 1, 4)
            noise = torch.randint(0, 400, (edge, edge, 4), generator=gen, device="cuda").to(torch.float32)
            value = ((x * (40000.0 / (edge - 1)) + y * (20000.0 / (edge - 1))) * scale + noise).clamp_(0, 65535)
            px = value.to(torch.int32).to(torch.int16).view(torch.uint16)      # (wraps: the low 16 bits)
            del value, noise
        return px
    out = {}
    try:
        for kind in ("opaque", "smooth"):
            img = im.Image(frame(kind, n))
            entry = {}
            for name, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
                im.set_precision(precision)
                t0 = time.perf_counter()                  # the headline's clock ramp
                while time.perf_counter() - t0 < 0.3:
                    im.blur_image(img, 0.0, sigma)
                    torch.cuda.synchronize()
                sec = timed(torch, lambda: im.blur_image(img, 0.0, sigma), 24)
                entry["blur_ms_" + name] = round(sec * 1e3, 4)
                entry["blur_Mpixels_per_s_" + name] = round(float(n) * n / sec / 1e6, 1)
            del img
            torch.cuda.empty_cache()
            im.set_precision(im.PRECISION_FAST)
            fresh = [frame(kind, k) for _ in range(6)]
            turn = {"i": 0}

            def c4():
                img4 = im.Image(fresh[turn["i"] % len(fresh)])
                turn["i"] += 1
                im.transform_colorspace_contrast_stretch_image(img4, "Lab", 0.02 * k * k, k * k - 0.01 * k * k)
            sec = timed(torch, c4, 4)
            prof = kernel_profile(im, c4, 2)
            entry["c4_ms"] = round(sec * 1e3, 4)
            entry["c4_kernels_ms"] = {name: round(v["avg_ms"], 4) for name, v in prof.items()}
            del fresh
            torch.cuda.empty_cache()
            out[kind] = entry
        out["note"] = ("8192^2 RGBA Q16 BlurImage(0x%g) per mode and one 4096^2 sRGB->Lab + ContrastStretch chain (FAST); "
                       "the headline's frame is uniform random in every channel" % sigma)
    finally:
        im.set_precision(im.PRECISION_FAST if PRECISION_IS_FAST[0] else im.PRECISION_EXACT)
    return out


def shim_measurements(n, sigma, host):
    """The drop-in boundary itself: MagickCore's own operators on the
    HIP-backed MagickCore build (shim/), as an unchanged application calls them.

    shim_blur: BlurImage on the 8192^2 RGBA Q16 frame, source NOT resident (the CPU wrote it), result
    brought back to the host by the cache's lazy sync — upload + kernel + download + CloneImage of
    a page-locked result cache, per call.  shim_blur_resident: the same with the source left on the
    device and the result not read (a link of a chain).  shim_batch: 64 x 4096^2 sRGB -> Lab +
    ContrastStretch from 8 host threads — every call arbitrated over the devices and their streams
    (AcquireHipQueue), results read back."""
    import threading
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "shim"))
    import magickcore as mc
    out = {}
    if not mc.available(False):
        return {"shim": "shim/_build is not built"}
    before = mc.accelerated_calls()
    source = mc.Image(host)
    source.blur(0.0, sigma).sync()                         # warm-up: library load, tables, pool
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        source.touch()
        source.blur(0.0, sigma).sync()
    out["shim_blur_Mpixels_per_s"] = round(reps * float(n) * n / (time.perf_counter() - t0) / 1e6, 1)
    # a link of a device-resident chain: source on the device, results not read.  (One untimed round
    # first: three results alive at once are three page-locked caches, and page-locking 537 MB for
    # the first time costs more than the operator; MhHostAlloc keeps released blocks for reuse.)
    keep = [source.blur(0.0, sigma) for _ in range(reps)]
    keep[-1].sync()
    del keep
    t0 = time.perf_counter()
    keep = [source.blur(0.0, sigma) for _ in range(reps)]
    mc.load().shim_image_sync  # (no read of the results: the chain would go on)
    elapsed_enqueue = time.perf_counter() - t0
    keep[-1].sync()
    elapsed = time.perf_counter() - t0
    out["shim_blur_resident_Mpixels_per_s"] = round(reps * float(n) * n / elapsed / 1e6, 1)
    out["shim_blur_resident"] = {"calls": reps, "ms_per_call_enqueue": round(elapsed_enqueue * 1e3 / reps, 3),
                                 "ms_total_with_one_download": round(elapsed * 1e3, 2)}
    del keep, source
    edge, count, threads = 4096, 64, 8
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 65536, (edge, edge, 4), dtype=np.uint16)
    images = [mc.Image(frame) for _ in range(count)]
    pixels = float(edge) * edge
    errors = []

    def work(t):
        try:
            for image in images[t::threads]:
                image.colorspace("Lab").contrast_stretch(0.02 * pixels, pixels - 0.01 * pixels).sync()
        except Exception as exc:                          # noqa: BLE001
            errors.append(repr(exc))
    t0 = time.perf_counter()
    pool = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    sec = time.perf_counter() - t0
    out["shim_batch_Mpixels_per_s"] = round(count * pixels / sec / 1e6, 1)
    out["shim_batch"] = {"images": count, "edge": edge, "threads": threads, "seconds": round(sec, 3),
                         "devices": [{"calls": c, "streams": s} for c, s in mc.device_statistics()],
                         "errors": errors}
    out["shim_accelerated_calls"] = mc.accelerated_calls() - before
    # ONE big host-resident image (BASELINE configs[4] through the boundary): the row bands of the frame
    # round every device, against the reference's one device per call.  Subprocesses: the shim lists its
    # devices when it starts.  (On a one-GPU box the logical devices share one host link: the row shows
    # the path, not the gain of several links.)
    try:
        import subprocess
        physical = torch_device_count()
        logical = max(2, physical)
        rows = {}
        for label, env_extra in (("one_device_per_call", {"MAGICK_HIP_SPREAD_BYTES": str(1 << 60)}),
                                 ("spread_over_devices", {"MAGICK_HIP_SPREAD_BYTES": str(256 << 20),
                                                          "MAGICKHIP_LOGICAL_DEVICES": str(logical)})):
            env = dict(os.environ, **env_extra)
            done = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shim_spread_bench.py"), "8192"],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            rows[label] = (json.loads(done.stdout.strip().splitlines()[-1]) if done.returncode == 0
                           else {"error": done.stderr[-300:]})
        out["shim_sharded"] = {"workload": "8192x8192 RGBA Q16, source on the host, result read on the host: MagickCore's "
                                           "MorphologyImage(Dilate, Disk:15), BlurImage(0x10), EqualizeImage",
                               "physical_devices": physical, "logical_devices": logical, **rows}
    except Exception as exc:                              # noqa: BLE001
        out["shim_sharded"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return out


def torch_device_count():
    import torch
    return torch.cuda.device_count()


# ------------------------------------------------------------------ N-rank configurations
def shard_range(total, rank, world):
    """Contiguous share of `total` units for `rank` (as imagemagick_amd.distributed.shard_range)."""
    return total * rank // world, total * (rank + 1) // world


def make_step(im, torch, dist, args, rank, world):
    """-> (step function, units per step for the whole job, description, scaling)."""
    gen = torch.Generator(device="cuda").manual_seed(42 + rank)
    if args.config == "c2":
        n = args.size or 8192
        src = random_q16(torch, gen, n, n)
        if os.environ.get("MAGICKHIP_BENCH_ZERO"):     # diagnostics only: data-dependent clocking
            src.zero_()
        image = im.Image(src)
        holder = {}

        def step():
            holder["out"] = im.blur_image(image, 0.0, args.sigma)
        workload = ("%dx%d RGBA Q16 BlurImage(radius=0,sigma=%g): 79-tap row pass + 79-tap column pass, "
                    "Quantum-rounded intermediate, edge clamp, alpha-weighted colour channels; one "
                    "independent image per GPU (BASELINE configs[1])" % (n, n, args.sigma))
        return step, float(n) * n * world, workload, "weak", image
    if args.config == "c4":
        # the batch of 512 independent images, sharded over the ranks: no collective.  A rank keeps a
        # pool of distinct images resident and walks its share of the batch through
        # MagickHipBatchImages (device-resident, in place, worker threads x streams).
        k = args.size or 4096
        batch = 512
        lo, hi = shard_range(batch, rank, world)
        # every image of the rank's share resident and distinct (512 x 134 MB = 68.7 GB on one GPU); a
        # step is one pass of the chain over each, in place (what MagickCore's in-place operators do).
        # Later steps work on the previous step's output relabelled sRGB: synthetic data of the same
        # shape, no refill copies inside the timed region.
        block = torch.randint(-32768, 32768, (hi - lo, k, k, 4), generator=gen, device="cuda",
                              dtype=torch.int16).view(torch.uint16)
        images = [im.Image(block[i]) for i in range(hi - lo)]
        chain = [("colorspace", "Lab"), ("contraststretch", 0.02 * k * k, k * k - 0.01 * k * k)]
        streams = int(os.environ.get("MAGICKHIP_BENCH_C4_STREAMS", "2"))

        def step():
            for image in images:
                image.colorspace = "srgb"
            im.batch_images(chain, images, devices=1, streams_per_device=streams)
        workload = ("batch of %d independent %dx%d RGBA Q16 images, sRGB->Lab + ContrastStretch 2%%x1%%, "
                    "sharded over the ranks, MagickHipBatchImages per rank (BASELINE configs[3])" % (batch, k, k))
        return step, float(batch) * k * k, workload, "strong", None
    n = args.size or 16384
    lo, hi = shard_range(n, rank, world)
    if args.config == "c5":
        # one image, row bands with halo rows: Dilate Disk:15 (reach 15) then UnsharpMask 0x10
        # (reach 39).  Each rank holds its band plus 54 halo rows of the SOURCE and recomputes the
        # dilated halo itself: no exchange, no collective ("upload overlapping bands").
        reach_a, reach_b = 15, 39
        top, bottom = min(lo, reach_a + reach_b), min(n - hi, reach_a + reach_b)
        band = random_q16(torch, gen, hi - lo + top + bottom, n)
        image = im.Image(band)
        holder = {}

        def step():
            dilated = im.morphology_image(image, "Dilate", 1, "Disk:15")
            holder["out"] = im.unsharp_mask_image(dilated, 0.0, 10.0, 1.0, 0.02)
        workload = ("%dx%d RGBA Q16 Dilate Disk:15 + UnsharpMask(0x10+1+0.02), row-sharded over the ranks "
                    "with 54 halo rows (BASELINE configs[4])" % (n, n))
        return step, float(n) * n, workload, "strong", None
    # equalize: local histogram over the owned rows -> ONE all-reduce of the table (RCCL) -> the
    # identical LUT on every rank -> local apply
    band = random_q16(torch, gen, hi - lo, n)
    image = im.Image(band)
    from imagemagick_amd import distributed as D

    def step():
        D.equalize_band(image, None, total_rows=n)
    workload = ("%dx%d RGBA Q16 EqualizeImage, row-sharded over the ranks, one all-reduce of the "
                "65536 x 4 table" % (n, n))
    return step, float(n) * n, workload, "strong", None


def timed_steps(torch, dist, args, world, step, steps, warmup):
    """W warm-up steps, then K steps between barriers; (max over ranks of the elapsed time, per-rank ms)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed / steps * 1e3]
    if world > 1:
        mine = torch.tensor([elapsed], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t.item()) / steps * 1e3 for t in every]
        elapsed = max(float(t.item()) for t in every)
    return elapsed, per_rank


def secondary_distributed(im, torch, dist, args, rank, world):
    """N > 1, default invocation: the other multi-GPU configurations in the same launch, a few steps
    each — `equalize` (one image in row bands, ONE all-reduce of the histogram table per step: the
    only collective of the design), `c5` (row bands with recomputed halos, no collective) and `c4`
    (the 512-image batch sharded over the ranks).  Every rank runs the same steps (collectives)."""
    import copy
    out = {}
    for config, steps in (("equalize", 5), ("c5", 3), ("c4", 2)):
        sub = copy.copy(args)
        # (MAGICKHIP_BENCH_SECONDARY_SIZE: the image edge, for the two-rank rehearsal of the test suite)
        sub.config, sub.size = config, int(os.environ.get("MAGICKHIP_BENCH_SECONDARY_SIZE", "0"))
        step, units, workload, scaling, _ = make_step(im, torch, dist, sub, rank, world)
        elapsed, per_rank = timed_steps(torch, dist, sub, world, step, steps, 1)
        out[config] = {"workload": workload, "scaling": scaling, "steps": steps,
                       "ms_per_step": round(elapsed / steps * 1e3, 4),
                       "Mpixels_per_s": round(units * steps / elapsed / 1e6, 1),
                       "per_rank_ms_per_step": [round(t, 4) for t in per_rank],
                       "collective": ({"backend": args.backend, "used_rccl": args.backend == "nccl",
                                       "what": "one all-reduce of the 65536 x channels histogram table per step"}
                                      if config == "equalize" else {"used_rccl": False, "what": "none"})}
        del step
        torch.cuda.empty_cache()
    return out


def config_dtype(args):
    if args.config == "c2":
        if os.environ.get("MAGICKHIP_NO_MFMA"):
            return "f64" if args.precision == "exact" else "f32"
        return MODE_DTYPE[args.precision]
    if args.config == "c4":
        return ("f32 colour transform (v_log/v_exp), u16 histogram counters, f64 map" if args.precision == "fast"
                else "f64 colour transform, u32/u64 histogram counters, f64 map")
    if args.config == "c5":
        return ("u16 min/max (Dilate) + the blur's arithmetic (UnsharpMask): " + MODE_DTYPE[args.precision])
    return "u64 histogram counters, f64 map (EqualizeImage)"


def config_tolerance(args):
    if args.config == "c2":
        return MODE_TOLERANCE[args.precision]
    if args.config == "c4":
        return ("sRGB->Lab within +-1 level of the reference, ContrastStretch of those levels bit-identical"
                if args.precision == "fast" else "bit-identical to the reference CPU path")
    if args.config == "c5":
        return "bit-identical to the reference CPU path (both modes: FAST UnsharpMask runs the exact launch)"
    return "bit-identical to the reference CPU path"


def main():
    args = parse_args()
    if os.environ.get("MAGICKHIP_BENCH_WATCHDOG"):      # diagnostics: dump every thread's stack and exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["MAGICKHIP_BENCH_WATCHDOG"]), exit=True)
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
    PRECISION_IS_FAST[0] = args.precision == "fast"
    im.set_precision(im.PRECISION_FAST if args.precision == "fast" else im.PRECISION_EXACT)

    step, units, workload, scaling, image = make_step(im, torch, dist, args, rank, world)

    # clocks: the chip is idle while the inputs are generated; a few hundred ms of the step bring
    # it to its sustained state before the W warm-up steps the contract counts (MAGICKHIP_BENCH_RAMP=0
    # skips it)
    ramp = float(os.environ.get("MAGICKHIP_BENCH_RAMP", "0.3"))
    if distributed and ramp > 0:
        # every rank must run the SAME number of steps (a step may hold a collective): rank 0
        # times one step and broadcasts the count
        t_ramp = time.perf_counter()
        step()
        torch.cuda.synchronize()
        one = max(time.perf_counter() - t_ramp, 1e-4)
        count = torch.tensor([min(int(ramp / one) + 1, 10000)], dtype=torch.int64,
                             device="cuda" if args.backend == "nccl" else "cpu")
        dist.broadcast(count, src=0)
        for _ in range(int(count.item())):
            step()
        torch.cuda.synchronize()
    else:
        t_ramp = time.perf_counter()
        while ramp > 0 and time.perf_counter() - t_ramp < ramp:
            step()
            torch.cuda.synchronize()
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
    per_rank_ms = [elapsed / args.steps * 1e3]
    if distributed:
        mine = torch.tensor([elapsed], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        elapsed = max(float(t.item()) for t in every)          # the job is as slow as its slowest rank

    # N > 1: the configurations with a collective / with row bands, in the same launch.  A watchdog
    # keeps the headline safe: if they do not finish in time rank 0 prints the line without them and
    # every rank leaves (a hung collective must not cost the measurement above).
    secondary = None
    if distributed and (args.config == "c2") and not args.no_secondary:
        import threading
        done = threading.Event()
        state = {}

        def bail():
            if done.is_set():
                return
            if rank == 0 and "line" in state:
                state["line"]["multi_gpu"] = {"error": "the secondary configurations did not finish in 150 s"}
                print(json.dumps(state["line"]), flush=True)
            os._exit(0 if rank == 0 else 1)
        watchdog = threading.Timer(150.0, bail)
        watchdog.daemon = True
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = units * args.steps / elapsed / 1e6
        result = {
            "metric": METRIC if args.config == "c2" else "Mpixels/sec, config " + args.config,
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None,
            "dtype": config_dtype(args),
            "data": "synthetic",
            "tolerance": config_tolerance(args),
            "per_rank_ms_per_step": [round(t, 4) for t in per_rank_ms],
            "collective": ({"backend": args.backend, "used_rccl": args.backend == "nccl",
                            "what": "one all-reduce of the 65536 x channels histogram table per step"}
                           if (distributed and args.config == "equalize") else
                           {"backend": args.backend if distributed else None, "used_rccl": False,
                            "what": "none: independent images / row bands with recomputed halos"}),
            "config": {"workload": workload, "precision": args.precision, "images_per_step": world
                       if args.config == "c2" else None, "config": args.config, "clock_ramp_seconds": ramp},
        }
        if args.config == "c2":
            n = image.rows
            frame = float(n) * n * 8.0
            conv = {k: v for k, v in prof.items() if k.startswith("conv_") or k.startswith("blur_fused")}
            if conv:
                dominant = max(conv, key=lambda k: conv[k]["avg_ms"])
                # algorithmic bytes of one launch: the frame read once and written once (the fused
                # kernel is the whole operator: this IS BASELINE's compulsory 1.074 GB)
                roof = roofline(dominant, 2.0 * frame, conv[dominant]["avg_ms"], traffic_key(args.precision, dominant))
                roof["kernels_ms"] = {k: round(v["avg_ms"], 4) for k, v in prof.items()}
                ntaps = im.optimal_kernel_width_1d(0.0, args.sigma)
                if dominant.startswith("blur_fused_exact"):
                    roof["alu"] = i8_roofline(float(n) * n, ntaps, conv[dominant]["avg_ms"], dominant)
                else:
                    # the arithmetic beside the stream: f16 matrix cores in the legacy FAST path, vector ALU otherwise
                    passes = 2.0 if dominant.startswith("blur_fused") else 1.0
                    flops = passes * float(n) * n * 4 * ntaps * 2.0
                    mfma = args.precision == "fast" and os.environ.get("MAGICKHIP_NO_MFMA") is None
                    peak = 2500.0 if mfma else (157.3 if args.precision == "fast" else 78.6)
                    tflops = flops / (conv[dominant]["avg_ms"] * 1e-3) / 1e12
                    roof["alu"] = {"achieved_tflops": round(tflops, 2), "peak_tflops": peak,
                                   "frac": round(tflops / peak, 4),
                                   "unit": "f16 MFMA dense" if mfma else
                                           ("f32 vector" if args.precision == "fast" else "f64 vector"),
                                   "note": "algorithmic multiply-adds only"}
                if world == 1 and not args.no_live_traffic and dominant.startswith("blur_fused") and n == 8192:
                    # counters of THIS run in place of the table kept under profiles/
                    live = live_traffic(args.precision, args.sigma)
                    roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes in this run" if live
                                              else "profiles/pmc_traffic.json (rocprofv3 did not run here)")
                    if live:
                        roof["traffic"] = live
                else:
                    roof["traffic_source"] = "profiles/pmc_traffic.json"
                if dominant.startswith("blur_fused") and n == 8192:
                    issue = issue_roofline(args.precision, conv[dominant]["avg_ms"])
                    if issue:
                        # bound / achieved / peak / frac are the HBM roofline the metric asks for (traffic 1.03x the
                        # compulsory bytes: no wasted re-reads); what keeps the kernel from it is on-chip — the
                        # counters' view beside it
                        roof["hbm_frac"] = roof["frac"]
                        roof["compute_roofline"] = issue
                result["roofline"] = roof
            if not args.no_extra and world == 1:
                result.update(extra_measurements(im, torch, args, image))
                # BASELINE's metric names two operators; `value` is the one the configuration it is
                # quoted on (configs[1], the blur) measures — the other half beside it, and the
                # bit-identical mode's rate
                components = {"blur_Mpixels_per_s": result["value"]}
                if "resize" in result:
                    components["resize_Mpixels_per_s"] = result["resize"].get("Mpixels_per_s")
                    components["resize_operator_frac_of_compulsory_bytes"] = \
                        result["resize"].get("operator_frac_of_compulsory_bytes")
                result["value_components"] = components
                if "modes" in result and "exact" in result["modes"]:
                    result["value_exact"] = result["modes"]["exact"]["Mpixels_per_s"]
            result["default_mode"] = ("fast: what an unchanged MagickCore caller gets since round 5 (the library's default; "
                                      "MAGICK_HIP_PRECISION=exact selects the bit-identical mode, `value_exact`)")
            if world == 1 and not args.no_cpu_baseline:
                try:
                    result["cpu_baseline"] = cpu_baseline_blur(args.sigma, im=im, torch=torch)
                    shares = result["cpu_baseline"].pop("identical_share", None) if result["cpu_baseline"] else None
                    if shares:
                        result["identical_share"] = shares
                        for name, share in shares.items():
                            if name in result.get("modes", {}):
                                result["modes"][name]["identical_share"] = share
                    if not args.no_extra:
                        for name, entry in cpu_baseline_configs().items():
                            if name == "c3_resize" and "resize" in result:
                                result["resize"]["cpu_baseline"] = entry
                            elif name in result.get("configs", {}):
                                result["configs"][name]["cpu_baseline"] = entry
                            elif name == "error":
                                result.setdefault("extra", {})["cpu_baseline_error"] = entry
                except Exception as exc:  # the baseline is a report, never a reason to lose the line
                    result["cpu_baseline"] = {"error": str(exc)}
        else:
            result["kernels_ms"] = {k: round(v["avg_ms"], 4) for k, v in prof.items()}
    if distributed and (args.config == "c2") and not args.no_secondary:
        if rank == 0:
            state["line"] = result
        watchdog.start()
        try:
            del step, image
            torch.cuda.empty_cache()
            secondary = secondary_distributed(im, torch, dist, args, rank, world)
        except Exception as exc:          # (a failure on every rank alike; a hang is the watchdog's)
            secondary = {"error": "%s: %s" % (type(exc).__name__, exc)}
        done.set()
        watchdog.cancel()
        if rank == 0:
            result["multi_gpu"] = secondary
            # the one collective of the design, where the driver's scaling table looks for it
            if isinstance(secondary, dict) and "equalize" in secondary:
                result["collective"] = dict(secondary["equalize"]["collective"],
                                            equalize_ms_per_step=secondary["equalize"]["ms_per_step"])
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
