#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the VarDCT decode back-end on MI355X.

One "step" = one pass of the hot path (dequant -> inverse transforms ->
[Gaborish] -> [EPF] -> XYB->linear RGB) over one synthetic frame whose quantized
coefficients and side info are already resident in HBM; output stays in HBM.

Workloads = BASELINE.json's configs, `--config`:
  c1  1024x1024    d1.0 mix, Gaborish+EPF1, int16          (configs[0]; cpu_baseline: 1 thread)
  c2  3840x2160    d1.0 mix, filters off, int16             (configs[1])
  c3  7680x4320    d1.0 mix, Gaborish+EPF1, int16           (configs[2]; DEFAULT at --gpus 1)
  c4  15360x8640   as c3, group-row stripes over the ranks, gather on rank 0 INSIDE the step
                                                            (configs[3]; DEFAULT at --gpus N>1)
  c5  7680x4320    all DCT32x32, int32 coefficients, HDR intensity target 1000, d0.5-like quant,
                   filters off                               (configs[4])
N>1 is strong scaling of c4: the 16K frame is fixed, every rank decodes its stripe (halo rows exchanged with the
two neighbours between the phases) and rank 0 collects the stripes.  `--config c4 --gpus 1` is the same frame on
one GPU (the base of the scaling curve).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

CONFIGS = {
    "c1": dict(width=1024, height=1024, gab=1, epf=1, mix="d1", coeff32=False, intensity=255.0, quant_mul=1.0),
    "c2": dict(width=3840, height=2160, gab=0, epf=0, mix="d1", coeff32=False, intensity=255.0, quant_mul=1.0),
    "c3": dict(width=7680, height=4320, gab=1, epf=1, mix="d1", coeff32=False, intensity=255.0, quant_mul=1.0),
    "c4": dict(width=15360, height=8640, gab=1, epf=1, mix="d1", coeff32=False, intensity=255.0, quant_mul=1.0),
    "c5": dict(width=7680, height=4320, gab=0, epf=0, mix="dct32", coeff32=True, intensity=1000.0, quant_mul=2.0),
}
METRIC = {"c1": "Mpixels/s decode (VarDCT d1.0, 1024x1024 RGB)", "c2": "Mpixels/s decode (VarDCT d1.0, 4K RGB, filters off)",
          "c3": "Mpixels/s decode (VarDCT d1.0, 8K RGB)", "c4": "Mpixels/s decode (VarDCT d1.0, 16K RGB)",
          "c5": "Mpixels/s decode (VarDCT d0.5, 8K HDR, DCT32x32)"}


# Instruction-issue roofline (round 6): the SIMDs of an MI355X retire one wave64 VALU instruction of
# the decoder's kind of code -- packed fp32, DPP neighbours, conversions, three-operand fma -- per ~4.15 shader cycles with
# four waves per SIMD (tools/probes/valu_issue.hip, profiles/r05_valu_issue_probe.txt: "mix: pk_fma,pk_add,add_dpp,fma",
# 4 waves/SIMD, 4.15 cyc/instr/SIMD at the 2.4 GHz event clock; plain v_fma_f32 alone reaches 2.39).  1024 SIMDs.
VALU_ISSUE_PEAK_GINSTR = 1024 * 2.4 / 4.15   # 592 G wave-instructions/s: the probe's mix
VALU_ISSUE_PEAK_FMA_GINSTR = 1024 * 2.4 / 2.39  # 1028: plain v_fma_f32, for scale


class ClockSampler:
    """Device clocks seen WHILE the timed steps run (a thread polling sysfs every ~0.3 ms): the pool's boxes differ by up
    to 30 % on the same code (profiles/r05_box_spread.txt), and a reader of the line must be able to tell a slow box from
    a regression.  sclk / mclk in MHz from the amdgpu hwmon nodes (freq1_input / freq2_input, Hz) or, failing that, the
    starred level of pp_dpm_sclk / pp_dpm_mclk; None when the box exposes neither."""

    def __init__(self, device_index=0):
        import glob
        self.paths = {}
        self.card = None
        # the DRM node of THIS HIP device: by PCI address (a container may see the sysfs nodes of every GPU of the host and
        # only one of them as a HIP device); failing that, every amdgpu card is polled and the busiest one reported
        roots = []
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            if os.path.isdir("/sys/bus/pci/devices/" + addr):
                roots = ["/sys/bus/pci/devices/" + addr]
                self.card = addr
        except Exception:
            roots = []
        if not roots:
            roots = [c for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
                     if os.path.exists(os.path.join(c, "pp_dpm_sclk")) or glob.glob(os.path.join(c, "hwmon/hwmon*/freq1_input"))]
        for n, c in enumerate(roots):
            for key, hw, dpm in (("sclk", "freq1_input", "pp_dpm_sclk"), ("mclk", "freq2_input", "pp_dpm_mclk")):
                h = glob.glob(os.path.join(c, "hwmon/hwmon*/" + hw))
                name = key if len(roots) == 1 else f"{key}@{os.path.basename(os.path.dirname(c)) if c.endswith('/device') else os.path.basename(c)}"
                if h:
                    self.paths[name] = ("hz", h[0])
                elif os.path.exists(os.path.join(c, dpm)):
                    self.paths[name] = ("dpm", os.path.join(c, dpm))
        self.samples = {k: [] for k in self.paths}
        self._stop = None
        self._thread = None

    def _read(self, kind, path):
        try:
            txt = open(path).read()
            if kind == "hz":
                return float(txt.strip()) / 1e6
            for line in txt.splitlines():
                if line.rstrip().endswith("*"):
                    return float("".join(ch for ch in line.split(":")[1] if ch.isdigit() or ch == "."))
        except (OSError, ValueError, IndexError):
            pass
        return None

    def start(self):
        import threading
        if not self.paths:
            return
        self.samples = {k: [] for k in self.paths}
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                for k, (kind, path) in self.paths.items():
                    v = self._read(kind, path)
                    if v is not None:
                        self.samples[k].append(v)
                self._stop.wait(0.0003)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join()
        self._thread = None
        out = {}
        for k, v in self.samples.items():
            if v:
                w = sorted(v)
                out[k + "_mhz"] = {"min": round(w[0]), "median": round(w[len(w) // 2]), "max": round(w[-1]), "samples": len(w)}
        if self.card is None and len(out) > 2:
            # no PCI match: the card whose shader clock ran highest is the one that worked
            best = max((k for k in out if k.startswith("sclk@")), key=lambda k: out[k]["median"], default=None)
            if best:
                tag = best[len("sclk"):-len("_mhz")]
                out = {"sclk_mhz": out[best], **({"mclk_mhz": out["mclk" + tag + "_mhz"]} if "mclk" + tag + "_mhz" in out else {}),
                       "card": tag[1:], "picked": "highest median shader clock of %d amdgpu cards" % (len(out) // 2)}
        elif self.card:
            out["pci"] = self.card
        return out or None


def algorithmic_bytes(xsize, ysize, coeff_bytes):
    """B = Wp*Hp*3*sizeof(coef) + Nblk*18 + W*H*12."""
    wp, hp = (xsize + 7) // 8 * 8, (ysize + 7) // 8 * 8
    return wp * hp * 3 * coeff_bytes + (wp * hp // 64) * 18 + xsize * ysize * 12


def resolve_mix(name):
    from libjxl_amd import synth
    named = {"d1": synth.MIX_D1, "dct8": synth.MIX_DCT8, "dct32": synth.MIX_DCT32, "all": synth.MIX_ALL}
    if name in named:
        return named[name]
    if name == "real4k":
        # area shares of the strategies in a genuine libjxl d1.0 stream (tests/data/real_4k_d1.npz, written by
        # the reference encoder: 42 % 64x64, 32 % 32x32, 7 % DCT8 ...)
        import numpy as np
        acs = np.load(os.path.join(ROOT, "tests", "data", "real_4k_d1.npz"))["ac_strategy"]
        n = np.bincount(acs.ravel() >> 1, minlength=27)
        return {int(s): float(v) for s, v in enumerate(n) if v}
    return {int(k): float(v) for k, v in (kv.split(":") for kv in name.split(","))}


def cpu_baseline(cfg, sample, threads):
    """libjxl's own CPU path timed on the host cores on a bounded sample of the
    same workload: oracle/_ref = the reference decoder sources compiled in place
    (DecodeGroupForRoundtrip + LowMemoryRenderPipeline, the executor djxl uses),
    threaded over groups.  NOTE the build: Highway is not vendored in the
    reference tree, so the SIMD layer is oracle/hwy_shim (ONE lane per vector,
    scalar code) -- this is libjxl's algorithm and code, not its AVX2/AVX-512
    speed.  Falls back to the C restatement (kind "port") without oracle/_ref."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frames
    import oracle
    w, h = sample
    cores = threads or (os.cpu_count() or 1)
    _, _, fr = frames.make_case(w, h, mix=resolve_mix(cfg["mix"]), gab=bool(cfg["gab"]), epf_iters=cfg["epf"],
                                coeff_type=int(cfg["coeff32"]), intensity_target=cfg["intensity"],
                                quant_mul=cfg["quant_mul"])
    use_ref = oracle.ref_available()
    v8 = use_ref and oracle.ref_lib_v8() is not None    # the hot path on 8 float lanes (oracle/hwy_shim_v)
    fma = use_ref and not v8 and oracle.ref_lib_fma() is not None
    kw = dict(v8_build=True) if v8 else dict(fma_build=True)
    run = (lambda: fr.decode_ref(threads=cores, **kw)) if use_ref else (lambda: fr.decode(threads=cores))
    run()  # warm
    # A cgroup CPU quota below the thread count (the GPU boxes of this pool: 16 CPUs under a 256-thread host) stops the
    # whole process for tens of ms once a 100 ms period's slice is used up: every repetition is followed by a pause that
    # pays its CPU time back, so that the clock only sees unthrottled decodes.
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass

    def paced(fn):
        c0, t0 = time.process_time(), time.perf_counter()
        fn()
        if quota:
            owed = (time.process_time() - c0) / quota * 1.25 - (time.perf_counter() - t0)
            if owed > 0:
                time.sleep(min(owed, 0.5))

    if threads == 0 and use_ref and cores > 16:
        # the reference's group-parallel decode does not scale to every hardware thread of a 256-thread host: time
        # the thread count that is fastest on THIS host and say which
        best = (0.0, cores)
        if quota:  # (more runnable threads than four times the CPUs the cgroup may use only queue behind each other)
            cands = sorted({int(min(cores, 4 * quota)), int(min(cores, 2 * quota)), int(min(cores, max(1, quota)))})
        else:
            cands = sorted({cores, max(16, cores // 2), max(16, cores // 4), max(16, cores // 8)})
        for thr in cands:
            paced(lambda: fr.decode_ref(threads=thr, **kw))
            sec = []
            for _ in range(5):
                paced(lambda: fr.decode_ref(threads=thr, **kw))
                sec.append(fr.last_decode_seconds())
            rate = 1.0 / sorted(sec)[2]  # the median of five: the host is shared (other tenants, the GPU runtime's own threads)
            if rate > best[0]:
                best = (rate, thr)
        cores = best[1]
        run = lambda: fr.decode_ref(threads=cores, **kw)
    # timed: the section libjxl's decoder runs per frame on its thread pool -- every AC group through
    # DecodeGroupForRoundtrip and the render pipeline (oracle/ref_driver.cc reports it: jxr_last_decode_seconds).  The
    # driver's own serial set-up in front of it (widening the coefficient buffers into an ACImage, filling
    # PassesSharedState from dense arrays) is not libjxl's decode and is left out, like the GPU side is timed with its
    # inputs resident; `wall` below includes it.
    reps, t, wall, best_rep, secs = 0, 0.0, 0.0, 1e30, []
    t_begin = time.perf_counter()
    while reps < 2 or (time.perf_counter() - t_begin < 10.0 and reps < 40):
        box = {}

        def one():
            t0 = time.perf_counter()
            run()
            box["wall"] = time.perf_counter() - t0

        paced(one)
        wall += box["wall"]
        sec = fr.last_decode_seconds() if use_ref else box["wall"]
        secs.append(sec)
        best_rep = min(best_rep, sec)
        reps += 1
    # the median repetition (a repetition that still ran into the quota, or into another tenant of the host, is an outlier
    # of tens of ms among repetitions of a few)
    t = sorted(secs)[len(secs) // 2] * reps
    if v8:
        what = ("libjxl reference sources (lib/jxl, DecodeGroupForRoundtrip + LowMemoryRenderPipeline); the decode hot path "
                "(dec_group.cc with the inverse transforms, the Gaborish / EPF / XYB / write stages) compiled against an 8-lane "
                "(256-bit, HWY_TARGET = HWY_AVX2) stand-in for Highway, oracle/hwy_shim_v, -O3 -mavx2 -mfma: libjxl's SIMD "
                "code paths, bit-identical to the single-lane checker build here; everything else single-lane.  Not "
                "Google Highway itself (un-vendored in the reference tree): reference, 8-lane shim")
    elif use_ref:
        what = ("libjxl reference sources (lib/jxl, DecodeGroupForRoundtrip + LowMemoryRenderPipeline) built with "
                "the single-lane Highway shim" + (", -O3 -mavx2 -mfma" if fma else " (-O2)") +
                " (the 8-lane build needs AVX2 + FMA on the host)")
    else:
        what = "oracle/ C restatement (libjxl reference library not available)"
    return {"value": round(w * h * reps / t / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "cgroup_cpu_quota": quota,
            "kind": "reference" if use_ref else "port", "simd_lanes": 8 if v8 else 1,
            "best_rep": round(w * h / best_rep / 1e6, 2),
            "value_with_driver_setup": round(w * h * reps / wall / 1e6, 2),
            "sample": f"{w}x{h} frame of this workload, median of {reps} reps (paced under the cgroup CPU quota when there is one), "
                      f"{cores} thread(s) over groups; {what}"}


def pcie_inclusive(torch, dec, params, t, dq, out, xs, ys, n):
    """Host-boundary rates of the bench frame (never `value`)."""
    from libjxl_amd import VarDctDecoder
    host_c = [torch.empty(c.shape, dtype=c.dtype, pin_memory=True).copy_(c) for c in t["coeffs"]]
    h2d = int(sum(c.numel() * c.element_size() for c in host_c))
    res = {"unit": "Mpixels/s", "h2d_bytes": h2d}

    # serial on one stream, f32
    host_o = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)

    def pstep():
        for d, h in zip(t["coeffs"], host_c):
            d.copy_(h, non_blocking=True)
        dec.decode_frame(out)
        host_o.copy_(out, non_blocking=True)

    pstep()
    torch.cuda.synchronize()
    p0 = time.perf_counter()
    for _ in range(n):
        pstep()
    torch.cuda.synchronize()
    pdt = (time.perf_counter() - p0) / n
    res["f32_serial"] = {"value": round(xs * ys / pdt / 1e6, 1), "ms_per_step": round(pdt * 1e3, 3),
                         "d2h_bytes": int(host_o.numel() * host_o.element_size())}
    del host_o

    def pipelined(make_params):
        s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s_k):
            d2 = VarDctDecoder(dec.device)  # launches go to s_k
        d2.begin_frame(make_params)
        sets = []
        for _ in range(2):
            tt = dict(t)
            tt["coeffs"] = [torch.empty_like(c) for c in t["coeffs"]]
            o = d2.alloc_output()
            sets.append(dict(t=tt, out=o, host=torch.empty(o.shape, dtype=o.dtype, pin_memory=True),
                             ev_in=torch.cuda.Event(), ev_k=torch.cuda.Event(), ev_out=torch.cuda.Event()))
        torch.cuda.synchronize()

        def frame(k):
            b = sets[k & 1]
            with torch.cuda.stream(s_in):
                s_in.wait_event(b["ev_k"])  # the kernels of frame k-2 no longer read this set's coefficients
                for d, h in zip(b["t"]["coeffs"], host_c):
                    d.copy_(h, non_blocking=True)
                b["ev_in"].record(s_in)
            with torch.cuda.stream(s_k):
                s_k.wait_event(b["ev_in"])
                s_k.wait_event(b["ev_out"])  # frame k-2's pixels have left this set's device frame
                d2.set_inputs(b["t"], dq)
                d2.decode_frame(b["out"])
                b["ev_k"].record(s_k)
            with torch.cuda.stream(s_out):
                s_out.wait_event(b["ev_k"])
                b["host"].copy_(b["out"], non_blocking=True)
                b["ev_out"].record(s_out)

        for k in range(2):
            frame(k)
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        for k in range(n):
            frame(k)
        torch.cuda.synchronize()
        qdt = (time.perf_counter() - q0) / n
        d2h = int(sets[0]["host"].numel() * sets[0]["host"].element_size())
        d2.sync()
        d2.close()
        return {"value": round(xs * ys / qdt / 1e6, 1), "ms_per_step": round(qdt * 1e3, 3), "d2h_bytes": d2h,
                "h2d_GBps": round(h2d / qdt / 1e9, 1), "d2h_GBps": round(d2h / qdt / 1e9, 1)}

    res["f32_pipelined"] = pipelined(dict(params))
    p8 = dict(params)
    p8["output_kind"] = 2
    p8["out_format"] = dict(transfer=1, sample_type=1, num_channels=4, bits_per_sample=8)  # sRGB RGBA8
    res["rgba8_pipelined"] = pipelined(p8)
    res["value"] = res["rgba8_pipelined"]["value"]
    res["what"] = ("pinned H2D of the coefficient stream + kernels + pinned D2H of the pixels.  value = sRGB RGBA8 "
                   "output, pipelined over three streams and two buffer sets (H2D of frame k+1 || kernels k || D2H k-1); "
                   "f32_pipelined = the same with the linear f32 frame; f32_serial = one stream, no overlap")
    return res


def e2e_block(torch, local):
    """Whole-file rates at the boundary north_star names ("drops in behind djxl"), never `value`: one genuine 8K d1.0 RGB
    codestream written by the reference encoder (oracle/make_e2e_stream.py; effort 7, libjxl's default) --
      codestream_8k_rgb : jxlhip_decode_codestream, bytes -> float RGB in HBM, DC groups and every header included, at
                          the best of a few worker counts, with the per-phase milliseconds of that call;
      djxl_hip / djxl_ref: the reference's own tool, unmodified, on the HIP back-end (libjxl_dec_hip.so +
                          libjxl_threads_hip.so) and on libjxl's CPU decoder (the single-lane Highway build of
                          oracle/_ref), `--num_reps 10 --disable_output`, MP/s as SpeedStats prints it
                          (tools/djxl_main.cc:392-426, tools/speed_stats.cc:102-121)."""
    import ctypes as C
    import re
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_e2e_stream
    from libjxl_amd import VarDctDecoder, abi
    # (made by __graft_entry__.build() in the build container and carried with oracle/_ref; a bench run only makes it
    # when asked to: the reference encoder at effort 7 needs ~3 minutes of one core for an 8K frame)
    path = make_e2e_stream.ensure(generate=os.environ.get("JXLHIP_E2E_GENERATE") == "1")
    if path is None:
        return {"error": "tests/data/e2e_8k_d1.jxl not present (python oracle/make_e2e_stream.py, or JXLHIP_E2E_GENERATE=1)"}
    blob = open(path, "rb").read()
    L = abi.load_library()
    info = abi.CodestreamInfo()
    if L.jxlhip_codestream_basic_info(blob, len(blob), C.byref(info)):
        return {"error": "e2e stream rejected"}
    w, h = info.xsize, info.ysize
    res = {"stream": f"{os.path.relpath(path, ROOT)}: {w}x{h} RGB VarDCT d1.0 effort 7, {len(blob)} bytes, written by the "
                     "reference encoder from a procedural image", "unit": "Mpixels/s"}
    R = C.CDLL(abi.runner_library_path())
    R.JxlThreadParallelRunnerCreate.restype = C.c_void_p
    R.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
    R.JxlThreadParallelRunnerDestroy.argtypes = [C.c_void_p]
    runner = C.cast(R.JxlThreadParallelRunner, C.c_void_p)
    ncpu = os.cpu_count() or 1
    names = ["headers", "dc_groups", "ac_global", "side_info", "ac_groups", "extra_channels", "kernels_sync"]
    dec = VarDctDecoder(local)
    out = torch.empty((h, w, 3), dtype=torch.float32, device=f"cuda:{local}")
    # The host's share of the box: a cgroup CPU quota (cpu.max) throttles the whole process group once it has used its
    # slice of a 100 ms period -- every thread stops for tens of ms.  Back-to-back repetitions on 64 threads run into it
    # (profiles/r04_e2e_cgroup.txt: 16 CPUs on the GPU boxes of this pool; a file costs ~0.2 s of CPU time), so the
    # repetitions are spaced to stay under the quota (20 ms: warm threads, ~45 % of the slice): `value` is the latency
    # of ONE file; `sustained_at_cpu_quota` is what the quota lets through, from the CPU seconds a file costs.
    quota_cpus = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota_cpus = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    gap = 0.02 if quota_cpus else 0.0
    best = None
    for threads in sorted({t for t in (16, 32, 64, 128) if t <= ncpu}):
        pool = R.JxlThreadParallelRunnerCreate(None, threads)
        ts, phases, cpu = [], [], []
        for rep in range(3 + 7):
            time.sleep(gap)
            c0, t0 = time.process_time(), time.perf_counter()
            rc = L.jxlhip_decode_codestream(dec.ctx, runner, pool, blob, len(blob), 1, None, out.data_ptr(), w * 12, 0,
                                            C.byref(info))
            dt, dc = time.perf_counter() - t0, time.process_time() - c0
            if rc:
                R.JxlThreadParallelRunnerDestroy(pool)
                return {"error": f"jxlhip_decode_codestream: {L.jxlhip_last_error(dec.ctx).decode()}"}
            ms = (C.c_double * 7)()
            L.jxlhip_codestream_phase_ms(dec.ctx, ms)
            if rep >= 3:  # the first repetitions allocate (pinned staging, per-thread scratch) and fault their pages in
                ts.append(dt)
                cpu.append(dc)
                phases.append(list(ms))
        R.JxlThreadParallelRunnerDestroy(pool)
        i = sorted(range(len(ts)), key=ts.__getitem__)[len(ts) // 2]
        geo = float(torch.tensor(ts).log().mean().exp())
        if best is None or geo < best["geo"]:
            best = {"geo": geo, "threads": threads, "median": ts[i], "phases": phases[i], "all": list(ts), "cpu": sorted(cpu)[len(cpu) // 2]}
    res["codestream_8k_rgb"] = {
        "value": round(w * h / best["geo"] / 1e6, 1), "median_rep": round(w * h / best["median"] / 1e6, 1), "threads": best["threads"],
        "ms_per_file": round(best["geo"] * 1e3, 2), "reps": 7, "ms_of_each_rep": [round(t * 1e3, 2) for t in best["all"]],
        "phase_ms_of_median_rep": {n: round(v, 2) for n, v in zip(names, best["phases"])},
        "cpu_ms_per_file": round(best["cpu"] * 1e3, 1), "cpu_quota_cpus": quota_cpus,
        "sustained_at_cpu_quota": round(w * h / (best["cpu"] / quota_cpus) / 1e6, 1) if quota_cpus else None,
        "what": "jxlhip_decode_codestream: bytes -> linear f32 RGB in HBM, whole file; geomean of 7 repetitions after 3 warm-up ones, "
                "20 ms apart to stay under the cgroup CPU quota (latency of one file).  dc_groups = until the last DC group ended; ac_groups = what "
                "of the single runner call came after that (the AC groups start as their DC group's block info is in)"}
    # ... and the same file on several contexts at once (a server decoding a queue of files: one context, one HIP stream
    # and one runner pool per file in flight, one host thread each -- the C call releases the GIL): the serial phases of
    # one file (its twelve Modular DC groups: 7-9 ms on one core each) overlap the parallel phases of the others
    try:
        import threading
        files = 4
        per_pool = max(4, min(32, ncpu // files))
        ctxs = []
        for _ in range(files):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                dd = VarDctDecoder(local)
            ctxs.append((dd, R.JxlThreadParallelRunnerCreate(None, per_pool),
                         torch.empty((h, w, 3), dtype=torch.float32, device=f"cuda:{local}"), abi.CodestreamInfo()))
        reps_each = 4
        errors = []

        def worker(i, n):
            dd, pool, o, inf = ctxs[i]
            for _ in range(n):
                rc = L.jxlhip_decode_codestream(dd.ctx, runner, pool, blob, len(blob), 1, None, o.data_ptr(), w * 12, 0, C.byref(inf))
                if rc:
                    errors.append(rc)

        for phase_reps in (1, reps_each):  # a warm-up round (allocations), then the timed one
            ths = [threading.Thread(target=worker, args=(i, phase_reps)) for i in range(files)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            dt_files = time.perf_counter() - t0
        same = all(torch.equal(ctxs[0][2], c[2]) for c in ctxs[1:]) and torch.equal(ctxs[0][2], out)
        for dd, pool, _, _ in ctxs:
            dd.close()
            R.JxlThreadParallelRunnerDestroy(pool)
        if errors or not same:
            res["codestream_8k_rgb_files_in_flight"] = {"error": f"rc {errors[:3]} identical {same}"}
        else:
            res["codestream_8k_rgb_files_in_flight"] = {
                "value": round(files * reps_each * w * h / dt_files / 1e6, 1), "files_in_flight": files, "threads_per_file": per_pool,
                "files": files * reps_each, "ms_per_file": round(dt_files / (files * reps_each) * 1e3, 2),
                "what": "aggregate whole-file rate: the same stream on 4 contexts / streams / runner pools at once, one host "
                        "thread each; every file decoded completely (identical pixels checked)"}
    except Exception as ex:
        res["codestream_8k_rgb_files_in_flight"] = {"error": repr(ex)[:200]}
    dec.close()
    # The unmodified djxl on three decoders: djxl_hip (libjxl's JxlDecoder + the seam -> this back-end), djxl_ref_v8 (the
    # reference with its decode hot path on the 8-lane Highway stand-in: libjxl's SIMD code paths, the honest CPU partner)
    # and djxl_ref (one lane: the tests' checker).  Two outputs each: the float frame djxl asks for under --disable_output
    # (398 MB per repetition, which djxl allocates afresh every time), and an 8-bit PPM of the stream's 8-bit sRGB twin
    # (tests/data/e2e_8k_d1_srgb8.jxl: what djxl does for a file made from a PNG; 100 MB back over PCIe).
    import tempfile
    path8 = os.path.join(ROOT, "tests", "data", "e2e_8k_d1_srgb8.jxl")
    for tool in ("djxl_hip", "djxl_ref_v8", "djxl_ref"):
        exe = os.path.join(ROOT, "oracle", "_ref", tool)
        if not os.path.exists(exe):
            res[tool] = {"error": "not built"}
            continue
        env = dict(os.environ, JXLHIP_SEAM_VERBOSE="1",
                   LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref") + ":" + os.path.join(ROOT, "libjxl_amd", "csrc") +
                   ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        entry = {}
        for mode in ("f32", "u8_ppm"):
            if mode == "u8_ppm" and not os.path.exists(path8):
                continue
            got = None
            with tempfile.TemporaryDirectory() as td:
                args = [path, "--disable_output"] if mode == "f32" else [path8, os.path.join(td, "o.ppm")]
                for threads in sorted({t for t in (16, 64) if t <= ncpu}):
                    try:
                        r = subprocess.run([exe] + args + ["--num_reps", "10", "--num_threads", str(threads)],
                                           capture_output=True, text=True, env=env, timeout=120)
                    except subprocess.TimeoutExpired:
                        continue
                    m = re.search(r"([0-9.]+) MP/s", r.stderr)
                    if r.returncode == 0 and m and (got is None or float(m.group(1)) > got["value"]):
                        got = {"value": float(m.group(1)), "threads": threads, "ms_per_rep": round(w * h / float(m.group(1)) / 1e3, 1),
                               "line": [l for l in r.stderr.strip().splitlines() if "MP/s" in l][-1][:200]}
                        # the binding's own clock (JXLHIP_SEAM_VERBOSE): what of a repetition is the back-end, what is
                        # libjxl's front end (its Modular DC groups) and what djxl + libjxl's headers / allocations
                        seam = [l for l in r.stderr.splitlines() if l.startswith("jxlhip seam: frame")]
                        if seam:
                            got["seam_ms_last_rep"] = seam[-1].split("; ms: ")[-1][:400]
            entry[mode] = got or {"error": "djxl failed"}
        entry["value"] = entry.get("u8_ppm", entry["f32"]).get("value")
        res[tool] = entry
    for mode in ("f32", "u8_ppm"):
        try:
            hv = res["djxl_hip"][mode]["value"]
            res.setdefault("djxl_hip_over_ref_v8", {})[mode] = round(hv / res["djxl_ref_v8"][mode]["value"], 2)
            res.setdefault("djxl_hip_over_ref_one_lane", {})[mode] = round(hv / res["djxl_ref"][mode]["value"], 2)
        except Exception:
            pass
    return res


def single_process_multi(args, torch):
    """`--gpus N --multi-mode single`: ONE process, one jxlhip_create_multi context over the N devices (include/jxl_hip.h):
    the stripes, the halo rows and the gather into devices[0] are stream-ordered peer copies below the C ABI -- no
    torch.distributed, no NCCL bootstrap.  The frame is handed over ONCE through the host-pointer calls (the multi
    context routes every group to the device that owns it); a step = jxlhip_decode_frame into a frame on devices[0]
    (the gather inside the step, like `value` of the process mode).  Checked once against a one-device context."""
    import ctypes as C
    from libjxl_amd import VarDctDecoder, abi, synth
    n = args.gpus
    name = args.config or "c4"
    cfg = dict(CONFIGS[name])
    for k in ("width", "height", "gab", "epf", "mix"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    custom = any(getattr(args, k) is not None for k in ("width", "height", "gab", "epf", "mix"))
    xs, ys = cfg["width"], cfg["height"]
    ndev = torch.cuda.device_count()
    forced = os.environ.get("JXLHIP_BENCH_DEVICE")
    devices = [int(forced)] * n if forced is not None else [i % ndev for i in range(n)]
    params, t = synth.synth_frame(xs, ys, mix=resolve_mix(cfg["mix"]), gab=bool(cfg["gab"]), epf_iters=cfg["epf"], device="cpu",
                                  coeff_type=int(cfg["coeff32"]), intensity_target=cfg["intensity"], quant_mul=cfg["quant_mul"])
    L = abi.load_library()
    one = VarDctDecoder(devices[0])
    one.begin_frame(params)
    dq = one.default_dequant_tables()
    table_host = dq.cpu().numpy()
    ctx = C.c_void_p()
    devs = (C.c_int * n)(*devices)

    def chk(rc, what):
        if rc:
            raise SystemExit(f"{what}: {L.jxlhip_last_error(ctx).decode() if ctx else rc}")
    chk(L.jxlhip_create_multi(devs, n, None, C.byref(ctx)), "jxlhip_create_multi")
    p = abi.make_params(params)
    chk(L.jxlhip_frame_begin(ctx, C.byref(p)), "frame_begin")
    npy = {k: ([x.numpy() for x in v] if isinstance(v, list) else v.numpy()) for k, v in t.items()}
    dc3 = (C.c_void_p * 3)(*[a.ctypes.data for a in npy["dc"]])
    chk(L.jxlhip_upload_side_info(ctx, npy["ac_strategy"].ctypes.data, npy["raw_quant"].ctypes.data, npy["epf_sharpness"].ctypes.data,
                                  npy["ytox_map"].ctypes.data, npy["ytob_map"].ctypes.data, dc3, table_host.ctypes.data), "upload_side_info")
    ng = ((xs + 255) // 256) * ((ys + 255) // 256)
    for g in range(ng):
        ptrs = (C.c_void_p * 3)(*[c[g * 65536:].ctypes.data for c in npy["coeffs"]])
        chk(L.jxlhip_submit_group(ctx, g, ptrs, 65536), "submit_group")
    full = torch.empty((ys, xs, 3), dtype=torch.float32, device=f"cuda:{devices[0]}")

    def step():
        chk(L.jxlhip_decode_frame(ctx, C.c_void_p(full.data_ptr()), xs * 12, 0), "decode_frame")

    def sync():
        chk(L.jxlhip_sync(ctx), "sync")
    step()
    sync()
    # the one-device frame of the same inputs
    dev_t = {k: ([x.to(f"cuda:{devices[0]}") for x in v] if isinstance(v, list) else v.to(f"cuda:{devices[0]}")) for k, v in t.items()}
    one.set_inputs(dev_t, dq)
    ref = one.decode_frame()
    one.sync()
    same = bool(torch.equal(ref, full))
    ck = lambda x: int(x.view(torch.int32).to(torch.int64).sum().item() & 0xFFFFFFFFFFFF)  # noqa: E731
    sums = (ck(full), ck(ref))
    del ref, dev_t
    one.close()
    settle = int(math.ceil(args.settle_ms / max(0.04, xs * ys / (100e9 * n) * 1e3))) if args.settle_ms > 0 else 0
    for i in range(settle + args.warmup):
        step()
        if i % 16 == 15:
            sync()
    sync()
    clocks = ClockSampler(devices[0])
    clocks.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    clk = clocks.stop()
    host_frame = None
    try:  # every stripe through its own device's PCIe link into a host frame (jxlhip_decode_frame_host)
        import numpy as np
        host = np.zeros((ys, xs, 3), np.float32)
        chk(L.jxlhip_decode_frame_host(ctx, host.ctypes.data, xs * 12, 0), "decode_frame_host")
        t1 = time.perf_counter()
        reps = max(2, min(args.steps, 8))
        for _ in range(reps):
            chk(L.jxlhip_decode_frame_host(ctx, host.ctypes.data, xs * 12, 0), "decode_frame_host")
        host_frame = {"value": round(xs * ys / ((time.perf_counter() - t1) / reps) / 1e6, 1), "unit": "Mpixels/s",
                      "what": "jxlhip_decode_frame_host on the same context: every stripe leaves through its own device's PCIe link, "
                              "no gather on one device; synchronous per frame"}
    except SystemExit as ex:
        host_frame = {"error": str(ex)[:200]}
    L.jxlhip_destroy(ctx)
    px = xs * ys
    ms_step = dt / args.steps * 1e3
    cb = 4 if cfg["coeff32"] else 2
    b_alg = algorithmic_bytes(xs, ys, cb)
    ach = b_alg / (ms_step * 1e-3) / 1e9
    line = {"metric": METRIC[name] if not custom else "Mpixels/s decode (VarDCT, custom workload)", "value": round(px / (dt / args.steps) / 1e6, 1),
            "unit": "Mpixels/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}{'*' if custom else ''}: {xs}x{ys} RGB VarDCT frame, gab={cfg['gab']} epf_iters={cfg['epf']}, "
                                   f"{'int32' if cfg['coeff32'] else 'int16'} coefficients, strategy mix {cfg['mix']}, linear RGB f32 out",
                       "n_multi_mode": "single", "devices": devices, "stripes": n, "gather_in_step": True,
                       "device_settle_ms": args.settle_ms, "device_settle_steps": settle,
                       "gathered_frame_equals_one_gpu_frame": same, "gathered_frame_checksum": sums[0], "one_gpu_frame_checksum": sums[1],
                       "device_clocks_during_timed_steps": {"value": clk} if clk else None,
                       "scaling_basis": "one process, jxlhip_create_multi: `value` has the gather into devices[0] inside the step "
                                        "(peer copies: bound by devices[0]'s incoming xGMI links, DESIGN.md section 7)"},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS / n, 4), "traffic": None,
                         "what": "frame's algorithmic bytes / whole step / (N x 8 TB/s)", "algorithmic_bytes_frame": b_alg},
            "host_frame": host_frame}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="BASELINE workload; default c3 at --gpus 1, c4 at --gpus N>1")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--gab", type=int, default=None)
    ap.add_argument("--epf", type=int, default=None)
    ap.add_argument("--mix", default=None,
                    help="d1 | dct8 | dct32 | all | real4k (the strategy shares of a genuine libjxl d1.0 stream), or an "
                         "explicit area mix 'strategy:share,...' (e.g. 18:1 = all 64x64)")
    ap.add_argument("--coeff32", action="store_true", default=None)
    ap.add_argument("--no-gather", action="store_true", help="N>1: leave the output stripes sharded")
    ap.add_argument("--multi-mode", choices=("process", "single"), default="process",
                    help="N>1: 'process' = one process per GPU, halo rows and the gather over torch.distributed / RCCL "
                         "(libjxl_amd/stripes.py; the driver's launch); 'single' = ONE process over the N devices through "
                         "jxlhip_create_multi (stream-ordered peer copies below the C ABI, no NCCL bootstrap): run it as "
                         "plain `python bench.py --gpus N --multi-mode single` (under torch.distributed.run rank 0 does "
                         "the work and the other ranks only keep the barriers)")
    ap.add_argument("--gather-mode", choices=("streamed", "after"), default="streamed",
                    help="N>1: 'streamed' (default) = every rank posts its rows to rank 0 as their launches are queued "
                         "and leaves them in flight across the next step (StripeDecoder.decode_gathered: a step costs "
                         "max(kernels, gather)); 'after' = rounds 2-5: the gather starts when the stripe is done and the "
                         "step waits for it (kernels + gather)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive measurement (N=1)")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed steps for this long before the W warm-up steps: the device's clock ramp (0 = none)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the whole-file block (N=1, default workload)")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="N=1: decoder contexts (each on a HIP stream of its own) the steps rotate over, default 3: launch "
                         "gaps, k_prepare and kernel tails of one frame overlap the next frame's kernels: the side figure "
                         "`frames_in_flight` (1 = skip it).  `value` is always one context, one frame at a time")
    ap.add_argument("--calib-copy", action="store_true",
                    help="run one known-size (1 GiB) device copy so PMC passes can be calibrated")
    ap.add_argument("--cpu-sample", type=int, nargs=2, default=None)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from libjxl_amd import VarDctDecoder, stripes, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.multi_mode == "single" and args.gpus > 1:
        if rank == 0:
            single_process_multi(args, torch)
        return
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (or --multi-mode single)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the VarDCT back-end has no CPU path")
    # (smoke tests of the N > 1 path on a one-GPU box: JXLHIP_BENCH_DEVICE=0 puts every rank on that device and
    # JXLHIP_BENCH_BACKEND=gloo moves the halo rows through the host -- RCCL refuses two ranks on one GPU.  Timings of such
    # a run mean nothing; the driver's runs use neither variable.)
    if os.environ.get("JXLHIP_BENCH_DEVICE") is not None:
        local = int(os.environ["JXLHIP_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("JXLHIP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    name = args.config or ("c3" if world == 1 else "c4")
    cfg = dict(CONFIGS[name])
    for k in ("width", "height", "gab", "epf", "mix"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    if args.coeff32:
        cfg["coeff32"] = True
    custom = any(getattr(args, k) is not None for k in ("width", "height", "gab", "epf", "mix")) or bool(args.coeff32)
    xs, ys = cfg["width"], cfg["height"]
    mix = resolve_mix(cfg["mix"])
    params, t = synth.synth_frame(xs, ys, mix=mix, gab=bool(cfg["gab"]), epf_iters=cfg["epf"],
                                  device=f"cuda:{local}", coeff_type=int(cfg["coeff32"]),
                                  intensity_target=cfg["intensity"], quant_mul=cfg["quant_mul"])
    if os.environ.get("JXLHIP_BENCH_NO_USED_ACS"):  # experiments: the caller does not know the frame's strategies (the merged phase-1 launch)
        params["used_acs"] = 0
    dec = VarDctDecoder(local)
    sd = stripes.StripeDecoder(dec, params, rank, world)
    dq = dec.default_dequant_tables()
    dec.set_inputs(t, dq)
    out = dec.alloc_output()
    gather = world > 1 and not args.no_gather
    full = sd.alloc_gather(out) if gather else None
    streamed = gather and args.gather_mode == "streamed"
    out_b = dec.alloc_output() if streamed else None  # the stripe buffers alternate: frame k's rows travel while k + 1 is decoded

    # N = 1, the side figure `frames_in_flight`: a pool of decoder contexts, one HIP stream each, every context with
    # its OWN device copy of the coefficient stream, the side info and the dequant tables (round 4 shared one copy: frame
    # k + 1's coefficient reads could then be served from the L2 / Infinity Cache that frame k had just filled): step k
    # runs on context k % F.  Every step is a whole frame -- k_prepare, transforms, fused filter kernel -- nothing is
    # shared between steps; what overlaps is one frame's launch gaps / tail with the next frame's kernels.
    # `value` is NOT this: it is one context, one frame at a time (the figure of rounds 1-3).
    inflight = max(1, args.frames_in_flight if args.frames_in_flight is not None else (3 if world == 1 else 1))
    if world > 1:
        inflight = 1

    def clone_inputs(tt):
        return {k: ([x.clone() for x in v] if isinstance(v, (list, tuple)) else v.clone()) for k, v in tt.items()}

    slots = [(dec, out)]
    private_inputs = []
    for _ in range(inflight - 1):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            d2 = VarDctDecoder(local)  # its launches go to st
        d2.begin_frame(params)
        t2, dq2 = clone_inputs(t), dq.clone()
        private_inputs.append((t2, dq2))
        d2.set_inputs(t2, dq2)
        slots.append((d2, d2.alloc_output()))
    torch.cuda.synchronize()
    counter = [0]

    def step():
        if world == 1:
            d, o = slots[counter[0] % inflight]
            counter[0] += 1
            d.decode_frame(o)
            return
        if streamed:
            counter[0] += 1
            sd.decode_gathered(out if counter[0] & 1 else out_b, full)
            return
        sd.decode(out)
        if gather:
            sd.gather(out, full)

    def drain():
        if streamed:
            sd.wait_gather()  # the current stream waits for every transfer still in flight (rank 0: every receive)

    def step_one():
        dec.decode_frame(out)

    red_dev = "cuda" if (world == 1 or dist.get_backend() == "nccl") else "cpu"  # (gloo smoke runs reduce on the host)
    trace = os.environ.get("JXLHIP_BENCH_TRACE") is not None

    def note(msg):
        if trace:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.calib_copy:
        a = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        del a, b
    # Device settle (set-up, before the W warm-up steps of every timed measurement): the GPU takes tens of milliseconds of
    # continuous work to reach the clocks it then holds -- measured: K = 20 steps after W = 5 give 100 Gpx/s, the same 20
    # steps after W = 100 give 121 (profiles/r04_short_runs.txt); a driver's short (W, K) would time the ramp, not the
    # decoder, and a measurement that follows host-side work (the PCIe and whole-file blocks) starts from idle clocks
    # again.  The same steps as the timed ones, untimed, for --settle-ms (default 60 ms; 0 = off), reported in
    # config.device_settle_ms / _steps.  (A step COUNT from the frame size, not from a clock: every rank must run the
    # same number of collective steps.)
    settle_steps = int(math.ceil(args.settle_ms / max(0.04, xs * ys / (100e9 * world) * 1e3))) if args.settle_ms > 0 else 0
    settle_on = [True]

    clocks = ClockSampler(local)
    clock_log = {}

    def timed(fn, tag=None):
        for i in range(settle_steps if settle_on[0] else 0):
            fn()
            if i % 16 == 15:
                dec.sync()
        for _ in range(args.warmup):
            fn()
        drain()
        dec.sync()  # also surfaces stream errors
        fence()
        if tag and rank == 0:
            clocks.start()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        drain()
        fence()
        t = time.perf_counter() - t0
        if tag and rank == 0:
            clock_log[tag] = clocks.stop()
        if world > 1:
            tt = torch.tensor([t], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t

    # Order: (1) `unsettled` -- the W warm-up steps and the K timed steps straight after set-up, the device at whatever
    # clocks it idles at (what a driver's short (W, K) run sees without the settle phase); (2) `value` -- the same K
    # steps on ONE context after the settle phase; (3) `frames_in_flight` -- the K steps rotated over the pool.
    dt_unsettled = None
    if world == 1 and settle_steps > 0:
        settle_on[0] = False
        dt_unsettled = timed(step_one)
        settle_on[0] = True
    note("set-up done; timed steps")
    dt = timed(step_one if world == 1 else step, tag="value")
    note("timed steps done")
    # N > 1: what arrived on rank 0 is the frame one GPU decodes (checked once, untimed: a whole-frame context on rank 0's
    # device; the stripes' two-phase / fused kernels and the whole frame's are held bit-identical by the test suite)
    gathered_equals_one_gpu = gathered_checksum = one_gpu_checksum = None
    ranks_seen = None
    if world > 1:
        mine = torch.tensor([rank, local, torch.cuda.current_device()], dtype=torch.int64, device=red_dev)
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        ranks_seen = [dict(rank=int(v[0]), local_rank=int(v[1]), device=int(v[2])) for v in seen]
        if gather and rank == 0:
            try:
                whole = VarDctDecoder(local)
                whole.begin_frame(params)
                whole.set_inputs(t, dq)
                ref = whole.decode_frame()
                whole.sync()
                gathered_equals_one_gpu = bool(torch.equal(ref, full))
                gathered_checksum = int(full.view(torch.int32).to(torch.int64).sum().item() & 0xFFFFFFFFFFFF)
                one_gpu_checksum = int(ref.view(torch.int32).to(torch.int64).sum().item() & 0xFFFFFFFFFFFF)
                whole.close()
                del ref, whole
            except Exception as ex:
                gathered_equals_one_gpu = repr(ex)[:200]
    # `graph_replay`: the frame's launches recorded ONCE into a hipGraph (stream capture around jxlhip_decode_frame) and
    # replayed per step -- what a caller that decodes frame after frame of one geometry (video, a tile server) can do;
    # the library needs nothing but to keep its per-frame state inside the graph (context.hip: DecodeFrameCoded).
    dt_graph = None
    # OPT-IN (JXLHIP_BENCH_GRAPH=1): measured slower than direct launches (profiles/r05_graph_replay_bench.json: 0.291 vs
    # 0.282 ms), and a side figure must not be able to take the whole line down with it.
    if world == 1 and os.environ.get("JXLHIP_BENCH_GRAPH") and not os.environ.get("JXLHIP_BENCH_NO_GRAPH"):
        try:
            cs = torch.cuda.Stream()
            main_stream = torch.cuda.current_stream()
            cs.wait_stream(main_stream)
            dec.set_stream(cs)
            with torch.cuda.stream(cs):
                dec.decode_frame(out)  # (buffers sized, nothing left to allocate inside the capture)
            cs.synchronize()
            note("graph: warm-up on the capture stream done")
            keep = out.clone()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=cs):
                dec.decode_frame(out)
            note("graph: captured")
            out.zero_()
            graph.replay()
            torch.cuda.synchronize()
            note("graph: first replay done")
            assert torch.equal(out, keep), "the replayed graph decoded a different frame"
            dt_graph = timed(graph.replay)
            note("graph: timed replays done")
            del keep
        except Exception as ex:  # (the figure is a side measurement: the line must come out)
            dt_graph = repr(ex)[:200]
        finally:
            dec.set_stream(main_stream)
    dt_flight = None
    if world == 1 and inflight > 1:
        for d, _ in slots:
            d.set_concurrency_hint(inflight)  # (only moves the frame size from which the fused kernel is taken)
        dt_flight = timed(step, tag="frames_in_flight")
        for d, _ in slots:
            d.set_concurrency_hint(1)  # from here on `dec` runs alone again: the per-kernel pass
        for d2, o2 in slots[1:]:
            assert torch.equal(o2, out), "a pooled context decoded a different frame"
    # N > 1: the same frame with the output stripes left sharded in each GPU's HBM (a consumer on the device, or
    # every GPU writing its own stripe to the host): the form in which the split scales -- the gather of a 1.59 GB
    # float frame into ONE GPU is per-link bound
    dt_sharded = timed(lambda: sd.decode(out)) if gather else None
    note("sharded steps done")

    # N > 1: what the HOST spends per step to enqueue a rank's work (three calls into the library + one batch of
    # sends / receives; no synchronisation inside the loop) -- on a 16K frame over 8 GPUs a rank's kernels take ~170 us,
    # and a step is host-bound as soon as this figure comes near that -- and the step when every stripe leaves through its
    # OWN GPU's PCIe link into pinned host memory instead of being gathered on one device (`host_sharded`: the consumer
    # is the host)
    host_enqueue_us, dt_host_sharded = None, None
    if world > 1:
        fence()
        n_h = max(4, min(args.steps, 20))
        t0 = time.perf_counter()
        for _ in range(n_h):
            sd.decode(out)
        th = (time.perf_counter() - t0) / n_h
        fence()
        tt = torch.tensor([th], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        host_enqueue_us = round(float(tt.item()) * 1e6, 1)
        pinned = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)

        def step_host_sharded():
            sd.decode(out)
            pinned.copy_(out, non_blocking=True)
        dt_host_sharded = timed(step_host_sharded)
        note("host-sharded steps done")

    # N > 1: where a step's time goes on each rank (HIP events on the compute stream around the phases of
    # StripeDecoder.decode and the gather; average over the steps, then the MAX over ranks): blocks = phase 1,
    # interior = halo export + posting the sends + the rows that need no halo, halo_wait = what is left of the exchange
    # after that, boundary = halo import + the two boundary block rows, gather = the stripes into rank 0's frame
    phase_ms = None
    if world > 1:
        T = {}
        n_prof = max(4, min(args.steps, 20))
        for _ in range(n_prof):
            sd.decode(out, timing=T)
            if gather:
                sd.gather(out, full)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            T.setdefault("gather", []).append(ev)
        fence()
        order = ["t0", "blocks", "interior", "halo_wait", "boundary", "gather"]
        vals = []
        for a, b in zip(order, order[1:]):
            if a in T and b in T and len(T[a]) == len(T[b]):
                vals.append(sum(x.elapsed_time(y) for x, y in zip(T[a], T[b])) / len(T[a]))
            else:
                vals.append(0.0)
        tt = torch.tensor(vals, dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        phase_ms = {k: round(float(v), 4) for k, v in zip(order[1:], tt.tolist())}

    note("per-phase pass done")
    # per-kernel device time with HIP events on the launch stream (own pass)
    dec.profile(True)
    for _ in range(args.steps):
        if world == 1:
            dec.decode_frame(out)  # the launches of the timed step (fused kernel when the frame qualifies)
        else:
            dec.decode_blocks()
            dec.decode_filters(out)
    prof = dec.profile_read()
    dec.profile(False)
    kern = {k: round(ms / max(n, 1), 4) for k, (ms, n) in prof.items()}

    # the same step when the boundary hands over HOST buffers: pinned H2D of the coefficient
    # stream, kernels, pinned D2H of the pixels.  Never `value`.  Serial on one stream (round 2's figure), and
    # pipelined: H2D of frame k+1 || kernels of frame k || D2H of frame k-1 on three streams over two sets of
    # buffers -- for the f32 frame and for sRGB RGBA8 (what djxl writes by default: a third of the bytes back).
    pcie = None
    if world == 1 and not args.no_pcie:
        pcie = pcie_inclusive(torch, dec, params, t, dq, out, xs, ys, max(4, min(args.steps, 12)))
    if rank == 0:
        px = xs * ys
        ms_step = dt / args.steps * 1e3
        value = px / (dt / args.steps) / 1e6
        cb = 4 if cfg["coeff32"] else 2
        # roofline: the FRAME's algorithmic bytes over the whole STEP -- every launch and every gap
        # between them -- against the 8 TB/s HBM peak (`frac`).  The per-kernel views stay beside it:
        #   frac_dominant_kernel  the frame's bytes over the dominant kernel's time alone (SURVEY 8(d) read
        #                         literally; it flatters: that kernel does not move the whole frame's bytes)
        #   frac_kernel           the dominant kernel's OWN bytes over its time
        # `traffic` = HBM bytes of ALL the step's launches from the TCC counters (profiles/pmc_traffic.json: separate
        # FETCH_SIZE / WRITE_SIZE passes of this command, calibrated on a known copy); traffic_kernel = the
        # dominant kernel's share.
        dom = "fused" if "fused" in kern else ("filters" if "filters" in kern else max(kern, key=kern.get))
        y0, y1 = sd.rows[rank]
        b_alg = algorithmic_bytes(xs, y1 - y0, cb)
        b_alg_frame = algorithmic_bytes(xs, ys, cb)
        b_own = xs * (y1 - y0) * 24  # planes in + pixels out (k_fused: an upper bound, its DCT8 cells come as coefficients)
        if dom == "blocks":  # all-DCT32X32 frame without filters: the class kernel reads coefficients, writes pixels
            b_own = b_alg
        achieved = b_alg_frame / (ms_step * 1e-3) / 1e9
        traffic, traffic_kernel, tsrc = None, None, None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf) and world == 1 and name == "c3" and not custom:
            try:
                tj = json.load(open(tf))
                traffic_kernel = tj.get("filters")  # pmc_summarize.py files k_fused under "filters"
                traffic = sum(v for k, v in tj.items() if not k.startswith("_") and isinstance(v, (int, float)))
                tsrc = ("profiles/pmc_traffic.json (replayed, not measured in this run: rocprofv3 --pmc FETCH_SIZE / "
                        "WRITE_SIZE passes of this command, tools/pmc_traffic.sh; commit " + str(tj.get("_commit", "?")) + ")")
            except Exception:
                traffic = None
        # instruction-issue roofline: wave-instructions of the step's launches (SQ_INSTS_VALU, a rocprofv3 --pmc pass of this
        # command, replayed from profiles/pmc_valu.json like `traffic`) over the step, against the probe's issue peak
        valu = None
        vf = os.path.join(ROOT, "profiles", "pmc_valu.json")
        if os.path.exists(vf) and world == 1 and name == "c3" and not custom:
            try:
                vj = json.load(open(vf))
                n_valu = sum(v.get("SQ_INSTS_VALU", 0) for k, v in vj.items() if not k.startswith("_"))
                n_salu = sum(v.get("SQ_INSTS_SALU", 0) for k, v in vj.items() if not k.startswith("_"))
                ach = n_valu / (ms_step * 1e-3) / 1e9
                valu = {"bound": "valu-issue", "achieved": round(ach, 1), "peak": round(VALU_ISSUE_PEAK_GINSTR, 1),
                        "unit": "G wave-instructions/s", "frac": round(ach / VALU_ISSUE_PEAK_GINSTR, 4),
                        "valu_instructions_per_step": int(n_valu), "salu_instructions_per_step": int(n_salu),
                        "per_kernel": {k: v for k, v in vj.items() if not k.startswith("_")},
                        "peak_source": "tools/probes/valu_issue.hip, profiles/r05_valu_issue_probe.txt: the mix pk_fma / pk_add / add_dpp / "
                                       "fma at 4 waves per SIMD retires one wave64 instruction per 4.15 cycles per SIMD (2.4 GHz event "
                                       "clock) x 1024 SIMDs; plain v_fma_f32 alone: 2.39 cycles = "
                                       f"{VALU_ISSUE_PEAK_FMA_GINSTR:.0f} G/s (frac_of_fma_peak below)",
                        "frac_of_fma_peak": round(ach / VALU_ISSUE_PEAK_FMA_GINSTR, 4),
                        "counter_source": "profiles/pmc_valu.json (replayed: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU pass of this "
                                          "command, tools/pmc_valu.sh; commit " + str(vj.get("_commit", "?")) + ")"}
            except Exception:
                valu = None
        hbm_frac = achieved / HBM_PEAK_GBS
        bound = "valu-issue" if (valu and valu["frac"] > hbm_frac) else "hbm"
        line = {
            "metric": METRIC[name] if not custom else "Mpixels/s decode (VarDCT, custom workload)", "value": round(value, 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}{'*' if custom else ''}: {xs}x{ys} RGB VarDCT frame, gab={cfg['gab']} "
                                   f"epf_iters={cfg['epf']}, {'int32' if cfg['coeff32'] else 'int16'} "
                                   f"coefficients, strategy mix {cfg['mix']}, intensity_target {cfg['intensity']:g}, "
                                   f"linear RGB f32 out",
                       "frames_in_flight": 1,
                       "device_settle_ms": args.settle_ms, "device_settle_steps": settle_steps,
                       "stripes": world, "halo_rows": dec.halo_rows(),
                       "gather_in_step": bool(gather),
                       "phase_ms_max_over_ranks": phase_ms,
                       "host_enqueue_us_per_step_max_over_ranks": host_enqueue_us,
                       "interior_first": os.environ.get("JXLHIP_STRIPES_INTERIOR_FIRST", "1") != "0" if world > 1 else None,
                       "kernel_ms": kern,
                       "kernel_ms_sum": round(sum(kern.values()), 4),
                       "kernel_ms_what": "a pass of its own with ONE HIP event between consecutive launches on the launch stream "
                                         "(jxlhip_profile_enable): an event between two launches keeps the second from being "
                                         "dispatched behind the first, so every launch reads a few us longer than inside the timed "
                                         "steps and the parts can add up to more than ms_per_step (rounds 1-5 recorded two events "
                                         "per boundary: +9 %); rocprofv3's averages of the same step: profiles/r06_c3_kernel_stats.csv",
                       "device_clocks_during_timed_steps": clock_log or None,
                       "n_multi_mode": args.multi_mode if world > 1 else None,
                       "gather_mode": (args.gather_mode if gather else None),
                       "ranks_seen": ranks_seen,
                       "gathered_frame_equals_one_gpu_frame": gathered_equals_one_gpu,
                       "gathered_frame_checksum": gathered_checksum, "one_gpu_frame_checksum": one_gpu_checksum},
            "roofline": {"bound": bound, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "what": "frame's algorithmic bytes / whole step (all launches + gaps) of `value`: ONE frame in flight, "
                                 "settled clocks",
                         "frac_dominant_kernel": round(b_alg / (kern[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "frac_kernel": round(b_own / (kern[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_kernel": traffic_kernel, "traffic_source": tsrc,
                         "kernel": {"filters": "k_filters_fast",
                                    "fused": "k_fused_pc",
                                    "blocks": "k_transform_mfma32<EMIT>"}.get(dom, dom),
                         "kernel_ms": kern[dom],
                         "algorithmic_bytes_per_launch": b_alg,
                         "algorithmic_bytes_frame": b_alg_frame,
                         "kernel_own_bytes_per_launch": b_own},
        }
        if valu:
            line["roofline_valu"] = valu
            line["roofline"]["bound_what"] = ("the higher of roofline.frac (HBM: algorithmic bytes / step / 8 TB/s) and roofline_valu.frac "
                                              "(instruction issue: wave-instructions / step / the probe's peak)")
        if world == 1:
            line["one_frame_in_flight"] = {"value": round(value, 1), "unit": "Mpixels/s", "ms_per_step": round(ms_step, 4),
                                           "what": "= `value` (round 4 reported the frames-in-flight rate as `value` and this "
                                                   "figure beside it; the series r1 79.5 / r2 89.6 / r3 98.7 / r4 106.9 is this one)"}
        if dt_unsettled is not None:
            line["unsettled"] = {"value": round(px / (dt_unsettled / args.steps) / 1e6, 1), "unit": "Mpixels/s",
                                 "ms_per_step": round(dt_unsettled / args.steps * 1e3, 4),
                                 "what": "the same W warm-up + K timed steps on one context straight after set-up, without the "
                                         "device-settle phase (--settle-ms 0): the device's clocks still ramping"}
        if isinstance(dt_graph, float):
            line["graph_replay"] = {"value": round(px / (dt_graph / args.steps) / 1e6, 1), "unit": "Mpixels/s",
                                    "ms_per_step": round(dt_graph / args.steps * 1e3, 4),
                                    "frac": round(b_alg_frame / (dt_graph / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                    "what": "one context, one frame at a time, the frame's three launches captured once into a "
                                            "hipGraph and replayed per step (pixels checked against the direct call)"}
        elif dt_graph is not None:
            line["graph_replay"] = {"error": dt_graph}
        if dt_flight is not None:
            line["frames_in_flight"] = {"value": round(px / (dt_flight / args.steps) / 1e6, 1), "unit": "Mpixels/s",
                                        "ms_per_step": round(dt_flight / args.steps * 1e3, 4), "contexts": inflight,
                                        "private_inputs": True,
                                        "frac": round(b_alg_frame / (dt_flight / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                        "what": "aggregate rate of the same steps rotated over `contexts` decoder contexts, each "
                                                "on its own HIP stream with its OWN device copy of coefficients / side info / "
                                                "dequant tables (settled clocks); outputs asserted bit-equal"}
        if dt_sharded is not None:
            line["sharded"] = {"value": round(px / (dt_sharded / args.steps) / 1e6, 1), "unit": "Mpixels/s",
                               "ms_per_step": round(dt_sharded / args.steps * 1e3, 4),
                               "what": "the same frame, output stripes left in each GPU's HBM (no gather)"}
        if world > 1:
            line["config"]["scaling_basis"] = (
                "`value` = the frame GATHERED on rank 0 inside the step (BASELINE configs[3]): 7/8 of a 1.59 GB float frame over "
                "rank 0's 7 point-to-point xGMI links = >= 1.3 ms at the 153 GB/s link rate against ~1.1 ms for the whole frame on "
                "ONE GPU -- its floor is the gather, whatever the kernels do (streamed: max(kernels, gather) per step).  The split "
                "itself is measured by `sharded` (stripes left in each GPU's HBM) and `host_sharded` (every stripe through its own "
                "GPU's PCIe link): the >= 6x of the north star is claimed on `sharded`; DESIGN.md section 7 holds the expected "
                "N = 2 / 4 / 8 figures of all three")
            line["config"]["value_gathered"] = round(value, 1)
            line["config"]["value_sharded"] = round(px / (dt_sharded / args.steps) / 1e6, 1) if dt_sharded else None
            line["config"]["value_host_sharded"] = round(px / (dt_host_sharded / args.steps) / 1e6, 1) if dt_host_sharded else None
        if dt_host_sharded is not None:
            line["host_sharded"] = {"value": round(px / (dt_host_sharded / args.steps) / 1e6, 1), "unit": "Mpixels/s",
                                    "ms_per_step": round(dt_host_sharded / args.steps * 1e3, 4),
                                    "what": "the same frame, every rank copying its output stripe to pinned host memory over "
                                            "its own GPU's PCIe link (no gather on one device)"}
        if pcie:
            line["pcie_inclusive"] = pcie
        if world == 1 and name == "c3" and not custom and not args.no_cpu_baseline:
            # the same step on the strategy shares of a GENUINE libjxl d1.0 stream (42 % 64x64, 32 % 32x32, 7 % DCT8:
            # what the reference encoder actually emits) -- the bench workload's mix is the contract's
            try:
                rp, rt = synth.synth_frame(xs, ys, mix=resolve_mix("real4k"), gab=True, epf_iters=1, device=f"cuda:{local}")
                dec.begin_frame(rp)
                dec.set_inputs(rt, dq)
                rdt = timed(step_one)
                line["real_content_mix"] = {"value": round(px / (rdt / args.steps) / 1e6, 1), "unit": "Mpixels/s",
                                            "ms_per_step": round(rdt / args.steps * 1e3, 4), "frames_in_flight": 1,
                                            "what": f"{xs}x{ys}, Gaborish + EPF1, strategy shares of tests/data/real_4k_d1.npz; "
                                                    "one frame at a time, like `value`"}
                if inflight > 1:
                    keep = []
                    for (d, _), (_, dq2) in zip(slots[1:], private_inputs):
                        rt2 = clone_inputs(rt)
                        keep.append(rt2)
                        d.begin_frame(rp)
                        d.set_inputs(rt2, dq2)
                    for d, _ in slots:
                        d.set_concurrency_hint(inflight)
                    counter[0] = 0
                    rdtf = timed(step)
                    for d, _ in slots:
                        d.set_concurrency_hint(1)
                    line["real_content_mix"]["in_flight"] = {"value": round(px / (rdtf / args.steps) / 1e6, 1), "contexts": inflight,
                                                              "ms_per_step": round(rdtf / args.steps * 1e3, 4), "private_inputs": True}
                    del keep
                del rt
            except Exception as ex:
                line["real_content_mix"] = {"error": repr(ex)[:200]}
        if world == 1 and name == "c3" and not custom and not args.no_e2e and not args.no_cpu_baseline:  # (profiling runs skip every side measurement)
            try:
                line["e2e"] = e2e_block(torch, local)
            except Exception as ex:  # the bench line must come out whatever happens to the side measurements
                line["e2e"] = {"error": repr(ex)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            if name == "c1":
                line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample or (1024, 1024), 1)
            else:
                line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample or (4096, 2160), 0)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
