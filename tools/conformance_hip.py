#!/usr/bin/env python3
"""Conformance harness for the djxl of this back-end.

libjxl's acceptance harness is tools/conformance/conformance.py (+ generator.py) of the reference tree: a corpus
directory with corpus.txt and, per test id, input.jxl, reference_image.npy (float32, [frame, y, x, channel]),
reference.icc and test.json (the image metadata djxl prints with --metadata_out plus rms_error / peak_error per
frame); the decoder under test is run as
    <decoder> input.jxl decoded_image.npy --metadata_out meta.json --icc_out decoded.icc --norender_spotcolors
and passes a test when the metadata agree and every frame is within the thresholds (conformance.py:34-66,112-238).
The reference tree is not present on the GPU box, so this file restates that contract (same corpus layout, same
decoder command line, same comparisons; the metadata check here walks ALL keys) and adds the corpus generator
(generator.py:21-98) for a mini-corpus of streams the reference encoder writes:

  python tools/conformance_hip.py generate --decoder oracle/_ref/djxl_ref --output DIR [--peak_error P --rmse R] a.jxl ...
  python tools/conformance_hip.py run --decoder oracle/_ref/djxl_hip --corpus DIR

tests/test_djxl.py drives both (GPU suite), and in the build container checks that the reference's own
conformance.py reaches the same verdicts on the same corpus.
"""
import argparse
import ctypes
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

TEST_KEYS = {"reconstructed_jpeg", "original_icc", "rms_error", "peak_error"}


def _convert_pixels(from_icc, to_icc, px):
    """ICC -> ICC conversion of float RGB with lcms2 (what conformance.py does when the decoder's output profile
    differs from the reference's); None when no lcms2 can be loaded."""
    lib = None
    for cand in (os.environ.get("LCMS2_LIB_PATH"), "liblcms2.so.2", "/opt/conda/lib/liblcms2.so.2"):
        if not cand:
            continue
        try:
            lib = ctypes.CDLL(cand)
            break
        except OSError:
            continue
    if lib is None:
        return None
    lib.cmsOpenProfileFromMem.restype = ctypes.c_void_p
    lib.cmsOpenProfileFromMem.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.cmsCreateTransform.restype = ctypes.c_void_p
    lib.cmsCreateTransform.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_uint32]
    lib.cmsDoTransform.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    lib.cmsDeleteTransform.argtypes = [ctypes.c_void_p]
    lib.cmsCloseProfile.argtypes = [ctypes.c_void_p]
    # TYPE_RGB_DBL: FLOAT_SH(1) | COLORSPACE_SH(PT_RGB = 4) | CHANNELS_SH(3) | BYTES_SH(0)
    fmt = (1 << 22) | (4 << 16) | (3 << 3) | 0
    a = lib.cmsOpenProfileFromMem(from_icc, len(from_icc))
    b = lib.cmsOpenProfileFromMem(to_icc, len(to_icc))
    if not a or not b:
        return None
    t = lib.cmsCreateTransform(a, fmt, b, fmt, 1, 0)  # relative colorimetric
    src = np.ascontiguousarray(px.reshape(-1, 3), dtype=np.float64)
    dst = np.empty_like(src)
    lib.cmsDoTransform(t, src.ctypes.data, dst.ctypes.data, src.shape[0])
    lib.cmsDeleteTransform(t)
    lib.cmsCloseProfile(a)
    lib.cmsCloseProfile(b)
    return dst.reshape(px.shape).astype(px.dtype)


def compare_npy(ref, ref_icc, dec, dec_icc, frame_idx, rmse_limit, peak_limit, log):
    if ref.shape != dec.shape:
        log(f"expected shape {ref.shape} but found {dec.shape}")
        return False
    rf, df = ref[frame_idx], dec[frame_idx].copy()
    nch = rf.shape[2]
    if ref_icc != dec_icc and peak_limit > 0:
        if nch < 3:
            log("only RGB images can be colour-converted")
            return False
        conv = _convert_pixels(dec_icc, ref_icc, df[:, :, :3])
        if conv is None:
            log("output ICC differs from the reference's and lcms2 is not available")
            return False
        df[:, :, :3] = conv
    err = np.abs(rf.astype(np.float64) - df.astype(np.float64))
    peak = float(err.max())
    rmses = [float(np.sqrt(np.mean(err[:, :, c] ** 2))) for c in range(nch)]
    log(f"RMSE: {rmses}, actual peak: {peak}")
    ok = True
    if max(rmses) > rmse_limit:
        log(f"RMSE too large: {max(rmses)} > {rmse_limit}")
        ok = False
    if peak > peak_limit:
        log(f"peak error too large: {peak} > {peak_limit}")
        ok = False
    return ok


def check_meta(dec, ref, log, path=""):
    if isinstance(ref, dict):
        if not isinstance(dec, dict):
            log(f"metadata {path}: not an object")
            return False
        ok = True
        for k, v in ref.items():
            if k in TEST_KEYS:
                continue
            if k not in dec:
                log(f"metadata {path}/{k}: missing")
                ok = False
            else:
                ok &= check_meta(dec[k], v, log, path + "/" + k)
        return ok
    if isinstance(ref, list):
        if not isinstance(dec, list) or len(dec) != len(ref):
            log(f"metadata {path}: list length")
            return False
        return all([check_meta(d, r, log, f"{path}[{i}]") for i, (d, r) in enumerate(zip(dec, ref))])
    if isinstance(ref, float):
        if not isinstance(dec, (int, float)) or abs(dec - ref) > 1e-4:
            log(f"metadata {path}: expected {ref}, found {dec}")
            return False
        return True
    if dec != ref:
        log(f"metadata {path}: expected {ref}, found {dec}")
        return False
    return True


def run_test(decoder_cmd, corpus_dir, test_id, work, log, env=None):
    tdir = os.path.join(corpus_dir, test_id)
    desc = json.load(open(os.path.join(tdir, "test.json")))
    desc.pop("sha256sums", None)
    inp = os.path.join(tdir, "input.jxl")
    prefix = os.path.join(work, "decoded")
    cmd = decoder_cmd + [inp, prefix + "_image.npy"]
    exact = []
    cmd_jpeg = None
    if "preview" in desc:
        cmd += ["--preview_out", os.path.join(work, "decoded_preview.npy")]
    if "reconstructed_jpeg" in desc:
        cmd_jpeg = decoder_cmd + [inp, os.path.join(work, "reconstructed.jpg")]
        exact.append(("reconstructed.jpg", os.path.join(work, "reconstructed.jpg")))
    if "original_icc" in desc:
        cmd += ["--orig_icc_out", os.path.join(work, "decoded_org.icc")]
        exact.append(("original.icc", os.path.join(work, "decoded_org.icc")))
    meta_fn = os.path.join(work, "meta.json")
    cmd += ["--metadata_out", meta_fn, "--icc_out", prefix + ".icc", "--norender_spotcolors"]
    for c in (cmd, cmd_jpeg):
        if c is None:
            continue
        r = subprocess.run(c, capture_output=True, text=True, env=env)
        if r.returncode != 0:
            log("decoder failed: %s\n%s" % (" ".join(c), r.stderr[-2000:]))
            return False, ""
    stderr = r.stderr
    ok = True
    for ref_name, got in exact:
        same = open(os.path.join(tdir, ref_name), "rb").read() == open(got, "rb").read()
        if not same:
            log(f"binary mismatch: {ref_name}")
        ok &= same
    ok &= check_meta(json.load(open(meta_fn)), desc, log)
    dec_icc = open(prefix + ".icc", "rb").read()
    ref_icc = open(os.path.join(tdir, "reference.icc"), "rb").read()
    if not os.path.exists(prefix + "_image.npy"):
        log("file not decoded: decoded_image.npy")
        return False, stderr
    ref = np.load(os.path.join(tdir, "reference_image.npy"))
    dec = np.load(prefix + "_image.npy")
    for i, fd in enumerate(desc["frames"]):
        ok &= compare_npy(ref, ref_icc, dec, dec_icc, i, fd["rms_error"], fd["peak_error"], log)
    if "preview" in desc:
        pfn = os.path.join(work, "decoded_preview.npy")
        if not os.path.exists(pfn):
            log("file not decoded: decoded_preview.npy")
            ok = False
        else:
            ok &= compare_npy(np.load(os.path.join(tdir, "reference_preview.npy")), ref_icc, np.load(pfn), dec_icc, 0,
                              desc["preview"]["rms_error"], desc["preview"]["peak_error"], log)
    return ok, stderr


def run_corpus(decoder, corpus, log=print, env=None):
    """Returns ({test_id: passed}, {test_id: decoder stderr})."""
    if os.path.isdir(corpus):
        cdir, txt = corpus, os.path.join(corpus, "corpus.txt")
    else:
        cdir, txt = os.path.dirname(corpus), corpus
    cmd = decoder.strip().split(" ")
    res, errs = {}, {}
    for tid in [l.strip() for l in open(txt) if l.strip()]:
        log(f"Testing {tid}")
        with tempfile.TemporaryDirectory(prefix=tid) as work:
            res[tid], errs[tid] = run_test(cmd, cdir, tid, work, log, env)
    log("%d of %d tests passed" % (sum(res.values()), len(res)))
    return res, errs


def generate(decoder, output, inputs, peak_error, rmse):
    """generator.py:21-98: the reference decoder's pixels, ICC and metadata become the expectations."""
    os.makedirs(output, exist_ok=True)
    ids = []
    for jxl in inputs:
        tid = os.path.basename(jxl).lower()
        tid = tid[:-4] if tid.endswith(".jxl") else tid
        base, n = tid, 2
        while tid in ids:
            tid = "%s%02d" % (base, n)
            n += 1
        ids.append(tid)
        tdir = os.path.join(output, tid)
        os.makedirs(tdir, exist_ok=True)
        shutil.copy(jxl, os.path.join(tdir, "input.jxl"))
        meta_fn = os.path.join(tdir, "test.json")
        subprocess.check_call(decoder.strip().split(" ") + [
            os.path.join(tdir, "input.jxl"), os.path.join(tdir, "reference_image.npy"), "--metadata_out", meta_fn,
            "--icc_out", os.path.join(tdir, "reference.icc")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        meta = json.load(open(meta_fn))
        for fr in meta["frames"]:
            fr["rms_error"], fr["peak_error"] = rmse, peak_error
        if "preview" in meta:
            meta["preview"]["rms_error"], meta["preview"]["peak_error"] = rmse, peak_error
        json.dump(meta, open(meta_fn, "w"), indent=2)
    open(os.path.join(output, "corpus.txt"), "w").write("".join(t + "\n" for t in ids))
    return ids


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    g = sub.add_parser("generate")
    g.add_argument("--decoder", required=True)
    g.add_argument("--output", required=True)
    g.add_argument("--peak_error", type=float, default=1e-4)
    g.add_argument("--rmse", type=float, default=2e-5)
    g.add_argument("inputs", nargs="+")
    r = sub.add_parser("run")
    r.add_argument("--decoder", required=True)
    r.add_argument("--corpus", required=True)
    a = ap.parse_args()
    if a.cmd == "generate":
        generate(a.decoder, a.output, a.inputs, a.peak_error, a.rmse)
        return 0
    res, _ = run_corpus(a.decoder, a.corpus)
    return 0 if all(res.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
