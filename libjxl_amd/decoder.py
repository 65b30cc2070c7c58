"""Host-side mirror of the reference's per-frame decode state for the VarDCT
back-end (PassesDecoderState + RenderPipeline, lib/jxl/dec_cache.h:86-229,
lib/jxl/render_pipeline/render_pipeline.h:139-152) on top of the C ABI.

PyTorch is plumbing only: it owns the HBM tensors and the stream; every
computation happens in libjxl_hip.so's HIP kernels.
"""
import ctypes as C

import torch

from . import abi


def _check(L, ctx, rc, what):
    if rc != 0:
        msg = L.jxlhip_last_error(ctx) if ctx else b""
        raise abi.JxlHipError(
            f"{what}: {L.jxlhip_status_string(rc).decode()} ({rc}) {msg.decode() if msg else ''}")


class VarDctDecoder:
    """One decoder context bound to one GPU (one per process / rank)."""

    def __init__(self, device=0, use_torch_stream=True):
        self.L = abi.load_library()
        self.device = int(device)
        if not torch.cuda.is_available():
            raise abi.JxlHipError("no HIP device visible: the VarDCT back-end has no CPU path")
        self.ctx = C.c_void_p()
        _check(self.L, None, self.L.jxlhip_create(self.device, C.byref(self.ctx)), "jxlhip_create")
        if use_torch_stream:
            s = torch.cuda.current_stream(self.device).cuda_stream
            _check(self.L, self.ctx, self.L.jxlhip_set_stream(self.ctx, C.c_void_p(s), 1), "set_stream")
        self.params = None
        self._keep = None
        self.out = None

    def close(self):
        if self.ctx:
            self.L.jxlhip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- frame set-up ------------------------------------------------------
    def begin_frame(self, params):
        """params: abi.FrameParams (or dict from synth.synth_frame)."""
        if isinstance(params, dict):
            params = abi.make_params(params)
        self.params = params
        _check(self.L, self.ctx, self.L.jxlhip_frame_begin(self.ctx, C.byref(params)), "frame_begin")

    def default_dequant_tables(self):
        t = torch.empty(2056 * 64 * 3, dtype=torch.float32, device=f"cuda:{self.device}")
        _check(self.L, self.ctx, self.L.jxlhip_default_dequant_tables(self.ctx, C.c_void_p(t.data_ptr())), "default_dequant_tables")
        return t

    def dequant_tables(self, encodings=None):
        """The 17 dequant tables of abi.QuantEncodings (None = the default library)."""
        t = torch.empty(2056 * 64 * 3, dtype=torch.float32, device=f"cuda:{self.device}")
        e = None if encodings is None else C.cast(C.byref(encodings), C.c_void_p)
        _check(self.L, self.ctx, self.L.jxlhip_dequant_tables(self.ctx, e, C.c_void_p(t.data_ptr())), "dequant_tables")
        return t

    def set_inputs(self, tensors, dequant_table):
        """tensors: dict of CUDA tensors laid out as jxlhip_frame_inputs."""
        dev = f"cuda:{self.device}"
        for k, v in tensors.items():
            for t in (v if isinstance(v, (list, tuple)) else [v]):
                assert t.is_cuda and t.is_contiguous(), k
        fi = abi.FrameInputs()
        for c in range(3):
            fi.coeffs[c] = tensors["coeffs"][c].data_ptr()
            fi.dc[c] = tensors["dc"][c].data_ptr()
        fi.ac_strategy = tensors["ac_strategy"].data_ptr()
        fi.raw_quant = tensors["raw_quant"].data_ptr()
        fi.epf_sharpness = tensors["epf_sharpness"].data_ptr()
        fi.ytox_map = tensors["ytox_map"].data_ptr()
        fi.ytob_map = tensors["ytob_map"].data_ptr()
        fi.dequant_table = dequant_table.data_ptr()
        self._keep = (tensors, dequant_table, dev)
        _check(self.L, self.ctx, self.L.jxlhip_frame_set_inputs(self.ctx, C.byref(fi)), "frame_set_inputs")

    # -- geometry ------------------------------------------------------------
    def stripe_rows(self):
        p = self.params
        ysg = (p.ysize + 255) // 256
        g0 = p.stripe_group_y0
        gr = p.stripe_group_rows if p.stripe_group_rows else ysg - g0
        y0 = g0 * 256
        y1 = min(p.ysize, (g0 + gr) * 256)
        return y0, y1

    def alloc_output(self):
        p = self.params
        y0, y1 = self.stripe_rows()
        dev = f"cuda:{self.device}"
        # undo_orientation 5..8: the display frame is ysize pixels wide and xsize rows high
        oh, ow = (p.xsize, p.ysize) if p.undo_orientation >= 5 else (y1 - y0, p.xsize)
        if p.output_kind == 1:
            return torch.empty((oh, ow, 3), dtype=torch.float32, device=dev)
        if p.output_kind == 2:  # packed RGB(A): dtype of the sample type (F16 as raw uint16 bits)
            dt = {0: torch.float32, 1: torch.uint8, 2: torch.int16, 3: torch.int16}[p.out_format.sample_type]
            return torch.empty((oh, ow, p.out_format.num_channels), dtype=dt, device=dev)
        return torch.empty((3, y1 - y0, p.xsize), dtype=torch.float32, device=dev)

    def _out_args(self, out):
        p = self.params
        if p.output_kind == 1:
            return C.c_void_p(out.data_ptr()), out.stride(0) * 4, 0
        if p.output_kind == 2:
            return C.c_void_p(out.data_ptr()), out.stride(0) * out.element_size(), 0
        return C.c_void_p(out.data_ptr()), out.stride(1), out.stride(0)

    def set_alpha(self, plane):
        """The frame's alpha channel for 4-channel packed outputs: a host float32 array [ysize, xsize] (1.0 = opaque),
        jxlhip_set_alpha; begin_frame resets to opaque."""
        import numpy as np
        a = np.ascontiguousarray(plane, dtype=np.float32)
        assert a.shape == (self.params.ysize, self.params.xsize), a.shape
        _check(self.L, self.ctx, self.L.jxlhip_set_alpha(self.ctx, a.ctypes.data, a.shape[1]), "set_alpha")
        self.sync()  # (the array may go away as soon as this returns)

    # -- decode ----------------------------------------------------------------
    def decode_blocks(self):
        _check(self.L, self.ctx, self.L.jxlhip_decode_blocks(self.ctx), "decode_blocks")

    def decode_filters(self, out, rows=None):
        """Phase 2 into `out` (the stripe's rows).  rows = (y_begin, y_end): only those frame rows (block-row multiples
        inside the stripe, jxlhip_decode_filters_rows) -- the interior first while the halo rows travel."""
        a = self._out_args(out)
        if rows is None:
            _check(self.L, self.ctx, self.L.jxlhip_decode_filters(self.ctx, *a), "decode_filters")
        else:
            _check(self.L, self.ctx, self.L.jxlhip_decode_filters_rows(self.ctx, *a, int(rows[0]), int(rows[1])), "decode_filters_rows")

    def decode_frame(self, out=None):
        if out is None:
            out = self.alloc_output()
        a = self._out_args(out)
        _check(self.L, self.ctx, self.L.jxlhip_decode_frame(self.ctx, *a), "decode_frame")
        return out

    def sync(self):
        _check(self.L, self.ctx, self.L.jxlhip_sync(self.ctx), "sync")

    # -- taps / profiling --------------------------------------------------------
    def export_xyb(self):
        """Row-major copies of the phase-1 XYB planes (the stripe's block-padded
        rows x padded width), as numpy arrays."""
        p = self.params
        xsb = (p.xsize + 7) // 8
        y0, y1 = self.stripe_rows()
        rows = (y1 - y0 + 7) // 8 * 8
        outs = [torch.empty((rows, xsb * 8), dtype=torch.float32, device=f"cuda:{self.device}")
                for _ in range(3)]
        ptrs = (C.c_void_p * 3)(*[o.data_ptr() for o in outs])
        _check(self.L, self.ctx, self.L.jxlhip_export_xyb(self.ctx, ptrs, xsb * 8), "export_xyb")
        self.sync()
        return [o.cpu().numpy() for o in outs]

    def sigma(self):
        ptr, stride = C.c_void_p(), C.c_size_t()
        _check(self.L, self.ctx, self.L.jxlhip_get_sigma(self.ctx, C.byref(ptr), C.byref(stride)), "get_sigma")
        p = self.params
        ysb = (p.ysize + 7) // 8
        return _as_tensor(ptr.value, ysb * stride.value, torch.float32, self.device).reshape(ysb, stride.value).clone()

    def halo_rows(self):
        return self.L.jxlhip_halo_rows(self.ctx)

    def halo_export(self, which, buf=None):
        """Dense [3, halo, xsize] tensor with this stripe's first (which=0) or
        last (which=1) halo rows (written into `buf` when given)."""
        p = self.params
        if buf is None:
            buf = torch.empty((3, self.halo_rows(), p.xsize), dtype=torch.float32,
                              device=f"cuda:{self.device}")
        _check(self.L, self.ctx, self.L.jxlhip_halo_export(self.ctx, which, C.c_void_p(buf.data_ptr())), "halo_export")
        return buf

    def halo_import(self, which, buf):
        """Installs rows received from the stripe above (which=0) / below (1)."""
        assert buf.is_cuda and buf.is_contiguous() and buf.dtype == torch.float32
        assert tuple(buf.shape) == (3, self.halo_rows(), self.params.xsize)
        _check(self.L, self.ctx, self.L.jxlhip_halo_import(self.ctx, which, C.c_void_p(buf.data_ptr())), "halo_import")

    def set_stream(self, stream):
        """All launches of this context go to `stream` (a torch.cuda.Stream) from now on."""
        _check(self.L, self.ctx, self.L.jxlhip_set_stream(self.ctx, C.c_void_p(stream.cuda_stream), 1), "set_stream")

    def stripe_begin(self, send_up=None, send_down=None):
        """Phase 1 of the stripe + its boundary rows into the dense [3, halo, xsize] send buffers (None = no neighbour
        on that side): one call (jxlhip_stripe_begin)."""
        a = C.c_void_p(send_up.data_ptr()) if send_up is not None else None
        b = C.c_void_p(send_down.data_ptr()) if send_down is not None else None
        _check(self.L, self.ctx, self.L.jxlhip_stripe_begin(self.ctx, a, b), "stripe_begin")

    def stripe_finish(self, out, recv_up=None, recv_down=None, interior=None):
        """The neighbours' rows installed + phase 2 of every row outside interior = (y_begin, y_end), which
        decode_filters(rows=interior) has filtered already (None: of every row): one call (jxlhip_stripe_finish)."""
        a = C.c_void_p(recv_up.data_ptr()) if recv_up is not None else None
        b = C.c_void_p(recv_down.data_ptr()) if recv_down is not None else None
        ya, yb = (int(interior[0]), int(interior[1])) if interior else (0, 0)
        _check(self.L, self.ctx, self.L.jxlhip_stripe_finish(self.ctx, a, b, *self._out_args(out), ya, yb), "stripe_finish")

    def set_concurrency_hint(self, frames_in_flight):
        """How many contexts the caller keeps busy on this device at a time (a pool of decoders): moves the frame size
        from which decode_frame takes the fused kernel (12 Mpx alone, 6 Mpx with several frames in flight)."""
        _check(self.L, self.ctx, self.L.jxlhip_set_concurrency_hint(self.ctx, int(frames_in_flight)), "set_concurrency_hint")

    def profile(self, enable=True):
        _check(self.L, self.ctx, self.L.jxlhip_profile_enable(self.ctx, int(enable)), "profile_enable")

    def profile_read(self):
        ms = (C.c_float * abi.KERNEL_COUNT)()
        n = (C.c_uint32 * abi.KERNEL_COUNT)()
        _check(self.L, self.ctx, self.L.jxlhip_profile_read(self.ctx, ms, n), "profile_read")
        return {abi.KERNEL_NAMES[i]: (ms[i], n[i]) for i in range(abi.KERNEL_COUNT) if n[i]}


class _CudaArray:
    """Minimal __cuda_array_interface__ holder for zero-copy torch views of
    context-owned HBM."""

    def __init__(self, ptr, nelem, typestr):
        self.__cuda_array_interface__ = {
            "shape": (nelem,), "typestr": typestr, "data": (int(ptr), False),
            "version": 2, "strides": None}


def _as_tensor(ptr, nelem, dtype, device):
    assert dtype == torch.float32
    return torch.as_tensor(_CudaArray(ptr, nelem, "<f4"), device=f"cuda:{device}")
