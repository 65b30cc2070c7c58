"""N>1 path on CPU: world_size-2 gloo run of the product's stripe partition,
halo exchange and gather (libjxl_amd/stripes.py).  The per-stripe compute is
done by the CPU oracle here (no GPU in this test); what is under test is that
"decode your groups, swap LoopFilter::Padding() rows with the neighbours,
filter your rows, gather" reproduces the whole-frame result bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, xs, ys, result_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    import frames
    import oracle as O
    from libjxl_amd import stripes, synth
    params, t, fr = frames.make_case(xs, ys, mix=synth.MIX_D1, gab=True, epf_iters=1, seed=31)
    parts = stripes.stripe_partition(ys, world)
    g0, gr = parts[rank]
    y0, y1 = stripes.stripe_pixel_rows(ys, g0, gr)
    halo = 3  # gab 1 + epf1 2
    xsb, ysb = fr.dims
    xsg = (xs + 255) // 256
    L = O.lib()
    # phase 1 for this rank's groups only
    planes = [np.zeros((ysb * 8, xsb * 8), np.float32) for _ in range(3)]
    rc = L.jxo_decode_groups(C.byref(fr.c), O._p3(planes), xsb * 8, g0 * xsg, (g0 + gr) * xsg)
    assert rc == 0
    tp = [torch.from_numpy(p) for p in planes]

    def rows(a, b):
        return torch.stack([p[a:b, :xs] for p in tp])  # [3, rows, xs] (a view per plane)

    # product code under test: halo exchange on strided views
    up_send, dn_send = rows(y0, y0 + halo), rows(y1 - halo, y1)
    up_recv = torch.zeros_like(up_send)
    dn_recv = torch.zeros_like(dn_send)
    stripes.exchange_halos(up_send, dn_send, up_recv, dn_recv, rank, world)
    if rank > 0:
        for c in range(3):
            tp[c][y0 - halo:y0, :xs] = up_recv[c]
    if rank + 1 < world:
        for c in range(3):
            tp[c][y1:y1 + halo, :xs] = dn_recv[c]
    # phase 2 on own rows (+2 rows of Gaborish output for EPF1)
    sigma = fr.compute_sigma()
    gab = [np.zeros_like(p) for p in planes]
    a, b = max(0, y0 - 2), min(ys, y1 + 2)
    L.jxo_gaborish(C.byref(fr.c), O._p3(planes), O._p3(gab), planes[0].shape[1], a, b)
    epf = [np.zeros_like(p) for p in planes]
    L.jxo_epf(C.byref(fr.c), 1, O._p(sigma), O._p3(gab), O._p3(epf), planes[0].shape[1], y0, y1)
    rgb = np.zeros((ys, xs, 3), np.float32)
    L.jxo_xyb_to_linear_rgb(C.byref(fr.c), O._p3(epf), planes[0].shape[1], O._p(rgb), xs * 3, y0, y1)
    stripe = torch.from_numpy(rgb[y0:y1].copy())
    rows_per_rank = [stripes.stripe_pixel_rows(ys, *p)[1] - stripes.stripe_pixel_rows(ys, *p)[0]
                     for p in parts]
    full = stripes.gather_stripes(stripe, rows_per_rank, rank, world)
    if rank == 0:
        ref = fr.decode(threads=2)
        ok = bool(np.array_equal(full.numpy(), ref))
        open(result_path, "w").write("ok" if ok else "mismatch %g" % np.abs(full.numpy() - ref).max())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


def test_stripe_partition():
    from libjxl_amd import stripes
    assert stripes.stripe_partition(8640, 8) == [(0, 5), (5, 5), (10, 4), (14, 4), (18, 4), (22, 4),
                                                 (26, 4), (30, 4)]
    assert stripes.stripe_partition(4320, 1) == [(0, 17)]
    parts = stripes.stripe_partition(4320 * 8, 8)
    assert sum(n for _, n in parts) == (4320 * 8 + 255) // 256
    assert all(b[0] == a[0] + a[1] for a, b in zip(parts, parts[1:]))
    with pytest.raises(ValueError):
        stripes.stripe_partition(300, 3)
    assert stripes.stripe_pixel_rows(700, 2, 1) == (512, 700)


def test_two_rank_stripes_halo_exchange_and_gather(tmp_path, oracle):
    port = 29500 + os.getpid() % 2000
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, port, 300, 600, str(result)), nprocs=2, join=True)
    assert result.read_text() == "ok"


class _RecordingDecoder:
    """Stands in for VarDctDecoder on the CPU: records the call sequence of StripeDecoder.decode and moves recognisable
    rows through the halo buffers."""
    tensor_device = "cpu"
    device = 0

    def __init__(self, rank, xs, epf=1):
        import types
        self.rank, self.xs, self.calls = rank, xs, []
        self.params = types.SimpleNamespace(lf=types.SimpleNamespace(epf_iters=epf), xsize=xs)
        self.imported = {}

    def begin_frame(self, params):
        self.frame = params

    def halo_rows(self):
        return 3

    def decode_blocks(self):
        self.calls.append(("blocks",))

    def halo_export(self, which, buf):
        self.calls.append(("export", which))
        buf.fill_(float(10 * self.rank + which))
        return buf

    def halo_import(self, which, buf):
        self.calls.append(("import", which))
        self.imported[which] = float(buf[0, 0, 0])

    def decode_filters(self, out, rows=None):
        self.calls.append(("filters", rows))


class _RecordingDecoderFast(_RecordingDecoder):
    """... with the three-call form of round 5 (jxlhip_stripe_begin / _finish): the same sequence, recorded as the C side
    runs it (context.hip)."""

    def stripe_begin(self, send_up=None, send_down=None):
        self.calls.append(("blocks",))
        if send_up is not None:
            self.halo_export(0, send_up)
        if send_down is not None:
            self.halo_export(1, send_down)

    def stripe_finish(self, out, recv_up=None, recv_down=None, interior=None):
        if recv_up is not None:
            self.halo_import(0, recv_up)
        if recv_down is not None:
            self.halo_import(1, recv_down)
        y0, y1 = self.frame["stripe_group_y0"] * 256, min(self.frame["ysize"], (self.frame["stripe_group_y0"] + self.frame["stripe_group_rows"]) * 256)
        if interior is None:
            self.decode_filters(out)
        else:
            self.decode_filters(out, rows=(y0, interior[0]))
            self.decode_filters(out, rows=(interior[1], y1))


def _flow_worker(rank, world, port, result_dir, interior_first, fast=False, timed=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["JXLHIP_STRIPES_INTERIOR_FIRST"] = "1" if interior_first else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libjxl_amd import stripes
    xs, ys = 300, 256 * 7
    d = (_RecordingDecoderFast if fast else _RecordingDecoder)(rank, xs)
    sd = stripes.StripeDecoder(d, dict(xsize=xs, ysize=ys), rank, world)
    timing = {} if timed else None
    sd.decode(torch.zeros(1), timing=timing)
    if timed:  # bench.py's per-phase pass: one stamp per phase, in this order, through the split calls
        order = ["t0", "blocks", "interior", "halo_wait", "boundary"]
        stamps = [timing[k][0] for k in order]
        assert list(timing) == order and stamps == sorted(stamps), timing
    y0, y1 = sd.rows[rank]
    up, dn = rank > 0, rank + 1 < world
    filt = [c[1] for c in d.calls if c[0] == "filters"]
    ok = d.calls[0] == ("blocks",)
    first_import = min([i for i, c in enumerate(d.calls) if c[0] == "import"], default=len(d.calls))
    first_filter = min(i for i, c in enumerate(d.calls) if c[0] == "filters")
    if interior_first:
        ya, yb = (y0 + 8 if up else y0), (y1 - 8 if dn else y1)
        ok &= filt == [(ya, yb), (y0, ya), (yb, y1)] and first_filter < first_import  # the interior rows go first
    else:
        ok &= filt == [None] and first_import < first_filter
    # the rows that arrived are the neighbours': from above its "down" export (which = 1), from below its "up" export
    ok &= d.imported.get(0, None) == (10.0 * (rank - 1) + 1 if up else None)
    ok &= d.imported.get(1, None) == (10.0 * (rank + 1) + 0 if dn else None)
    open(os.path.join(result_dir, "r%d" % rank), "w").write("ok" if ok else "bad: %r" % (d.calls,))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("interior_first", [True, False])
def test_stripe_decoder_call_sequence_over_gloo(tmp_path, interior_first):
    """libjxl_amd.stripes.StripeDecoder.decode -- what `bench.py --gpus N` runs on every rank -- with a recording stand-in
    for the decoder, three ranks over gloo: phase 1, both exports, the interior rows BEFORE the halo rows are imported,
    then the two boundary block rows (JXLHIP_STRIPES_INTERIOR_FIRST=0: round 3's order), and each rank receives its
    neighbours' rows."""
    port = 29500 + (os.getpid() + 7 + int(interior_first)) % 2000
    mp.spawn(_flow_worker, args=(3, port, str(tmp_path), interior_first), nprocs=3, join=True)
    for r in range(3):
        assert (tmp_path / ("r%d" % r)).read_text() == "ok", (tmp_path / ("r%d" % r)).read_text()


@pytest.mark.parametrize("fast,timed", [(True, False), (True, True), (False, True)])
def test_stripe_decoder_three_call_form_and_timed_pass_over_gloo(tmp_path, fast, timed):
    """The three-call form (stripe_begin, the interior rows, stripe_finish: round 5) issues the same sequence as the
    split calls -- exports before the sends, the interior rows before the imports -- and bench.py's per-phase pass
    (timing = {}) takes the split calls and leaves one stamp per phase, in order."""
    port = 29500 + (os.getpid() + 23 + 2 * int(fast) + int(timed)) % 2000
    mp.spawn(_flow_worker, args=(3, port, str(tmp_path), True, fast, timed), nprocs=3, join=True)
    for r in range(3):
        assert (tmp_path / ("r%d" % r)).read_text() == "ok", (tmp_path / ("r%d" % r)).read_text()


class _PaintingDecoder(_RecordingDecoderFast):
    """... whose decode_filters really writes the rows it is asked for: row y of frame number n becomes 1000 n + y, so
    that a gathered frame says which step and which row every value came from."""
    frame_no = 0

    def decode_filters(self, out, rows=None):
        super().decode_filters(out, rows)
        y0 = self.frame["stripe_group_y0"] * 256
        y1 = min(self.frame["ysize"], (self.frame["stripe_group_y0"] + self.frame["stripe_group_rows"]) * 256)
        a, b = rows if rows is not None else (y0, y1)
        if b > a:
            out[a - y0:b - y0] = (1000.0 * self.frame_no + torch.arange(a, b, dtype=torch.float32)).view(-1, 1, 1)


def _gather_worker(rank, world, port, result_dir, interior_first, epf):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["JXLHIP_STRIPES_INTERIOR_FIRST"] = "1" if interior_first else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libjxl_amd import stripes
    xs, ys = 40, 256 * 7 + 100
    d = _PaintingDecoder(rank, xs, epf=epf)
    sd = stripes.StripeDecoder(d, dict(xsize=xs, ysize=ys), rank, world)
    y0, y1 = sd.rows[rank]
    outs = [torch.zeros((y1 - y0, xs, 3)), torch.zeros((y1 - y0, xs, 3))]
    full = sd.alloc_gather(outs[0])
    ok = True
    for n in range(1, 6):  # five frames back to back, two stripe buffers alternating, nothing waited for in between
        d.frame_no = n
        sd.decode_gathered(outs[n & 1], full)
        if n == 3:  # a reader in the middle of the run: everything posted so far must have landed
            sd.wait_gather()
            if rank == 0:
                want = (1000.0 * n + torch.arange(ys, dtype=torch.float32)).view(-1, 1, 1).expand(ys, xs, 3)
                ok &= bool(torch.equal(full, want))
    sd.wait_gather()
    if rank == 0:
        want = (1000.0 * 5 + torch.arange(ys, dtype=torch.float32)).view(-1, 1, 1).expand(ys, xs, 3)
        ok &= bool(torch.equal(full, want))
        ok &= len(sd._gather_pending) == 0
    # the interior rows were posted BEFORE the halo rows were imported (the transfer starts while the exchange runs)
    if interior_first and epf < 3:
        first_import = min([i for i, c in enumerate(d.calls) if c[0] == "import"], default=len(d.calls))
        first_filter = min(i for i, c in enumerate(d.calls) if c[0] == "filters")
        ok &= first_filter < first_import
    open(os.path.join(result_dir, "r%d" % rank), "w").write("ok" if ok else "bad")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("interior_first,epf", [(True, 1), (False, 1), (True, 3)])
def test_streamed_gather_over_gloo(tmp_path, interior_first, epf):
    """StripeDecoder.decode_gathered (round 6): every rank posts its stripe's rows to rank 0 as soon as their launches
    are queued -- interior rows first, then the boundary block rows -- and leaves the transfers in flight across the next
    step (two stripe buffers alternating).  Three ranks over gloo, five frames back to back: rank 0's frame holds exactly
    the rows of the LAST frame of every rank at the end, and of frame 3 when a reader waits in the middle; unsplit
    stripes (interior-first off, or epf_iters = 3) travel as one chunk."""
    port = 29500 + (os.getpid() + 41 + 2 * int(interior_first) + epf) % 2000
    mp.spawn(_gather_worker, args=(3, port, str(tmp_path), interior_first, epf), nprocs=3, join=True)
    for r in range(3):
        assert (tmp_path / ("r%d" % r)).read_text() == "ok"
