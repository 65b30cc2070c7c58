"""Group-row stripe decomposition of a VarDCT frame across ranks (one process
per GPU) and the halo hand-off between the two decode phases.

AC groups are independent through the inverse transforms; the loop filters
couple neighbouring stripes through LoopFilter::Padding() rows
(lib/jxl/loop_filter.h:26-29).  The exchange is point-to-point with the two
neighbours only (xGMI is point-to-point: no ring, no all-reduce); tensors may be
CUDA (backend nccl = RCCL) or CPU (backend gloo, used by the CPU tests).
"""
import torch
import torch.distributed as dist


def stripe_partition(ysize, world):
    """Contiguous group-row stripes: returns [(group_y0, group_rows)] * world.
    Every rank gets floor or ceil of ysg/world rows (the first `rem` get one
    more); requires ysg >= world."""
    ysg = (ysize + 255) // 256
    if ysg < world:
        raise ValueError(f"{ysg} group rows cannot be split over {world} ranks")
    base, rem = divmod(ysg, world)
    out, g0 = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((g0, n))
        g0 += n
    return out


def stripe_pixel_rows(ysize, g0, gr):
    return g0 * 256, min(ysize, (g0 + gr) * 256)


def exchange_halos(send_up, send_down, recv_up, recv_down, rank, world, group=None):
    """send_* / recv_*: tensors [3, halo, width] (any strides).  Rank r sends its
    first rows to r-1 and its last rows to r+1 and receives theirs.  Edge ranks
    skip the missing neighbour.  Returns after the received rows are in place."""
    if world == 1 or send_up.shape[1] == 0:
        return
    ops, unpack = [], []
    if rank > 0:
        s = send_up.contiguous()
        r = torch.empty_like(s)
        ops += [dist.P2POp(dist.isend, s, rank - 1, group), dist.P2POp(dist.irecv, r, rank - 1, group)]
        unpack.append((recv_up, r))
    if rank + 1 < world:
        s = send_down.contiguous()
        r = torch.empty_like(s)
        ops += [dist.P2POp(dist.isend, s, rank + 1, group), dist.P2POp(dist.irecv, r, rank + 1, group)]
        unpack.append((recv_down, r))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for dst, src in unpack:
        dst.copy_(src)


def gather_stripes(stripe, rows_per_rank, rank, world, dst=0, group=None):
    """Collects the output stripes ([rows, width, 3] float) on `dst` with
    point-to-point receives (each stripe travels over its own xGMI link).
    Returns the full frame on dst, None elsewhere."""
    if world == 1:
        return stripe
    if rank == dst:
        parts = [None] * world
        parts[dst] = stripe
        ops = []
        for r in range(world):
            if r == dst:
                continue
            parts[r] = torch.empty((rows_per_rank[r],) + tuple(stripe.shape[1:]),
                                   dtype=stripe.dtype, device=stripe.device)
            ops.append(dist.P2POp(dist.irecv, parts[r], r, group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return torch.cat(parts, dim=0)
    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, stripe.contiguous(), dst, group)]):
        req.wait()
    return None


class _HostStaged:
    """Point-to-point transfers of device tensors over a backend that moves host memory only (gloo): the sends are
    copied to the host when posted, the receives copied back by wait().  Only for smoke-testing the N > 1 flow on a box
    with one GPU (bench.py JXLHIP_BENCH_BACKEND=gloo); with RCCL the device tensors travel as they are."""

    def __init__(self, sends, recvs, group):
        if sends and sends[0][0].is_cuda:
            torch.cuda.synchronize()
        self.recvs = [(t, torch.empty(t.shape, dtype=t.dtype), peer) for t, peer in recvs]
        ops = [dist.P2POp(dist.isend, t.detach().to("cpu").contiguous(), peer, group) for t, peer in sends]
        ops += [dist.P2POp(dist.irecv, h, peer, group) for _, h, peer in self.recvs]
        self.reqs = dist.batch_isend_irecv(ops) if ops else []

    def wait(self):
        for r in self.reqs:
            r.wait()
        for t, h, _ in self.recvs:
            t.copy_(h)


def _post(sends, recvs, group, staged):
    """sends / recvs: [(tensor, peer)].  Returns an object with wait()."""
    if staged:
        return _HostStaged(sends, recvs, group)
    ops = [dist.P2POp(dist.isend, t, peer, group) for t, peer in sends]
    ops += [dist.P2POp(dist.irecv, t, peer, group) for t, peer in recvs]
    reqs = dist.batch_isend_irecv(ops) if ops else []

    class _Reqs:
        def wait(self):
            for r in reqs:
                r.wait()
    return _Reqs()


class StripeDecoder:
    """VarDctDecoder for this rank's stripe + the halo exchange."""

    def __init__(self, decoder, params, rank, world, group=None):
        self.dec, self.rank, self.world, self.group = decoder, rank, world, group
        self.parts = stripe_partition(params["ysize"], world)
        g0, gr = self.parts[rank]
        self.params = dict(params, stripe_group_y0=g0, stripe_group_rows=gr)
        self.rows = [stripe_pixel_rows(params["ysize"], a, b) for a, b in self.parts]
        decoder.begin_frame(self.params)
        self._halo = None  # persistent send / receive buffers of the halo exchange
        self._gather_pending = []  # decode_gathered: (step, request) of the transfers still in flight
        self._gather_step = 0
        # device tensors over a host-only backend (a one-GPU smoke test): staged through the host
        self.staged = bool(world > 1 and dist.is_initialized() and dist.get_backend(group) == "gloo" and
                           getattr(decoder, "tensor_device", "cuda") != "cpu")
        import os
        self.interior_first = os.environ.get("JXLHIP_STRIPES_INTERIOR_FIRST", "1") != "0"

    def _halo_buffers(self):
        if self._halo is None:
            d, h = self.dec, self.dec.halo_rows()
            shape = (3, h, self.params["xsize"])
            dev = getattr(d, "tensor_device", None) or f"cuda:{d.device}"  # (a CPU stand-in decoder in the gloo tests)
            mk = lambda: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
            self._halo = dict(up_send=mk(), dn_send=mk(), up_recv=mk(), dn_recv=mk())
        return self._halo

    # -- gather: the final collective of the 16K configuration ------------------
    def alloc_gather(self, stripe):
        """Rank 0: the whole-frame output buffer the stripes are received into (each stripe is a
        contiguous row range of it: no staging, no concatenation); None elsewhere."""
        if self.rank != 0:
            return None
        rows = sum(b - a for a, b in self.rows)
        return torch.empty((rows,) + tuple(stripe.shape[1:]), dtype=stripe.dtype, device=stripe.device)

    def gather(self, stripe, full):
        """One point-to-point transfer per stripe straight into rank 0's frame (xGMI is
        point-to-point: every stripe travels over its own link, no ring)."""
        if self.world == 1:
            return stripe
        if self.rank == 0:
            a, b = self.rows[0]
            full[a:b].copy_(stripe)
            req = _post([], [(full[self.rows[r][0]:self.rows[r][1]], r) for r in range(1, self.world)], self.group, self.staged)
        else:
            req = _post([(stripe, 0)], [], self.group, self.staged)
        req.wait()
        return full

    # -- the same gather, STREAMED with the step (round 6) ------------------------------------------------
    # gather() above starts when the whole stripe is done and makes the compute stream wait for it: a step costs
    # kernels + gather.  decode_gathered() posts a stripe's rows to rank 0 as soon as the launches that write them are
    # queued -- the interior rows (all but the two boundary block rows) while the halo rows still travel, the boundary rows
    # behind stripe_finish -- and does NOT wait for them: the caller alternates between two stripe buffers, so that the
    # rows of frame k travel while frame k + 1 is decoded, and a step costs max(kernels, gather).  What bounds it then is
    # rank 0's incoming links (7/8 of the frame over 7 point-to-point xGMI links), not the sum.
    def _post_rows(self, out, full, ya, yb):
        """Rows [ya, yb) of this rank's stripe (frame coordinates) to rank 0's frame; rank 0 posts the matching receives
        for the SAME chunk of every peer (chunk k of peer r = the k-th call of peer r: per-peer order is what pairs
        them).  Returns the request object (wait() = the current stream waits), or None when there is nothing to move."""
        y0 = self.rows[self.rank][0]
        if self.rank != 0:
            if yb <= ya:
                return None
            return _post([(out[ya - y0:yb - y0], 0)], [], self.group, self.staged)
        if yb > ya:
            full[ya:yb].copy_(out[ya - y0:yb - y0], non_blocking=True)
        return None

    def _post_peer_receives(self, full, which):
        """Rank 0: receives for chunk `which` (0 interior, 1 top boundary rows, 2 bottom boundary rows) of every peer."""
        recvs = []
        for r in range(1, self.world):
            a, b = self._chunk_rows(r, which)
            if b > a:
                recvs.append((full[a:b], r))
        return _post([], recvs, self.group, self.staged) if recvs else None

    def _chunk_rows(self, r, which):
        """Frame rows of chunk `which` of rank r's stripe -- the split decode() makes: interior = all but the block rows
        next to a neighbour, top / bottom = those block rows (empty on the frame's outer sides)."""
        y0, y1 = self.rows[r]
        ya = y0 + 8 if r > 0 else y0
        yb = y1 - 8 if r + 1 < self.world else y1
        if not self._splits(r):
            return (y0, y1) if which == 0 else (y0, y0)
        return [(ya, yb), (y0, ya), (yb, y1)][which]

    def _splits(self, r):
        y0, y1 = self.rows[r]
        ya = y0 + 8 if r > 0 else y0
        yb = y1 - 8 if r + 1 < self.world else y1
        return bool(self.interior_first and yb - ya >= 8 and self.dec.params.lf.epf_iters < 3 and self.dec.halo_rows() > 0)

    def wait_gather(self, older_than=None):
        """The current stream waits for the streamed-gather transfers posted before step `older_than` (None: for all of
        them -- the end of a run, or before the frame is read on rank 0)."""
        keep = []
        for step, req in self._gather_pending:
            if older_than is None or step < older_than:
                req.wait()
            else:
                keep.append((step, req))
        self._gather_pending = keep

    def decode_gathered(self, out, full):
        """decode(out) with the rows streamed into rank 0's `full` as they are produced (see above).  The transfers are
        left in flight across ONE following step: pass a different stripe buffer than at the step before (two buffers
        alternating); before a buffer is written again the transfers of the step that used it have been waited for
        here.  wait_gather() before the frame is read / at the end of a run."""
        if self.world == 1:
            return self.dec.decode_frame(out)
        step = self._gather_step
        self._gather_step += 1
        self.wait_gather(older_than=step - 1)  # everything of step - 2 and before (the last user of this buffer)

        def chunk(which):
            a, b = self._chunk_rows(self.rank, which)
            req = self._post_rows(out, full, a, b)
            if req is not None:
                self._gather_pending.append((step, req))
            if self.rank == 0:
                r0 = self._post_peer_receives(full, which)
                if r0 is not None:
                    self._gather_pending.append((step, r0))

        self.decode(out, on_interior=lambda: chunk(0))
        chunk(1)
        chunk(2)
        return out

    def decode(self, out, timing=None, on_interior=None):
        """Both phases of this rank's stripe.  timing: an optional dict that receives torch.cuda.Event pairs around the
        phases ("blocks", "interior", "halo_wait", "boundary"), for bench.py's per-phase report.  on_interior: called once
        the interior rows' launches are queued (decode_gathered posts their transfer there); when the stripe is not split
        it is called at the end, with every row queued."""
        d = self.dec
        if self.world == 1:
            # no neighbours: both phases in one call, walked in bands so that the
            # XYB planes of a band are filtered while still cache resident
            return d.decode_frame(out)

        def mark(name):
            if timing is not None:
                if torch.cuda.is_available():
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                else:  # (the gloo tests: host time stamps)
                    import time
                    ev = time.perf_counter()
                timing.setdefault(name, []).append(ev)

        h = d.halo_rows()
        up, dn = self.rank > 0, self.rank + 1 < self.world
        # the boundary rows leave phase 1's planes for dense buffers that live as long as the decoder and travel to
        # the two neighbours point to point (RCCL over the direct xGMI link; requests are ordered on streams: wait()
        # makes the compute stream wait, not the host)
        b = self._halo_buffers() if h else None
        fast = hasattr(d, "stripe_begin")  # three calls into the library per frame instead of up to nine (also under `timing`:
                                           # the per-phase report describes the sequence that is timed)
        mark("t0")
        if fast:
            d.stripe_begin(b["up_send"] if h and up else None, b["dn_send"] if h and dn else None)
        else:
            d.decode_blocks()
        mark("blocks")
        if h == 0:
            d.decode_filters(out)
            mark("interior")
            if on_interior:
                on_interior()
            return out
        sends, recvs = [], []
        if up:
            if not fast:
                d.halo_export(0, b["up_send"])
            sends.append((b["up_send"], self.rank - 1))
            recvs.append((b["up_recv"], self.rank - 1))
        if dn:
            if not fast:
                d.halo_export(1, b["dn_send"])
            sends.append((b["dn_send"], self.rank + 1))
            recvs.append((b["dn_recv"], self.rank + 1))
        # (send up / receive from up / send down / receive from down: the order the ranks' batches pair up in)
        req = _post(sends, recvs, self.group, self.staged)
        # INTERIOR FIRST: while the halo messages fly, filter the rows whose support stays inside the stripe -- all
        # but the first / last block row next to a neighbour (8 rows >= LoopFilter::Padding(), loop_filter.h:26-29;
        # the reference overlaps its neighbour hand-off as well, dec_group_border.cc:68-187).  (Enqueued AFTER the
        # sends are posted: the transfer waits for what is on the compute stream at that moment.)
        y0, y1 = self.rows[self.rank]
        ya = y0 + 8 if up else y0
        yb = y1 - 8 if dn else y1
        split = self.interior_first and yb - ya >= 8 and d.params.lf.epf_iters < 3
        if split:
            d.decode_filters(out, rows=(ya, yb))
            if on_interior:
                on_interior()
        mark("interior")
        req.wait()
        mark("halo_wait")
        if fast:
            d.stripe_finish(out, b["up_recv"] if up else None, b["dn_recv"] if dn else None, (ya, yb) if split else None)
        else:
            if up:
                d.halo_import(0, b["up_recv"])
            if dn:
                d.halo_import(1, b["dn_recv"])
            if split:
                d.decode_filters(out, rows=(y0, ya))
                d.decode_filters(out, rows=(yb, y1))
            else:
                d.decode_filters(out)
        mark("boundary")
        if on_interior and not split:
            on_interior()
        return out
