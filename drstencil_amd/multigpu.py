"""Slab decomposition (z slabs in 3D, y slabs in 2D) of a stencil run across the GPUs of one node, one process per
GPU, halo exchange by RCCL send/recv over xGMI (torch.distributed, backend "nccl" == RCCL)
overlapped with the interior sweep on a second HIP stream.

The reference is single-GPU (no cudaSetDevice / streams / NCCL), so
this is additive; its semantics are fixed by the single-domain run it must reproduce:
the global frozen ring stays on the outer faces of rank 0 and rank R-1, every launch
(src -> dst) is followed by an exchange of dst's H = step*order boundary planes with the
<= 2 neighbours, and both ping-pong buffers carry H ghost planes per interior face.

Per launch and rank (boundary first, so the exchange overlaps the interior sweep):
    main stream : boundary views (first/last owned planes; ONE launch of a --pair-launch kernel when the rank has two
                  neighbours) -> event b ; interior kernel ; wait c
    side stream : wait b; batch_isend_irecv of dst's boundary planes -> event c
Three things measured with scripts/rccl_probe.py (profiles/r01_rccl_probe.md) shape this:
  * the interior kernel is enqueued BEFORE the RCCL call is made -- batch_isend_irecv costs ~100 us of host time
    per call and the GPU idles through it otherwise;
  * the side stream and RCCL's own stream are high priority (init the process group with
    ProcessGroupNCCL.Options(is_high_priority_stream=True), see nccl_options()): HIP multiplexes streams of one
    priority onto a few in-order hardware queues, and an RCCL kernel that lands on the interior kernel's queue
    runs only after it;
  * the boundary views stay in front of the interior kernel on the same stream (~10 us): beside it they wait
    55 us each for the long-lived stream blocks of the interior kernel to release their CUs.
Critical path per launch = boundary + max(interior, exchange).  Each neighbour pair talks over its own xGMI link;
nothing is all-reduced in the loop.

With `every = 2` (SlabPlan) the ghost planes are 2H wide and only every second launch -- the one that writes A --
is followed by an exchange; the launch before it sweeps the whole local slab on the main stream.

Kernels are ordinary generated kernels: a z sub-range of a slab is a contiguous view, so
a "boundary kernel" is the generator's kernel for L = G + 2H planes (G = ghost width) and the interior kernel
the one for the remaining view -- no special device code (dr2_<name> is the same body over two buffer pairs).
"""
import os

import drstencil_amd as drs


def slab_bounds(L, world, rank):
    """Planes [z0, z1) owned by `rank` (balanced split of the outermost dim)."""
    return (rank * L) // world, ((rank + 1) * L) // world


class SlabPlan:
    """Index bookkeeping for one rank (pure Python; unit-tested on CPU).

    every = 1: ghost width G = H, every launch is followed by an exchange of its output's H boundary planes.
    every = 2: ghost width G = 2H and ONE exchange per ping-pong pair.  Launch 1 (A -> B) sweeps the whole local
    slab with no exchange: it also computes the inner H planes of each ghost region, so B is valid on [H, Lloc-H).
    Launch 2 (B -> A) produces exactly the owned planes and A's 2H boundary planes are exchanged: the same bytes per
    pair of launches in half as many messages, with one boundary/interior split per pair instead of two, for 2H
    redundantly computed planes per rank and pair (B's ghost planes are never exchanged or trusted)."""

    def __init__(self, L, H, world, rank, every=1):
        assert every in (1, 2)
        self.L, self.H, self.world, self.rank, self.every = L, H, world, rank, every
        G = self.G = every * H
        self.z0, self.z1 = slab_bounds(L, world, rank)
        self.has_up = rank > 0            # neighbour holding smaller z
        self.has_dn = rank < world - 1
        self.lo = self.z0 - (G if self.has_up else 0)   # first global plane held locally
        self.hi = self.z1 + (G if self.has_dn else 0)
        self.Lloc = self.hi - self.lo
        if world > 1 and self.z1 - self.z0 < 2 * G:
            raise ValueError("slab of %d planes is thinner than twice the ghost width %d" % (self.z1 - self.z0, G))
        # views (local plane ranges [a,b)) whose kernels write outputs [a+H, b-H); those of the exchanging launch:
        self.top = (G - H, 2 * G + H) if self.has_up else None                       # outputs = the planes sent up
        self.bot = (self.Lloc - 2 * G - H, self.Lloc - G + H) if self.has_dn else None
        a = 2 * G - H if self.has_up else 0
        b = self.Lloc - 2 * G + H if self.has_dn else self.Lloc
        self.interior = (a, b)
        # the launch without exchange (every = 2 only): the whole local slab
        self.full = (0, self.Lloc)
        # planes of dst to send / ghost planes to receive after an exchanging launch
        self.send_up = (G, 2 * G) if self.has_up else None
        self.recv_up = (0, G) if self.has_up else None
        self.send_dn = (self.Lloc - 2 * G, self.Lloc - G) if self.has_dn else None
        self.recv_dn = (self.Lloc - G, self.Lloc) if self.has_dn else None

    def pair_view(self):
        """Length of the two boundary views when the rank has both (they go out in ONE launch of a --pair-launch kernel)."""
        if self.top is not None and self.bot is not None and self.top[1] - self.top[0] == self.bot[1] - self.bot[0]:
            return self.top[1] - self.top[0]
        return None

    def views(self):
        """Every view length this rank launches a kernel on (for prebuilding)."""
        out = [v[1] - v[0] for v in (self.top, self.bot, self.interior) if v is not None and v[1] - v[0] > 2 * self.H]
        if self.every == 2 and self.world > 1:
            out.append(self.Lloc)
        return sorted(set(out))

    def outputs(self):
        """Global output planes the exchanging launch writes on this rank (for tests)."""
        out = []
        for v in (self.top, self.interior, self.bot):
            if v is not None and v[1] - v[0] > 2 * self.H:
                out.append((self.lo + v[0] + self.H, self.lo + v[1] - self.H))
        return out


def choose_exchange_every(planes_per_rank, plane_bytes, H, link_GBps=60.0, kernel_GBps=5200.0):
    """Launches per exchange for a slab run, from a two-number model (no multi-GPU box was available to measure it):
    one exchange per ping-pong pair (every = 2) sends 2H planes per face and must hide under ONE interior sweep, so it
    is chosen only when that sweep (2 * plane_bytes per plane at the slab kernels' ~5.2 TB/s) takes at least 1.3x the
    transfer (one xGMI link, ~60 GB/s per direction through RCCL send/recv); otherwise every launch exchanges H planes.
    C4 (4 MiB planes, H = 2): every = 2 up to 4 GPUs, 1 at 8."""
    sweep_us = planes_per_rank * 2.0 * plane_bytes / (kernel_GBps * 1e3)
    xfer_us = 2.0 * H * plane_bytes / (link_GBps * 1e3)
    return 2 if planes_per_rank >= 8 * H and sweep_us >= 1.3 * xfer_us else 1


def decide_exchange_every(sweep_us, x1_us, x2_us, b1_us, b2_us, planes, H):
    """The decision rule of measure_exchange_every on its (MAX-over-ranks) inputs: estimated time of one ping-pong pair with an exchange
    of H planes per face beside EACH launch's interior sweep (every = 1) against one exchange of 2H planes beside the second launch's
    sweep after a whole-slab launch that recomputes 2H ghost planes (every = 2).  A slow link (x2 > sweep) flips the choice to 1:
    the 2H-plane transfer no longer hides under one sweep, two H-plane transfers still hide under two.  Returns (every, pair1, pair2)."""
    pair1 = 2.0 * (b1_us + max(sweep_us, x1_us))
    pair2 = sweep_us * (1.0 + 2.0 * H / planes) + b2_us + max(sweep_us, x2_us)
    return (2 if planes >= 8 * H and pair2 < pair1 else 1), pair1, pair2


def measure_exchange_every(torch, dist, dims, H, rank, world, sweep, device, dtype, reps=6, self_neighbour=False):
    """The same decision as choose_exchange_every, from numbers measured on the machine the run is on (called during
    warm-up, before the slab buffers exist): one interior sweep of this rank's slab against one exchange of 2H planes per
    face (and of H planes) with the real neighbours through the real process group; MAX over ranks, so every rank
    decides alike (checked: the choices are gathered, and a disagreement -- NaNs, a rank with different code -- makes everybody
    take every = 1, the mode that needs nothing of the other ranks' layout): the mode with the shorter estimated ping-pong pair
    wins (decide_exchange_every).  On a CPU device (the gloo tests) the same measurements are taken with the wall clock.
    Returns (every, measurements); measurements["this_rank"] holds the rank's own numbers before the MAX."""
    import time as _time
    dims = tuple(dims)
    p = SlabPlan(dims[0], H, world, rank, 1)
    rest = dims[1:]
    gpu = getattr(device, "type", str(device)) == "cuda"
    main = torch.cuda.current_stream(device) if gpu else None
    handle = main.cuda_stream if gpu else 0
    src = torch.zeros((p.Lloc,) + rest, dtype=dtype, device=device)
    dst = torch.zeros_like(src)
    a, b = p.interior

    def timed(fn):
        for _ in range(2):
            fn()
        if not gpu:
            t0 = _time.perf_counter()
            for _ in range(reps):
                fn()
            return (_time.perf_counter() - t0) / reps * 1e6
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(reps):
            fn()
        e1.record(main)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    # a rank that owns exactly 2H planes has no interior view (SlabRun.launch skips it as well): nothing to time, and asking
    # for a kernel without interior would raise here while the other ranks wait in all_reduce
    sweep_us = timed(lambda: sweep(src[a:b], dst[a:b], handle)) if b - a > 2 * H else 0.0
    up = 0 if self_neighbour else rank - 1
    dn = 0 if self_neighbour else rank + 1
    has_up, has_dn = (p.has_up or self_neighbour), (p.has_dn or self_neighbour)

    def exchange_us(G):
        bufs = [torch.zeros((G,) + rest, dtype=dtype, device=device) for _ in range(4)]

        def exchange():
            ops = []
            if has_up:
                ops += [("send", bufs[0], up), ("recv", bufs[1], dn if self_neighbour else up)]
            if has_dn:
                ops += [("send", bufs[2], dn), ("recv", bufs[3], up if self_neighbour else dn)]
            batch_p2p(torch, dist, ops)
        return timed(exchange) if (has_up or has_dn) else 0.0

    x1, x2 = exchange_us(H), exchange_us(2 * H)

    def boundary_us(every):
        """The boundary launch(es) of an exchanging launch (both views in one dr2_ launch where the rank has two neighbours) plus the
        event the side stream waits for -- what round 2 put in as a constant 30 us."""
        pe = SlabPlan(dims[0], H, world, rank, every)
        if self_neighbour and not (pe.has_up and pe.has_dn):
            return 0.0
        s2 = torch.zeros((pe.Lloc,) + rest, dtype=dtype, device=device)
        d2 = torch.zeros_like(s2)
        ev = torch.cuda.Event() if gpu else None

        def boundary():
            if pe.pair_view() and hasattr(sweep, "pair"):
                sweep.pair(s2[pe.top[0]:pe.top[1]], d2[pe.top[0]:pe.top[1]], s2[pe.bot[0]:pe.bot[1]], d2[pe.bot[0]:pe.bot[1]], handle)
            else:
                for v in (pe.top, pe.bot):
                    if v is not None:
                        sweep(s2[v[0]:v[1]], d2[v[0]:v[1]], handle)
            if ev is not None:
                ev.record(main)
        return timed(boundary) if (pe.top is not None or pe.bot is not None) else 0.0

    planes = min(slab_bounds(dims[0], world, r)[1] - slab_bounds(dims[0], world, r)[0] for r in range(world))
    b1 = boundary_us(1)
    b2 = boundary_us(2) if planes >= 8 * H else b1
    mine = {"interior_sweep_us": sweep_us, "exchange_H_planes_us": x1, "exchange_2H_planes_us": x2, "boundary_launch_us_every1": b1, "boundary_launch_us_every2": b2}
    cdev = coll_device(torch, dist, device)
    t = torch.tensor([sweep_us, x1, x2, b1, b2], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sweep_us, x1, x2, b1, b2 = (float(v) for v in t)
    # time of one ping-pong pair (two launches) in each mode: boundary launch, then the interior sweep beside the exchange
    every, pair1, pair2 = decide_exchange_every(sweep_us, x1, x2, b1, b2, planes, H)
    # every rank decided from the same reduced numbers; make sure of it before anybody allocates ghosts of a width the others do not expect
    lo_hi = torch.tensor([float(every), -float(every)], dtype=torch.float64, device=cdev)
    dist.all_reduce(lo_hi, op=dist.ReduceOp.MAX)
    agreed = int(lo_hi[0]) == every and int(-lo_hi[1]) == every
    if not agreed:
        every = 1
    return every, {"interior_sweep_us": sweep_us, "exchange_H_planes_us": x1, "exchange_2H_planes_us": x2,
                   "boundary_launch_us_every1": b1, "boundary_launch_us_every2": b2,
                   "pair_us_exchange_every_launch": pair1, "pair_us_one_exchange_per_pair": pair2, "chosen_every": every,
                   "ranks_agreed": bool(agreed), "this_rank": mine}


def _write_view_stc(base_stc, ndim, L_view, cache_dir, tag):
    """A copy of the spec with its outermost size (L in 3D, M in 2D) replaced (reference .stc format)."""
    import re
    text = open(base_stc).read()
    key = "L" if ndim == 3 else "M"
    new_text, n = re.subn(r"(^|\s)%s\s+\d+" % key, lambda m: "%s%s %d" % (m.group(1), key, L_view), text, count=1)
    assert n == 1, "spec has no " + key
    name = os.path.basename(base_stc)[:-4]
    path = os.path.join(cache_dir, "%s_%s%d.stc" % (name, tag, L_view))
    tmp = path + ".%d.tmp" % os.getpid()
    with open(tmp, "w") as f:
        f.write(new_text)
    os.replace(tmp, path)
    return path


def nccl_options(dist):
    """Process-group options for the slab run: RCCL's internal stream must not share an in-order hardware queue
    with the interior kernel's stream (it would run only after it), so it is created high priority."""
    o = dist.ProcessGroupNCCL.Options()
    o.is_high_priority_stream = os.environ.get("DRS_RCCL_HIGH_PRIORITY", "1") != "0"   # 0: experiments only
    return o



def coll_device(torch, dist, device):
    """Where the few-element tensors of this package's collectives live: on the device with RCCL; on the CPU when the process group
    is gloo -- the CPU tests, and rehearsals of the N > 1 path with several rank processes sharing ONE GPU (DRS_BENCH_BACKEND=gloo),
    where gloo has no GPU all_gather / send / recv."""
    return torch.device("cpu") if (dist is not None and getattr(dist, "get_backend", lambda: "")() == "gloo") else device


def batch_p2p(torch, dist, ops):
    """ops = [("send" | "recv", tensor, peer), ...] as ONE batch (RCCL: a single ncclGroupStart / ncclGroupEnd).  With a gloo group and
    device tensors the planes are staged through host memory (rehearsal only: correct, not fast); the copies run on the current stream
    and block the host, so the caller's stream / event choreography still orders them."""
    if not ops:
        return
    staged = getattr(dist, "get_backend", lambda: "")() == "gloo" and any(t.is_cuda for _, t, _ in ops)      # (the tests' in-process stand-in has no backend)
    if not staged:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend if k == "send" else dist.irecv, t, peer) for k, t, peer in ops]):
            w.wait()
        return
    host = [(k, t, peer, t.cpu() if k == "send" else torch.empty(t.shape, dtype=t.dtype)) for k, t, peer in ops]
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend if k == "send" else dist.irecv, h, peer) for k, _, peer, h in host]):
        w.wait()
    for k, t, _, h in host:
        if k == "recv":
            t.copy_(h)

class HipSweep:
    """Product sweep backend: generated HIP kernel for a view of Lv planes."""

    def __init__(self, base_stc, opts, cache_dir, alone_opts=None):
        """alone_opts: generator options for launches that run with NO exchange beside them (the whole-slab launch of an
        every = 2 pair): e.g. one stream block per tile, which sweeps a slab 7 % faster but whose long-lived workgroups would
        keep the RCCL send/recv kernel off the CUs."""
        self.base_stc, self.opts, self.cache_dir = base_stc, list(opts), cache_dir
        self.alone_opts = list(alone_opts) if alone_opts else None
        self.ndim = 3 if "--3d" in self.opts else 2
        os.makedirs(cache_dir, exist_ok=True)
        self.kernels = {}

    def kernel(self, Lv, pair=False, alone=False):
        alone = alone and self.alone_opts is not None
        if (Lv, pair, alone) not in self.kernels:
            stc = _write_view_stc(self.base_stc, self.ndim, Lv, self.cache_dir, "slabL")
            self.kernels[(Lv, pair, alone)] = drs.Kernel((self.alone_opts if alone else self.opts) + (["--pair-launch", "1"] if pair else []) + [stc])
        return self.kernels[(Lv, pair, alone)]

    def prebuild(self, plan):
        """Build (or find cached) every kernel `plan` launches -- call before HIP is initialised."""
        for lv in plan.views():
            self.kernel(lv)
        if plan.pair_view():
            self.kernel(plan.pair_view(), pair=True)
        if plan.every == 2 and plan.world > 1 and self.alone_opts is not None:
            self.kernel(plan.Lloc, alone=True)

    def alone(self, src_view, dst_view, stream):
        """A launch with nothing beside it on the GPU (SlabRun.launch_local)."""
        self.kernel(src_view.shape[0], alone=True).launch(src_view.data_ptr(), dst_view.data_ptr(), stream)

    def pair(self, src0, dst0, src1, dst1, stream):
        """Both boundary views in one launch."""
        self.kernel(src0.shape[0], pair=True).launch_pair(src0.data_ptr(), dst0.data_ptr(), src1.data_ptr(), dst1.data_ptr(), stream)

    def __call__(self, src_view, dst_view, stream):
        k = self.kernel(src_view.shape[0])
        k.launch(src_view.data_ptr(), dst_view.data_ptr(), stream)



PLACEMENT_PERIOD = 64 << 20      # csrc/emit_hip.hpp: kPlacementPeriod


def slab_pair(torch, shape, dtype, device, skew=None):
    """The two slab arrays (zero-filled).  On a GPU they are views of ONE allocation with the output `skew` bytes past a multiple of
    the 64 MiB placement period behind the input (launch time of a z-streaming kernel depends on (out - in) mod 64 MiB,
    profiles/r03_probe_skew4.log; default 32 MiB when a plane is 2 MiB or more, DRS_SLAB_SKEW_MIB overrides); on the CPU (gloo tests)
    two plain tensors.  Returns (A, B, arena)."""
    import os
    if getattr(device, "type", str(device)) != "cuda":
        return torch.zeros(shape, dtype=dtype, device=device), torch.zeros(shape, dtype=dtype, device=device), None
    esz = torch.empty(0, dtype=dtype).element_size()
    n = 1
    for d in shape:
        n *= d
    nb = n * esz
    plane = nb // shape[0] if len(shape) == 3 else 0
    if skew is None:
        env = os.environ.get("DRS_SLAB_SKEW_MIB")
        skew = (int(env) << 20) if env else ((32 << 20) if plane >= (2 << 20) else 0)
    off0 = -(-nb // PLACEMENT_PERIOD) * PLACEMENT_PERIOD
    arena = torch.zeros(off0 + PLACEMENT_PERIOD + nb, dtype=torch.uint8, device=device)      # one period of slack: the output can be moved (place_slab_output)
    off = off0 + skew % PLACEMENT_PERIOD
    return arena[:nb].view(dtype).view(shape), arena[off:off + nb].view(dtype).view(shape), arena


def place_slab_output(arena, like, skew):
    """The output array of a slab_pair arena at `skew` bytes (mod 64 MiB) past the period boundary behind the input; `like` = the input view."""
    nb = like.numel() * like.element_size()
    off = -(-nb // PLACEMENT_PERIOD) * PLACEMENT_PERIOD + skew % PLACEMENT_PERIOD
    return arena[off:off + nb].view(like.dtype).view(like.shape)


def calibrate_slab_placement(torch, run, kernel, positions=(0, 16 << 20, 32 << 20, 48 << 20), launches=6):
    """Measure where this rank's output slab should sit (LOCAL launches only, no exchange: every rank decides for itself, nothing
    collective).  `kernel` sweeps the first kernel.info["L"] planes of the slab -- the whole-slab kernel of an every = 2 run (one
    stream block per tile: a single z front, 193 us in the good half of the period and up to 216 in the bad one on a 136-plane slab,
    profiles/r03_probe_skew_slab.log) or the interior view's.  The phase of the pattern differs from allocation to allocation, so a
    fixed skew cannot be right everywhere.  Moves run.B (contents zero afterwards, like run.A's are the caller's to fill); returns the table."""
    if run._arena is None:
        return None
    st = torch.cuda.current_stream(run.device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    a = run.A.data_ptr()
    table = []
    b_ = place_slab_output(run._arena, run.A, positions[len(positions) // 2]).data_ptr()
    for _ in range(12):                                  # clocks up before the first position is timed
        kernel.launch(a, b_, st.cuda_stream); kernel.launch(b_, a, st.cuda_stream)
    for sk in positions:
        b = place_slab_output(run._arena, run.A, sk).data_ptr()
        kernel.launch(a, b, st.cuda_stream); kernel.launch(b, a, st.cuda_stream)
        ev[0].record(st)
        for _ in range(launches):
            kernel.launch(a, b, st.cuda_stream)
        ev[1].record(st)
        for _ in range(launches):
            kernel.launch(b, a, st.cuda_stream)
        ev[2].record(st)
        torch.cuda.synchronize(run.device)
        table.append((sk, ev[0].elapsed_time(ev[1]) / launches, ev[1].elapsed_time(ev[2]) / launches))
    best = min(table, key=lambda r: r[1] + r[2])
    run.B = place_slab_output(run._arena, run.A, best[0])
    run.A.zero_(); run.B.zero_()
    run.placement = {"out_minus_in_mod_period_bytes": best[0], "measured_us_fwd_bwd_by_skew_MiB": {str(sk >> 20): [round(1e3 * f, 1), round(1e3 * b, 1)] for sk, f, b in table}}
    return run.placement

class SlabRun:
    """One rank of a z-slab decomposed run.

    sweep(src_view, dst_view, stream_handle) must apply one launch of the (fused) stencil
    to a contiguous [Lv, M, N] view: outputs planes [H, Lv-H) of dst_view (HipSweep in the
    product; the CPU tests inject an oracle-backed callable to check the decomposition)."""

    def __init__(self, torch, dist, dims, H, step, iterations, rank, world, sweep, device, dtype, every=1):
        """dims = (L, M, N) for a 3D run cut along z, or (M, N) for a 2D run cut along y; every = launches per
        exchange (1 or 2, see SlabPlan)."""
        self.torch, self.dist = torch, dist
        dims = tuple(dims)
        self.plan = SlabPlan(dims[0], H, world, rank, every if world > 1 else 1)
        self.rest, self.H, self.step, self.iterations = dims[1:], H, step, iterations
        self.rank, self.world, self.sweep = rank, world, sweep
        self.device, self.dtype = device, dtype
        self.gpu = (device.type == "cuda")
        p = self.plan
        self.A, self.B, self._arena = slab_pair(torch, (p.Lloc,) + self.rest, dtype, device)
        if self.gpu:
            self.main = torch.cuda.current_stream(device)
            self.side = torch.cuda.Stream(device=device, priority=-1)
            self.ev_b = torch.cuda.Event()
            self.ev_c = torch.cuda.Event()
        self.launch_count = 0

    def load_global(self, fill):
        """fill(lo, hi) -> tensor/array of global planes [lo, hi) for buffer A (B starts as zeros)."""
        p = self.plan
        self.A.copy_(self.torch.as_tensor(fill(p.lo, p.hi)).to(self.device, self.dtype))
        self.B.zero_()

    def _stream_handle(self):
        return self.main.cuda_stream if self.gpu else 0

    def _exchange(self, dst):
        p, dist = self.plan, self.dist
        ops = []
        if p.has_up:
            ops.append(("send", dst[p.send_up[0]:p.send_up[1]], self.rank - 1))
            ops.append(("recv", dst[p.recv_up[0]:p.recv_up[1]], self.rank - 1))
        if p.has_dn:
            ops.append(("send", dst[p.send_dn[0]:p.send_dn[1]], self.rank + 1))
            ops.append(("recv", dst[p.recv_dn[0]:p.recv_dn[1]], self.rank + 1))
        batch_p2p(self.torch, dist, ops)

    def launch(self, src, dst):
        """One launch src -> dst with halo exchange of dst."""
        p, H = self.plan, self.H
        a, b = p.interior
        if self.gpu and self.world > 1:
            sh = self.main.cuda_stream
            exp = os.environ.get("DRS_EXP_SLAB", "")      # timing experiments (wrong results), DESIGN.md section 4: what each piece of an exchanging launch costs
            if exp == "folded":
                # the best a boundary-inside-the-interior-launch scheme could do: ONE launch over boundary + interior, the exchange started at once
                self.ev_b.record(self.main)
                self.sweep(src[p.top[0] if p.top else a:(p.bot[1] if p.bot else b)], dst[p.top[0] if p.top else a:(p.bot[1] if p.bot else b)], sh)
                with self.torch.cuda.stream(self.side):
                    self.side.wait_event(self.ev_b)
                    self._exchange(dst)
                    self.ev_c.record(self.side)
                self.main.wait_event(self.ev_c)
                self.launch_count += 1
                return
            if exp == "noboundary":
                pass
            elif p.pair_view() and hasattr(self.sweep, "pair"):
                self.sweep.pair(src[p.top[0]:p.top[1]], dst[p.top[0]:p.top[1]], src[p.bot[0]:p.bot[1]], dst[p.bot[0]:p.bot[1]], sh)
            else:
                for v in (p.top, p.bot):
                    if v is not None:
                        self.sweep(src[v[0]:v[1]], dst[v[0]:v[1]], sh)
            self.ev_b.record(self.main)
            if b - a > 2 * H:
                self.sweep(src[a:b], dst[a:b], sh)
            if exp == "noexchange":
                self.launch_count += 1
                return
            with self.torch.cuda.stream(self.side):
                self.side.wait_event(self.ev_b)
                self._exchange(dst)
                self.ev_c.record(self.side)
            self.main.wait_event(self.ev_c)
        else:
            sh = self._stream_handle()
            for v in (p.top, p.bot):
                if v is not None:
                    self.sweep(src[v[0]:v[1]], dst[v[0]:v[1]], sh)
            if b - a > 2 * H:
                self.sweep(src[a:b], dst[a:b], sh)
            if self.world > 1:
                self._exchange(dst)
        self.launch_count += 1

    def timeline(self, pairs=4):
        """What one exchanging launch is made of on THIS rank, in microseconds averaged over `pairs` ping-pong pairs of the run's own loop
        (collective: every rank calls it with the same count): the boundary launch, the interior sweep, the halo exchange on the side
        stream (from the moment the boundary planes exist), and how long the main stream then still waits for the ghosts -- 0 when the
        exchange hides under the interior sweep.  every = 2: also the whole-slab launch that has no exchange beside it."""
        if not (self.gpu and self.world > 1):
            return None
        torch, p, H = self.torch, self.plan, self.H
        a, b = p.interior
        E = lambda: torch.cuda.Event(enable_timing=True)
        acc = {"boundary_us": 0.0, "interior_us": 0.0, "exchange_us": 0.0, "wait_for_ghosts_us": 0.0, "exchanging_launch_us": 0.0, "whole_slab_launch_us": 0.0}
        n_x = n_l = 0
        sh = self.main.cuda_stream
        for _ in range(pairs):
            for src, dst, local in ((self.A, self.B, p.every == 2), (self.B, self.A, False)):
                e = [E() for _ in range(6)]
                e[0].record(self.main)
                if local:
                    self.launch_local(src, dst)
                    e[1].record(self.main)
                    torch.cuda.synchronize(self.device)
                    acc["whole_slab_launch_us"] += e[0].elapsed_time(e[1]) * 1e3
                    n_l += 1
                    continue
                if p.pair_view() and hasattr(self.sweep, "pair"):
                    self.sweep.pair(src[p.top[0]:p.top[1]], dst[p.top[0]:p.top[1]], src[p.bot[0]:p.bot[1]], dst[p.bot[0]:p.bot[1]], sh)
                else:
                    for v in (p.top, p.bot):
                        if v is not None:
                            self.sweep(src[v[0]:v[1]], dst[v[0]:v[1]], sh)
                e[1].record(self.main)
                self.ev_b.record(self.main)
                if b - a > 2 * H:
                    self.sweep(src[a:b], dst[a:b], sh)
                e[2].record(self.main)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(self.ev_b)
                    e[3].record(self.side)
                    self._exchange(dst)
                    e[4].record(self.side)
                    self.ev_c.record(self.side)
                self.main.wait_event(self.ev_c)
                e[5].record(self.main)
                self.launch_count += 1
                torch.cuda.synchronize(self.device)
                acc["boundary_us"] += e[0].elapsed_time(e[1]) * 1e3
                acc["interior_us"] += e[1].elapsed_time(e[2]) * 1e3
                acc["exchange_us"] += e[3].elapsed_time(e[4]) * 1e3
                acc["wait_for_ghosts_us"] += e[2].elapsed_time(e[5]) * 1e3
                acc["exchanging_launch_us"] += e[0].elapsed_time(e[5]) * 1e3
                n_x += 1
        out = {k: round(v / max(1, n_l if k == "whole_slab_launch_us" else n_x), 1) for k, v in acc.items()}
        out["every"] = p.every
        out["planes_held"] = p.Lloc
        return out

    def launch_local(self, src, dst):
        """One launch src -> dst over the whole local slab, no exchange (first launch of a pair, every = 2)."""
        (getattr(self.sweep, "alone", None) or self.sweep)(src, dst, self._stream_handle())
        self.launch_count += 1

    def run(self, iterations=None):
        """The reference's ping-pong loop (codegen.hpp:581-584) on the slab; result in A."""
        it = self.iterations if iterations is None else iterations
        n, t = 0, 0
        while t < it:
            if self.plan.every == 2:
                self.launch_local(self.A, self.B)
            else:
                self.launch(self.A, self.B)
            self.launch(self.B, self.A)
            n += 2
            t += 2 * self.step
        return n

    def owned(self, buf):
        """The planes of `buf` this rank owns (global [z0, z1))."""
        p = self.plan
        return buf[p.z0 - p.lo:p.z1 - p.lo]


class SelfNeighbourRun(SlabRun):
    """Rehearsal of a middle rank on ONE GPU: both neighbours are this rank (what it sends "up" arrives in its lower
    ghost planes and vice versa), through the same batch_isend_irecv calls on a real process group of size 1.  Used by
    scripts/rccl_probe.py, bench.py's DRS_REHEARSE mode and the GPU tests; bytes move through RCCL's self-copy, not xGMI."""

    def _exchange(self, dst):
        p, dist = self.plan, self.dist
        assert p.has_up and p.has_dn, "rehearse a middle rank"
        ops = [dist.P2POp(dist.isend, dst[p.send_up[0]:p.send_up[1]], 0), dist.P2POp(dist.irecv, dst[p.recv_dn[0]:p.recv_dn[1]], 0),
               dist.P2POp(dist.isend, dst[p.send_dn[0]:p.send_dn[1]], 0), dist.P2POp(dist.irecv, dst[p.recv_up[0]:p.recv_up[1]], 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()


class NativeSlabRun:
    """SlabRun's interface over the native N > 1 entry points of the C ABI (drs_slab_*, csrc/slab.hpp): the plan, the kernels,
    the RCCL group of send/recv on a high-priority side stream and the HIP graph of one ping-pong pair all live in
    libdrstencil_amd.so; torch only owns the two device buffers and carries the 128-byte communicator id from rank 0 to the
    others (any transport would do).  `rehearse_world` > 0 plays a middle rank on one GPU with itself as both neighbours
    (SelfNeighbourRun's counterpart).  Every view kernel must already be in the cache (HipSweep.prebuild built them)."""

    def __init__(self, torch, dist, base_stc, opts, dims, H, step, iterations, rank, world, device, dtype, every=1, alone_opts=None,
                 rehearse_world=0, cache_dir=None):
        self.torch, self.dist = torch, dist
        dims = tuple(dims)
        pworld = rehearse_world or world
        self.plan = SlabPlan(dims[0], H, pworld, rank, every if pworld > 1 else 1)
        self.rest, self.H, self.step, self.iterations = dims[1:], H, step, iterations
        self.rank, self.world, self.device, self.dtype = rank, world, device, dtype
        self.slab = drs.Slab(list(opts) + [base_stc], world=1 if rehearse_world else world, rank=rank, every=every, alone_args=alone_opts,
                             rehearse_world=rehearse_world, cache_dir=cache_dir)
        p = self.plan
        assert (self.slab.lo, self.slab.hi, self.slab.z0, self.slab.z1, self.slab.Lloc) == (p.lo, p.hi, p.z0, p.z1, p.Lloc)
        self.A, self.B, self._arena = slab_pair(torch, (p.Lloc,) + self.rest, dtype, device)
        # the communicator id: made by rank 0, carried by whatever the host has (here: the torch process group)
        if rehearse_world or world == 1:
            uid = drs.slab_unique_id()
        else:
            box = [drs.slab_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        self.slab.connect(uid)
        self.main = torch.cuda.ExternalStream(self.slab.stream, device=device)
        self.launch_count = 0

    def load_global(self, fill):
        p = self.plan
        self.A.copy_(self.torch.as_tensor(fill(p.lo, p.hi)).to(self.device, self.dtype))
        self.B.zero_()
        self.torch.cuda.synchronize()

    def run(self, iterations=None):
        n = self.slab.run(self.A.data_ptr(), self.B.data_ptr(), self.iterations if iterations is None else iterations)
        self.launch_count += n
        return n

    def owned(self, buf):
        p = self.plan
        return buf[p.z0 - p.lo:p.z1 - p.lo]
