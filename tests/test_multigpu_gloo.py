"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the z-slab decomposition with the
oracle sweep standing in for the HIP kernel, checked bit-exactly against the single-domain
oracle run.  Exercises SlabPlan's index bookkeeping, ghost planes of both ping-pong
buffers, the frozen global ring and the send/recv pairing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from drstencil_amd.multigpu import SlabPlan, SlabRun, choose_exchange_every, slab_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STC = os.path.join(ROOT, "tests", "stc", "smoke3.stc")   # 40 x 36 x 256, 3d7pt_star


def test_slab_plan_covers_every_interior_plane_once():
    for L, H, R in [(40, 1, 2), (40, 2, 3), (1024, 1, 8), (1024, 2, 8), (37, 3, 4), (64, 1, 1)]:
        seen = np.zeros(L, int)
        for r in range(R):
            p = SlabPlan(L, H, R, r)
            assert (p.z0, p.z1) == slab_bounds(L, R, r)
            for a, b in p.outputs():
                seen[a:b] += 1
            if p.has_up:
                assert p.send_up == (H, 2 * H) and p.recv_up == (0, H)
            if p.has_dn:
                assert p.send_dn[1] == p.Lloc - H and p.recv_dn[1] == p.Lloc
        assert np.all(seen[H:L - H] == 1) and not seen[:H].any() and not seen[L - H:].any()
    with pytest.raises(ValueError):
        SlabPlan(16, 3, 4, 1)


def test_slab_plan_wide_ghosts_one_exchange_per_pair():
    """every = 2: ghost width 2H; the exchanging launch writes exactly the owned planes, the launch before it the
    whole local slab; what is sent are the first/last 2H owned planes."""
    for L, H, R in [(40, 1, 2), (60, 2, 3), (1024, 2, 8), (1024, 1, 4)]:
        seen = np.zeros(L, int)
        for r in range(R):
            p = SlabPlan(L, H, R, r, every=2)
            G = 2 * H
            assert p.G == G and p.lo == p.z0 - (G if p.has_up else 0) and p.hi == p.z1 + (G if p.has_dn else 0)
            for a, b in p.outputs():
                seen[a:b] += 1
            outs = sorted(p.outputs())
            assert outs[0][0] == max(p.z0, H) and outs[-1][1] == min(p.z1, L - H)
            assert p.full == (0, p.Lloc) and p.Lloc in p.views()
            if p.has_up:
                assert p.send_up == (G, 2 * G) and p.recv_up == (0, G) and p.lo + p.send_up[0] == p.z0
            if p.has_dn:
                assert p.send_dn == (p.Lloc - 2 * G, p.Lloc - G) and p.lo + p.send_dn[1] == p.z1
        assert np.all(seen[H:L - H] == 1) and not seen[:H].any() and not seen[L - H:].any()
    with pytest.raises(ValueError):
        SlabPlan(40, 3, 4, 1, every=2)


def test_exchange_frequency_model():
    # C4: 4 MiB planes, H = 2 -> one exchange per pair up to 4 GPUs (sweep of 256 planes >> transfer of 4 planes), every launch at 8
    assert [choose_exchange_every(1024 // n, 4 << 20, 2) for n in (2, 4, 8)] == [2, 2, 1]
    assert choose_exchange_every(12, 4 << 20, 2) == 1          # slab too thin for ghosts twice as wide


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, step, iterations, every, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = oracle.Spec(STC, 3, step)
    L, M, N = spec.dims
    H = spec.halo

    def sweep(src, dst, stream):
        vs = oracle.Spec(STC, 3, step)
        vs.set_dims(src.shape[0], M, N)
        a, b = src.numpy(), dst.numpy()
        assert a.flags.c_contiguous and b.flags.c_contiguous
        oracle.sweep(vs, a, b, 1)

    run = SlabRun(torch, dist, (L, M, N), H, step, iterations, rank, world, sweep, torch.device("cpu"), torch.float32, every=every)
    full = oracle.fill_random((L, M, N), np.float32)
    run.load_global(lambda lo, hi: full[lo:hi])
    n = run.run()
    q.put((rank, n, run.plan.z0, run.plan.z1, run.owned(run.A).numpy().copy(), run.owned(run.B).numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,step,iterations,every", [(2, 1, 4, 1), (2, 2, 4, 1), (3, 1, 5, 1), (2, 2, 8, 2), (3, 1, 6, 2)])
def test_slab_decomposition_matches_single_domain(world, step, iterations, every):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, step, iterations, every, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    spec = oracle.Spec(STC, 3, step)
    spec.iterations = iterations
    A = oracle.fill_random(spec.shape, np.float32)
    B = np.zeros_like(A)
    n_ref = oracle.run(spec, A, B, contract=1)
    for rank, n, z0, z1, a, b in parts:
        assert n == n_ref
        assert np.array_equal(a, A[z0:z1]), "rank %d buffer A differs" % rank
        assert np.array_equal(b, B[z0:z1]), "rank %d buffer B differs" % rank


STC2 = os.path.join(ROOT, "tests", "stc", "t2_box25.stc")   # 203 x 772, 2d25pt_box (halo 2)


def _worker2d(rank, world, port, every, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = oracle.Spec(STC2, 2, 1)
    _, M, N = spec.dims

    def sweep(src, dst, stream):
        vs = oracle.Spec(STC2, 2, 1)
        vs.set_dims(1, src.shape[0], N)
        oracle.sweep(vs, src.numpy(), dst.numpy(), 1)

    run = SlabRun(torch, dist, (M, N), spec.halo, 1, spec.iterations, rank, world, sweep, torch.device("cpu"), torch.float64, every=every)
    full = oracle.fill_random((M, N), np.float64)
    run.load_global(lambda lo, hi: full[lo:hi])
    n = run.run()
    q.put((rank, n, run.plan.z0, run.plan.z1, run.owned(run.A).numpy().copy(), run.owned(run.B).numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("every", [1, 2])
def test_2d_y_slab_decomposition_matches_single_domain(every):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker2d, args=(r, world, port, every, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    spec = oracle.Spec(STC2, 2, 1)
    A = oracle.fill_random(spec.shape, np.float64)
    B = np.zeros_like(A)
    n_ref = oracle.run(spec, A, B, contract=1)
    for rank, n, z0, z1, a, b in parts:
        assert n == n_ref
        assert np.array_equal(a, A[z0:z1]) and np.array_equal(b, B[z0:z1])


def _verify_worker(rank, world, port, step, iterations, every, break_exchange, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = oracle.Spec(STC, 3, step)
    L, M, N = spec.dims
    H = spec.halo

    def sweep(src, dst, stream):
        vs = oracle.Spec(STC, 3, step)
        vs.set_dims(src.shape[0], M, N)
        oracle.sweep(vs, src.numpy(), dst.numpy(), 1)

    run = SlabRun(torch, dist, (L, M, N), H, step, iterations, rank, world, sweep, torch.device("cpu"), torch.float32, every=every)
    if break_exchange and rank == 1:
        real = run._exchange

        def stale(dst):           # a rank whose received ghost planes are wrong: the check must say so on every rank
            real(dst)
            if run.plan.has_up:
                dst[run.plan.recv_up[0]:run.plan.recv_up[1]].zero_()
        run._exchange = stale
    ok, detail = bench.verify_slab_run(torch, dist, run, sweep, (L, M, N), H, spec.launches, iterations, rank, world, torch.device("cpu"), torch.float32)
    # the oracle-as-checker half of a GPU run's verify_slab_run (skipped above: CPU tensors), called directly: this rank's own planes after the
    # exchanged run against oracle sweeps of the wider slab -- it must agree with the verdict above on this rank
    bench.WORKLOADS["_gloo_test"] = dict(stc=STC, ndim=3, dtype="fp32", name="test")
    n_launches = 2 * (-(-iterations // (2 * step)))
    ok_oracle = bench.oracle_check_slab_run(torch, run, "_gloo_test", step, (L, M, N), H, spec.launches, n_launches, rank, world, torch.float32)
    assert ok_oracle == detail["decomposed_vs_single_domain"]["this_rank_ok"], (rank, ok_oracle, detail)
    assert detail["vs_cpu_oracle_own_planes"]["this_rank_ok"] is None      # not run inside verify_slab_run on CPU tensors
    q.put((rank, ok, detail["decomposed_vs_single_domain"]["this_rank_ok"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,step,every,break_exchange", [(2, 2, 1, False), (3, 1, 2, False), (2, 2, 1, True)])
def test_bench_verifies_a_decomposed_run_without_a_global_reference(world, step, every, break_exchange):
    """bench.py's N > 1 check (verify_slab_run): each rank compares its exchanged run with plain launches on a wider slab of the
    same seeded global grid -- green for a correct exchange, red on EVERY rank when one rank's ghost planes are stale."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, world, port, step, 4, every, break_exchange, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert [r[0] for r in res] == list(range(world))
    if break_exchange:
        assert not any(r[1] for r in res) and not res[1][2]        # AND over the ranks; rank 1 itself saw the difference
    else:
        assert all(r[1] and r[2] for r in res)


def test_exchange_decision_rule():
    """decide_exchange_every on its own: a fast link takes one exchange per pair; a link too slow to hide 2H planes under one sweep, but
    fast enough to hide H planes under each of two, flips to an exchange per launch; thin slabs never get wide ghosts."""
    from drstencil_amd.multigpu import decide_exchange_every
    assert decide_exchange_every(200.0, 40.0, 80.0, 20.0, 20.0, 128, 2)[0] == 2
    assert decide_exchange_every(200.0, 150.0, 300.0, 20.0, 20.0, 128, 2)[0] == 1
    assert decide_exchange_every(200.0, 400.0, 800.0, 20.0, 20.0, 128, 2)[0] == 1      # slower than the sweep either way: no redundant planes on top
    assert decide_exchange_every(200.0, 1.0, 2.0, 20.0, 20.0, 12, 2)[0] == 1


def _calib_worker(rank, world, port, us_per_plane, q):
    """measure_exchange_every on CPU ranks over gloo: a `sweep` that takes 10 ms + 400 us per plane and a link that takes `us_per_plane` per plane
    sent (the sleep is inside the exchange of THIS run's process group, i.e. where a slow xGMI link would show)."""
    import time
    sys.path.insert(0, ROOT)
    import drstencil_amd.multigpu as mg
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    real = mg.batch_p2p

    def slow_link(torch_, dist_, ops):
        planes = max((t.shape[0] for k, t, _ in ops if k == "send"), default=0)
        real(torch_, dist_, ops)
        time.sleep(planes * us_per_plane * 1e-6)
    mg.batch_p2p = slow_link

    def sweep(src, dst, stream):
        time.sleep(10e-3 + src.shape[0] * 400e-6)          # a launch costs a fixed 10 ms + 0.4 ms per plane: separate boundary launches are what every = 2 saves
    every, m = mg.measure_exchange_every(torch, dist, (256, 8, 16), 2, rank, world, sweep, torch.device("cpu"), torch.float32, reps=3)
    q.put((rank, every, m))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("us_per_plane,expected", [(50.0, 2), (30000.0, 1)])
def test_slow_link_flips_the_exchange_mode(us_per_plane, expected):
    """world 2 over gloo, 128 planes per rank, H = 2: an interior sweep takes ~60 ms, a boundary launch ~12 ms (sleeps long enough to stand
    out of a loaded test machine's noise).  With a link at 50 us per plane one exchange per pair wins (it saves a boundary launch per pair:
    ~135 against ~144 ms); at 30 ms per plane the 4-plane exchange takes 120 ms and no longer hides under one sweep, the 2-plane one (60 ms)
    still does -- the calibration must flip to an exchange per launch, on every rank alike, and say that the ranks agreed."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_calib_worker, args=(r, world, port, us_per_plane, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [expected] * world, res
    for _, _, m in res:
        assert m["ranks_agreed"] and m["chosen_every"] == expected and set(m["this_rank"]) >= {"interior_sweep_us", "exchange_H_planes_us", "exchange_2H_planes_us"}
        assert m["exchange_H_planes_us"] > 0 and (expected == 2 or m["exchange_2H_planes_us"] > m["exchange_H_planes_us"])
    assert res[0][2]["pair_us_one_exchange_per_pair"] == res[1][2]["pair_us_one_exchange_per_pair"]      # decided from the same reduced numbers
