"""bench.py --gpus N as a plain command (no torch.distributed.run): it must start N rank processes itself, BEFORE anything
in the parent imports torch or touches HIP (a launcher hop is only legal before any GPU call), relay rank 0's JSON line and
propagate failures.  The rank command is stubbed through DRS_BENCH_CHILD, so no GPU is needed."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
import json, os, sys, time
out = os.environ["STUB_OUT"]
tag = "prebuild" if "--prebuild-only" in sys.argv else "rank_" + os.environ.get("RANK", "none")
keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DRS_NO_COMPILE", "HSA_ENABLE_IPC_MODE_LEGACY")
json.dump({"argv": sys.argv[1:], "env": {k: os.environ.get(k) for k in keys}, "pid": os.getpid()}, open(os.path.join(out, tag + ".json"), "w"))
if tag == "prebuild":
    sys.exit(0)
if os.environ.get("STUB_HANG"):
    time.sleep(120)         # a rank stuck in a rendezvous: only the launcher's deadline ends it
fail = os.environ.get("STUB_FAIL_RANK")
if fail is not None:
    if os.environ["RANK"] == fail:
        sys.exit(3)
    time.sleep(60)          # the other ranks would wait in RCCL for ever: the parent has to end them
if os.environ["RANK"] == "0":
    print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": int(os.environ["WORLD_SIZE"])}))
'''

PARENT = r'''
import sys
sys.path.insert(0, %r)
import bench
code = 0
try:
    bench.main(%r)
except SystemExit as e:
    code = e.code or 0
assert "torch" not in sys.modules, "the parent imported torch"
# (round 4: bench.py reads its option lists from drstencil_amd/tuned_defaults.tsv through the package's pure-Python reader; the package
# may be imported, the native library -- which links the HIP runtime -- must not be loaded)
assert getattr(sys.modules.get("drstencil_amd"), "_lib", None) is None, "the parent loaded the native library"
sys.exit(code)
'''


def _run(tmp_path, argv, extra_env=None, timeout=60):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DRS_REHEARSE")}
    env.update(DRS_BENCH_CHILD=json.dumps([sys.executable, str(stub)]), STUB_OUT=str(tmp_path))
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", PARENT % (ROOT, argv)], env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_n_starts_n_ranks_with_the_rendezvous_environment(tmp_path):
    r = _run(tmp_path, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["metric"] == "stub" and line["n_gpus"] == 4                 # rank 0's line, relayed
    pre = json.load(open(tmp_path / "prebuild.json"))
    assert pre["argv"] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--prebuild-only"] and pre["env"]["RANK"] is None
    ranks = [json.load(open(tmp_path / ("rank_%d.json" % i))) for i in range(4)]
    ports = {x["env"]["MASTER_PORT"] for x in ranks}
    assert len(ports) == 1 and int(ports.pop()) > 0
    assert len({x["pid"] for x in ranks}) == 4
    for i, x in enumerate(ranks):
        e = x["env"]
        assert x["argv"] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["LOCAL_WORLD_SIZE"]) == (str(i), str(i), "4", "4")
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["DRS_NO_COMPILE"] == "1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_failed_rank_ends_the_others_and_the_exit_code_is_propagated(tmp_path):
    t0 = time.time()
    r = _run(tmp_path, ["--gpus", "3"], {"STUB_FAIL_RANK": "1"})
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
    assert time.time() - t0 < 30, "the surviving ranks were not terminated"


def test_hung_ranks_are_ended_at_the_deadline(tmp_path):
    """spawn_ranks used to poll for ever when a rank hangs without exiting; --rank-timeout ends the ranks and the
    launcher exits 124."""
    t0 = time.time()
    r = _run(tmp_path, ["--gpus", "2", "--rank-timeout", "2"], {"STUB_HANG": "1"})
    assert r.returncode == 124, (r.returncode, r.stdout, r.stderr)
    assert time.time() - t0 < 30 and "still running after 2 s" in r.stderr


def test_world_size_mismatch_is_refused_before_the_gpu_is_touched(tmp_path):
    # as one rank of a launcher (WORLD_SIZE set) with a --gpus that does not match: an error, not an assert after HIP init
    r = _run(tmp_path, ["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_oracle_slab_checker_compares_in_chunks():
    """bench.oracle_check_slabs (the checker half of the CPU leg) on a slab thicker than one 64-slice chunk: equal output -> ok with max_rel 0;
    one changed value in the last chunk -> not ok, and the relative error is the one planted (the whole-grid check of a 1024^3 launch walks
    the same loop)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    import oracle
    stc = os.path.join(ROOT, "tests", "stc", "t3_star.stc")
    bench.WORKLOADS["_chunk_test"] = dict(stc=stc, ndim=3, dtype="fp64", name="test")
    try:
        spec = oracle.Spec(stc, 3, 1)
        L, M, N = spec.dims
        h = spec.halo
        nsl = 70                              # two chunks: 64 + 4 output slices
        cs = oracle.Spec(stc, 3, 1)
        cs.set_dims(nsl, M, N)
        src = oracle.fill_random((nsl, M, N), np.float64)
        dst = np.zeros_like(src)
        oracle.sweep(cs, src, dst, contract=1)
        out = dst[h:nsl - h].copy()
        good = bench.oracle_check_slabs("_chunk_test", 1, [{"label": "whole_grid", "z0": 0, "input": src, "output": out}], False)
        assert good["ok"] and good["max_rel"] == 0.0 and good["slabs"][0]["slices"] == nsl
        bad_out = out.copy()
        bad_out[-1, h + 1, h + 2] *= 1.0 + 1e-9
        bad = bench.oracle_check_slabs("_chunk_test", 1, [{"label": "whole_grid", "z0": 0, "input": src, "output": bad_out}], False)
        assert not bad["ok"] and 0.5e-9 < bad["max_rel"] < 2e-9
        tol = bench.oracle_check_slabs("_chunk_test", 1, [{"label": "whole_grid", "z0": 0, "input": src, "output": bad_out}], True)      # a pipeline's bar: 1e-12
        assert not tol["ok"]
        bad_out[-1, h + 1, h + 2] = out[-1, h + 1, h + 2] * (1.0 + 1e-14)
        assert bench.oracle_check_slabs("_chunk_test", 1, [{"label": "whole_grid", "z0": 0, "input": src, "output": bad_out}], True)["ok"]
    finally:
        del bench.WORKLOADS["_chunk_test"]
