"""Runs one function on `world` FRESH processes that all bind cuda:0 and form one communicator of the library
(kc_comm_init): the way the N > 1 path of csrc/comm.cpp is executed on a box with a single GPU.  The children are
spawned (not forked) before they touch the GPU; each imports tests/rank_scenarios.py, runs `fn_name(kc, rank, world, **kwargs)`
and sends its (picklable) result back.  A child that fails sends its traceback; a child that hangs is killed by PID after
`timeout` seconds (the library's own host-side waits give up after KC_COMM_TIMEOUT_S, set to 30 s here)."""
import os
import traceback

import multiprocessing as mp


def _child(rank, world, comm_id, fn_name, kwargs, q, env):
    os.environ.update(env)
    try:
        import rank_scenarios
        import kanter_core_amd as kc
        kc.init(0)
        kc.comm_init(rank, world, comm_id)
        assert kc.comm_info() == (rank, world)
        out = getattr(rank_scenarios, fn_name)(kc, rank, world, **kwargs)
        kc.sync()
        kc.comm_destroy()
        q.put((rank, "ok", out))
    except BaseException:  # noqa: BLE001 -- the parent wants to see everything
        q.put((rank, "error", traceback.format_exc()))


def run_ranks(world, fn_name, timeout=300, env=None, transport=None, **kwargs):
    """-> [result of rank 0, result of rank 1, ...]; raises AssertionError with the children's tracebacks."""
    import kanter_core_amd as kc
    if transport:
        os.environ["KC_COMM_TRANSPORT"] = transport
    try:
        comm_id = kc.comm_unique_id()  # host only: a segment name (and an RCCL id when that wire is asked for)
    finally:
        if transport:
            del os.environ["KC_COMM_TRANSPORT"]
    child_env = {"KC_COMM_TIMEOUT_S": "30"}
    child_env.update(env or {})
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_child, args=(r, world, comm_id, fn_name, kwargs, q, child_env)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            rank, status, out = q.get(timeout=timeout)
            got[rank] = (status, out)
    finally:
        for p in procs:
            p.join(timeout=60 if len(got) == world else 1)
            if p.is_alive():
                p.kill()  # exactly the process started above
                p.join(timeout=10)
    errors = ["rank %d:\n%s" % (r, o) for r, (s, o) in sorted(got.items()) if s != "ok"]
    assert not errors and len(got) == world, "\n".join(errors) or "a rank did not answer"
    return [got[r][1] for r in range(world)]


def run_ranks_expect_failure(world, fn_name, timeout=120, env=None, **kwargs):
    """Like run_ranks for scenarios in which ranks are EXPECTED to fail: -> [(status, result or traceback)] per rank."""
    import kanter_core_amd as kc
    comm_id = kc.comm_unique_id()
    child_env = {"KC_COMM_TIMEOUT_S": "8"}
    child_env.update(env or {})
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_child, args=(r, world, comm_id, fn_name, kwargs, q, child_env)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            rank, status, out = q.get(timeout=timeout)
            got[rank] = (status, out)
    finally:
        for p in procs:
            p.join(timeout=30 if len(got) == world else 1)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
    assert len(got) == world, "a rank hung instead of failing"
    return [got[r] for r in range(world)]
