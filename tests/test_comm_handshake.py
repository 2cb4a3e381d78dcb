"""Forming the multi-GPU communicator must never crash a process or leave ranks in different collectives.

CPU tier (no GPU): the RCCL loader of rt_comm.h with candidates that cannot be opened, the non-collective probe
rt_comm_available, and the hand-shake of TileParallelRenderer over gloo (world 2) when RCCL is missing on one rank
or rank 0 cannot make an id — every rank must get CommUnavailable, none may hang."""
import os
import socket
import subprocess
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(code, env_extra, timeout=120):
    env = dict(os.environ)
    env.update(env_extra)
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n%s" % (ROOT, code)],
                          capture_output=True, text=True, timeout=timeout, env=env)


def test_a_candidate_that_cannot_be_opened_is_reported_not_fatal():
    """rt_comm.h used to call dlerror() twice (the second call returns NULL -> std::string + nullptr -> SIGSEGV) for any
    candidate dlopen rejects, e.g. a bad $RT355_RCCL_LIB or a host without librccl on the loader path."""
    code = (
        "import ctypes as C\n"
        "from rsraytracingv3_amd import _native as N\n"
        "lib = N.load()\n"
        "buf = C.create_string_buffer(2048)\n"
        "rc = lib.rt_comm_available(buf, len(buf))\n"
        "print(rc); print(buf.value.decode())\n"
        "id_ = C.create_string_buffer(128)\n"
        "print(lib.rt_comm_get_unique_id(id_))\n"
    )
    out = _run(code, {"RT355_RCCL_LIB": "/nonexistent/librccl.so.1"})
    assert out.returncode == 0, "the process died (rc %d): %s" % (out.returncode, out.stderr[-800:])
    lines = out.stdout.strip().splitlines()
    rc = int(lines[0])
    assert rc in (0, -3)  # bound through a later candidate (/opt/rocm) — or RT_ERR_NOT_READY, with the reasons
    if rc == -3:
        assert "/nonexistent/librccl.so.1" in lines[1]
        assert int(lines[-1]) == -3
    else:
        assert "librccl" in lines[1]


def test_a_file_that_is_not_an_rccl_is_rejected_by_its_symbols(tmp_path):
    """A library that loads but lacks the RCCL entry points (ncclCommAbort included: it is required) is skipped."""
    code = (
        "import ctypes as C\n"
        "from rsraytracingv3_amd import _native as N\n"
        "lib = N.load()\n"
        "buf = C.create_string_buffer(4096)\n"
        "print(lib.rt_comm_available(buf, len(buf))); print(buf.value.decode())\n"
    )
    out = _run(code, {"RT355_RCCL_LIB": "libz.so.1"})
    assert out.returncode == 0, out.stderr[-800:]
    lines = out.stdout.strip().splitlines()
    if int(lines[0]) == -3:
        assert "libz.so.1: not an RCCL" in lines[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _FakeDevice:
    """Stands in for rsraytracingv3_amd.Device on the CPU tier: the hand-shake only needs these four calls."""
    mode = "ok"
    rank = 0
    calls = None

    @classmethod
    def comm_available(cls):
        if cls.mode == "no_rccl_on_rank_1" and cls.rank == 1:
            return False, "librccl.so.1: cannot open shared object file"
        return True, "/fake/librccl.so.1"

    @classmethod
    def comm_unique_id(cls):
        if cls.mode == "no_id":
            raise RuntimeError("rt_comm_get_unique_id: RT_ERR_DEVICE")
        return b"\x07" * 128

    def comm_init(self, ident, rank, world, strip_rows):
        type(self).calls.append(("init", rank, world, strip_rows, ident))
        if type(self).mode == "init_fails_on_rank_1" and rank == 1:
            raise RuntimeError("ncclCommInitRank failed: unhandled system error")

    def comm_abort(self):
        type(self).calls.append(("abort",))

    class _Lib:
        rt_render = rt_render_frames = None

    _lib = _Lib()


def _handshake_worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rsraytracingv3_amd.distributed import CommUnavailable, TileParallelRenderer
        _FakeDevice.mode, _FakeDevice.rank, _FakeDevice.calls = mode, rank, []
        try:
            TileParallelRenderer(_FakeDevice(), 64, 40, 8, init_timeout_s=20.0)
            verdict = "joined"
        except CommUnavailable as exc:
            verdict = "unavailable fatal=%s: %s" % (exc.fatal, exc)
        with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
            f.write(verdict + "\n" + repr(_FakeDevice.calls))
        dist.barrier()  # every rank is still able to run a collective afterwards
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ok", "no_rccl_on_rank_1", "no_id", "init_fails_on_rank_1"])
def test_every_rank_reaches_the_same_verdict(tmp_path, mode):
    port = _free_port()
    mp.spawn(_handshake_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "rank0").read_text().splitlines()
    r1 = (tmp_path / "rank1").read_text().splitlines()
    if mode == "ok":
        assert r0[0] == "joined" and r1[0] == "joined"
        assert "('init', 0, 2, 8" in r0[1] and "('init', 1, 2, 8" in r1[1]
    else:
        assert r0[0].startswith("unavailable fatal=False") and r1[0].startswith("unavailable fatal=False"), (r0, r1)
        if mode in ("no_rccl_on_rank_1", "no_id"):
            assert "init" not in r0[1] and "init" not in r1[1]  # nobody entered the collective
        else:
            assert "('abort',)" in r0[1]  # the rank that did join gave the communicator up instead of waiting for its peer
