"""Which GPU a consensus job takes (falcon_amd/devices.py): one by default, chosen so that the jobs
fc_run starts together spread over the node -- the reference's worker sizes itself from --n-core
alone (falcon_kit/mains/consensus.py:258-264) and several of them run at once
(falcon_kit/mains/consensus_split.py:55-85).  CPU only: lock files and stand-in engines."""
import multiprocessing as mp
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from falcon_amd import devices as D  # noqa: E402


def _job(lock_dir, n_dev, out, go):
    os.environ["FALCON_AMD_LOCK_DIR"] = lock_dir
    out.put(D.choose_device(list(range(n_dev))))
    go.wait()  # (the lock lives as long as the process)


def _start_jobs(n, lock_dir, n_dev):
    ctx = mp.get_context("fork")
    out, go = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=_job, args=(lock_dir, n_dev, out, go)) for _ in range(n)]
    for p in ps:
        p.start()
    got = [out.get(timeout=60) for _ in ps]
    return ps, go, got


def test_jobs_started_together_take_different_gpus(tmp_path):
    """Eight jobs on an eight-GPU node: eight different devices; the ninth shares one; when the
    eight are gone, the next job has the node to itself again."""
    lock_dir = str(tmp_path)
    ps, go, got = _start_jobs(8, lock_dir, 8)
    try:
        assert sorted(got) == list(range(8))
        ps9, go9, got9 = _start_jobs(1, lock_dir, 8)
        assert got9[0] in range(8)
        go9.set()
        for p in ps9:
            p.join(30)
    finally:
        go.set()
        for p in ps:
            p.join(30)
    ps, go, got = _start_jobs(3, lock_dir, 8)
    go.set()
    for p in ps:
        p.join(30)
    assert len(set(got)) == 3


def test_more_jobs_than_gpus_fill_every_gpu_before_doubling_up(tmp_path):
    ps, go, got = _start_jobs(12, str(tmp_path), 4)
    go.set()
    for p in ps:
        p.join(30)
    assert sorted(got.count(d) for d in range(4)) == [3, 3, 3, 3]


class _Engine:
    made = []

    def __init__(self, device):
        self.device = device
        _Engine.made.append(device)

    def close(self):
        pass


@pytest.fixture
def stand_in(monkeypatch, tmp_path):
    import falcon_amd.engine
    _Engine.made = []
    D._held_locks.clear()
    monkeypatch.setenv("FALCON_AMD_LOCK_DIR", str(tmp_path))
    monkeypatch.delenv("FALCON_AMD_DEVICES", raising=False)
    monkeypatch.setattr(falcon_amd.engine, "Engine", _Engine)
    monkeypatch.setattr(D, "visible_devices", lambda: list(range(8)))
    yield
    for fd in D._held_locks:
        os.close(fd)
    D._held_locks.clear()


def test_a_worker_opens_one_context_and_grows_only_under_pressure(stand_in):
    """The single-stream worker: ONE engine on ONE device; a second device -- an idle one -- only
    when every batch slot of the first is taken; never a context it gives no batch to."""
    pool = D.open_pool()
    assert len(_Engine.made) == 1 and len(pool.devices) == 1
    taken = [pool.take() for _ in range(D.DevicePool.MAX_QUEUED)]
    assert len(_Engine.made) == 1 and all(d is pool.devices[0] for d in taken)
    extra = pool.take()  # the queue is full: now, and only now, another GPU
    assert len(_Engine.made) == 2 and extra is pool.devices[1] and extra.batches == 1
    assert _Engine.made[1] != _Engine.made[0]
    for d in taken + [extra]:
        pool.give_back(d)


def test_named_devices_and_the_multi_stream_worker(stand_in, monkeypatch):
    assert [e.device for e in D.open_engines(all_devices=True)] == list(range(8))
    monkeypatch.setenv("FALCON_AMD_DEVICES", "2,5")
    assert [e.device for e in D.open_engines()] == [2, 5]
    monkeypatch.setenv("FALCON_AMD_DEVICES", "all")
    assert len(D.open_engines()) == 8


def test_slots_that_cannot_be_locked_do_not_fail_the_job(tmp_path, monkeypatch):
    """another user's lock directory (or none at all) must not end a consensus job before it
    starts -- the device is then chosen by pid alone, without an exception."""
    import os
    from falcon_amd import devices
    not_a_dir = tmp_path / "a_file"
    not_a_dir.write_text("x")
    monkeypatch.setenv("FALCON_AMD_LOCK_DIR", str(not_a_dir / "locks"))   # makedirs fails: ENOTDIR
    assert devices._lock_dir() is None
    assert devices._try_lock(0, 0) is None
    got = devices.choose_device([0, 1, 2, 3])
    assert got == [0, 1, 2, 3][os.getpid() % 4]
    assert devices.choose_device([0, 1, 2, 3], idle_only=True) is None
    # a directory that is there is used as it is, and its slot files are open to every user
    d = tmp_path / "locks"
    monkeypatch.setenv("FALCON_AMD_LOCK_DIR", str(d))
    assert devices._try_lock(7, 0) is True
    assert (os.stat(d).st_mode & 0o1777) == 0o1777
    assert (os.stat(d / "falcon_amd.dev7.slot0").st_mode & 0o666) == 0o666
