"""world_size-2 test of the N>1 path on CPU (gloo): contiguous sharding of piles,
order-preserving gather and the timing reduction bench.py relies on.  The per-pile
work is served by the CPU oracle here (test harness); on the GPU box every rank
would own an Engine instead."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from falcon_amd.multigpu import partition, shard


def test_partition_is_balanced_and_covers():
    for n in (0, 1, 5, 8, 1000, 1001):
        for world in (1, 2, 3, 8):
            parts = [partition(n, r, world) for r in range(world)]
            assert [i for p in parts for i in p] == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, piles, expected, q):
    import torch.distributed as dist
    from falcon_amd.multigpu import gather_in_order, reduce_measurement
    from oracle.pyoracle import Port
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard(piles, rank, world)
    port_impl = Port()
    local = [port_impl.generate_consensus(p, 4, 8, 0.70)[0] for p in mine]
    bases = float(sum(len(c) for c in local))
    dist.barrier()
    tot_bases, tot_piles, tmax = reduce_measurement(bases, float(len(mine)), 1.0 + rank)
    everything = gather_in_order(local, rank, world)
    ok = (everything == expected and tot_piles == len(piles) and
          tot_bases == float(sum(len(c) for c in expected)) and tmax == float(world))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_ranks_gloo(port):
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    piles = []
    for seed in range(5):
        s, rd = make_pile(200 + seed, S=1500, coverage=12, min_read=1000, mean_read=1200, sd_read=200)
        piles.append([codes_to_str(x) for x in pile_to_seqs(s, rd)])
    expected = [port.generate_consensus(p, 4, 8, 0.70)[0] for p in piles]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, p_, piles, expected, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
