"""
Multi-process tests of the env-sharding helpers on CPU (gloo backend, world_size 2): contiguous shard ranges and
the all-gather of per-rollout results, with even and uneven shard sizes.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simurlacra_amd.sampling import distributed as D


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1000, 2**21):
        for ws in (1, 2, 3, 8):
            ranges = [D.shard_range(n, r, ws) for r in range(ws)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        start, stop = D.shard_range(n_total, rank, world)
        # "returns" of the global rollouts start..stop: a function of the GLOBAL index only
        ret = torch.arange(start, stop, dtype=torch.float64)**2
        length = torch.arange(start, stop, dtype=torch.int32) % 7
        dp = torch.stack([torch.arange(start, stop, dtype=torch.float64), -torch.arange(start, stop, dtype=torch.float64)], dim=1)
        g_ret, g_len, g_dp = D.all_gather_rows(ret, n_total), D.all_gather_rows(length, n_total), D.all_gather_rows(dp, n_total)
        ok = (torch.equal(g_ret, torch.arange(n_total, dtype=torch.float64)**2)
              and torch.equal(g_len, torch.arange(n_total, dtype=torch.int32) % 7)
              and torch.equal(g_dp[:, 1], -torch.arange(n_total, dtype=torch.float64)))
        q.put((rank, bool(ok), D.world()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [10, 11])
def test_all_gather_rows_world_size_2(n_total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2][1] == 2 for r in res)
