"""N>1 path on CPU: stream sharding + the result gather over torch.distributed (gloo, world 2).
The tracker itself is not involved (no GPU here); ranks carry deterministic fake result rows."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from boxmot_amd.streams import gather_results, pack_results, shard_streams


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows_for(stream, t, max_dets):
    rng = np.random.default_rng(1000 * stream + t)
    n = int(rng.integers(0, max_dets + 1))
    return rng.uniform(0, 100, (n, 8)).astype(np.float32)


def _worker(rank, world, port, total_streams, T, max_dets, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_streams(total_streams, rank, world)
    rows = np.zeros((len(mine), T, max_dets, 8), dtype=np.float32)
    cnts = np.zeros((len(mine), T), dtype=np.int32)
    for t in range(T):
        b, c = pack_results([_rows_for(s, t, max_dets) for s in mine], max_dets)
        rows[:, t], cnts[:, t] = b, c
    g_rows, g_cnts = gather_results(torch.from_numpy(rows), torch.from_numpy(cnts), dst=0)
    if rank == 0:
        ok = True
        for r in range(world):
            for k, s in enumerate(shard_streams(total_streams, r, world)):
                for t in range(T):
                    want = _rows_for(s, t, max_dets)
                    n = int(g_cnts[r][k, t])
                    ok &= n == len(want) and np.array_equal(g_rows[r][k, t, :n].numpy(), want)
        q.put(ok)
    else:
        assert g_rows is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_streams_partition():
    for total, world in ((8, 2), (7, 3), (64, 8), (3, 4)):
        parts = [shard_streams(total, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(total))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_gather_results_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, 3, 16, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
