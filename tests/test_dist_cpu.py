"""world_size-2 gloo test of the only multi-GPU exchange on the path: sequences are
sharded per rank, throughput is reduced with {SUM scans, MAX seconds}."""
import os
import socket

import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from cfear_radarodometry_code_public_amd.dist import reduce_throughput, shard_sequences
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_sequences(7, rank, world)
    scans, seconds = reduce_throughput(len(mine) * 10, 1.0 + rank)
    q.put((rank, mine, scans, seconds))
    dist.destroy_process_group()


def test_shard_and_reduce_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    for _, _, scans, seconds in res:
        assert scans == 70.0 and seconds == 2.0  # SUM over ranks, MAX over ranks


def test_single_process_passthrough():
    from cfear_radarodometry_code_public_amd.dist import reduce_throughput, shard_sequences
    assert reduce_throughput(5, 2.5) == (5.0, 2.5)
    assert shard_sequences(4, 0, 1) == [0, 1, 2, 3]
