"""CPU, world_size 2 over gloo: the N > 1 path (batch sharding + the single flat gradient
all-reduce that replaces DDP's bucketed one for the few-MB trainable set)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from asr_amd.distributed import FlatGradAllReducer, shard_batch, shard_range

        torch.manual_seed(0)  # identical replicas
        model = torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.GELU(), torch.nn.Linear(16, 8))
        red = FlatGradAllReducer(model.parameters())
        assert red.numel == sum(p.numel() for p in model.parameters())
        g = torch.Generator().manual_seed(1)
        x_all = torch.randn(10, 12, generator=g)
        y_all = torch.randn(10, 8, generator=g)
        idx = list(shard_range(10, rank, world))
        assert shard_batch(list(range(10))) == idx
        # local loss = SUM over local samples; after averaging, multiply by world/N to get the global mean loss grad
        loss = ((model(x_all[idx]) - y_all[idx]) ** 2).sum()
        red.zero_()
        loss.backward()
        work = red.all_reduce(async_op=True)
        red.wait()
        assert work is not None
        flat = red.flat.clone() * world / 10.0
        # single-process reference on the whole batch
        torch.manual_seed(0)
        ref = torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.GELU(), torch.nn.Linear(16, 8))
        ((ref(x_all) - y_all) ** 2).sum().div(10.0).backward()
        ref_flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
        assert torch.allclose(flat, ref_flat, atol=1e-6), (flat - ref_flat).abs().max()
        # .grad tensors are views of the flat buffer (no copies around the collective)
        assert all(p.grad.data_ptr() >= red.flat.data_ptr() for p in model.parameters())
        # survives zero_grad(set_to_none=True)
        torch.optim.SGD(model.parameters(), lr=0.1).zero_grad(set_to_none=True)
        red.rebind()
        assert all(p.grad is not None for p in model.parameters())
        out.put((rank, True, ""))
    except Exception as e:  # pragma: no cover
        out.put((rank, False, repr(e)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_shard_range_partitions_everything():
    from asr_amd.distributed import shard_range

    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in shard_range(n, r, w)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(120)
def test_flat_grad_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(ok for _, ok, _ in res), res
