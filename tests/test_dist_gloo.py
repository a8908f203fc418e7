"""world_size-2 data-parallel logic on CPU (gloo): ray sharding + global-mean weights + gradient all-reduce give the
single-process gradient of the concatenated batch.  The HIP engine uses exactly these helpers
(engine.NAFEngine.all_reduce_grads / dist.global_mean_weights) with backend "nccl" (RCCL) on the GPUs."""
import os

import numpy as np
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from neuralvolumetricreconstructionformedicalimages_amd import dist


def _model_grads(params, x, y, w):
    """A small stand-in 'renderer' (the real one needs a GPU): acc = sigmoid(x @ W1) @ w2."""
    W1, w2 = (p.clone().requires_grad_(True) for p in params)
    acc = torch.sigmoid(x @ W1) @ w2
    loss = (w * (acc - y) ** 2).sum()
    loss.backward()
    return [W1.grad, w2.grad], loss.detach()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _, group = dist.init_from_env(device_type="cpu")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    n = 101
    x, y = torch.randn(n, 6, generator=g), torch.randn(n, generator=g)
    mask = torch.rand(n, generator=g) > 0.25
    params = [torch.randn(6, 5, generator=g), torch.randn(5, generator=g)]
    if rank != 0:                                   # ranks start from different weights until the broadcast
        params = [p + 1.0 for p in params]
    dist.broadcast_parameters(params, group)
    b, e = dist.shard_range(n, rank, world)
    weights = dist.global_mean_weights(mask[b:e], group)
    grads, loss = _model_grads(params, x[b:e], y[b:e], weights)
    loss = loss.reshape(1)
    dist.all_reduce_sum_(grads + [loss], group)
    q.put((rank, [t.numpy() for t in grads], float(loss)))
    td.barrier()
    td.destroy_process_group()


def test_dp_gradients_equal_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    n = 101
    x, y = torch.randn(n, 6, generator=g), torch.randn(n, generator=g)
    mask = torch.rand(n, generator=g) > 0.25
    params = [torch.randn(6, 5, generator=g), torch.randn(5, generator=g)]
    ref_grads, ref_loss = _model_grads(params, x, y, dist.global_mean_weights(mask, None))
    for rank, grads, loss in results:
        np.testing.assert_allclose(loss, float(ref_loss), rtol=1e-5)
        for a, b in zip(grads, ref_grads):
            np.testing.assert_allclose(a, b.numpy(), rtol=1e-4, atol=1e-6)
