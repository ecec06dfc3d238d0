"""Multi-rank path on CPU: two gloo ranks, frames sharded by rank, the
LoRA-gradient mean all-reduce of image_segmentation_amd.dp.LoRAGradReducer.  The per-rank
fwd+bwd is the oracle's (CPU restatement; the HIP trunk needs a GPU) -- what is under test is
the exchange: after reduce() every rank must hold the gradient of the full global batch, the
semantics of the reference's DistributedDataParallel wrap (train_lora_sam.py:126,172)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import TINY, rel_err

IMG = 128          # tiny trunk geometry at 128x128: 32x32 tokens in stage 1
PER_RANK = 2


def _lora_names(p):
    return sorted(k for k in p if "linear_a_" in k or "linear_b_" in k)


def _grads(frames):
    from oracle import hiera_oracle as O
    from oracle.params import make_params
    spec = O.block_specs(**TINY)
    p = make_params(spec, 4, 0)
    loss = O.fpn_loss(O.image_encoder_forward(p, frames, spec))
    loss.backward()
    names = _lora_names(p)
    return names, p, float(loss.detach())


def _frames(first, n):
    from image_segmentation_amd.synth import synth_frames
    return synth_frames(first + n, IMG, 0)[first:]      # frame i is the same tensor on every rank


def _worker(rank, world, port, q, overlap):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_segmentation_amd.dp import LoRAGradReducer
        names, p, _ = _grads(_frames(rank * PER_RANK, PER_RANK))
        params = [torch.nn.Parameter(p[n].detach().clone()) for n in names]
        for prm, n in zip(params, names):
            prm.grad = p[n].grad.clone()
        if rank == 1:                       # rank 1 starts from perturbed LoRA factors ...
            with torch.no_grad():
                for prm in params:
                    prm.add_(1.0)
        red = LoRAGradReducer(params)
        assert len(red.buckets) == 2 and all(len(b.params) == 4 for b in red.buckets)   # one bucket per LoRA block
        red.broadcast_params(0)             # ... and must end up with rank 0's
        if overlap:
            # what Hiera's backward does through the hook that attach() registers: the LAST block's four
            # gradients first, then the block before it; autograd assigns .grad only afterwards
            local = [prm.grad for prm in params]
            for prm in params:
                prm.grad = None
            if overlap == "accum":
                # gradient accumulation: two backwards before reduce(); DDP sums the micro-batches
                for frac in (0.25, 0.75):
                    for lo in (4, 0):
                        red.on_bucket(params[lo:lo + 4], [frac * g for g in local[lo:lo + 4]])
                assert all(len(b.pending) == 1 for b in red.buckets)
            else:
                for lo in (4, 0):
                    red.on_bucket(params[lo:lo + 4], local[lo:lo + 4])
            assert all(b.work is not None for b in red.buckets)
            for prm, g in zip(params, local):
                prm.grad = g
        red.reduce()
        q.put((rank, {n: prm.grad.clone().numpy() for n, prm in zip(names, params)},
               {n: prm.detach().clone().numpy() for n, prm in zip(names, params)}))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("overlap", [False, True, "accum"])
def test_two_rank_lora_grad_allreduce_matches_global_batch(overlap):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, overlap)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    # single-process ground truth: gradient of the loss over the whole global batch
    names, p, _ = _grads(_frames(0, world * PER_RANK))
    for rank, grads, params in got:
        for n in names:
            assert rel_err(grads[n], p[n].grad) <= 1e-5, (rank, n)
            assert rel_err(params[n], p[n].detach()) == 0.0, (rank, n)   # broadcast from rank 0
    # both ranks hold bit-identical reduced gradients
    (_, g0, _), (_, g1, _) = sorted(got, key=lambda t: t[0])
    for n in names:
        assert (g0[n] == g1[n]).all()


def test_single_process_reduce_is_identity():
    from image_segmentation_amd.dp import LoRAGradReducer
    w = torch.nn.Parameter(torch.randn(4, 8))
    w.grad = torch.randn(4, 8)
    g = w.grad.clone()
    red = LoRAGradReducer([w])
    red.broadcast_params(0)
    red.reduce()
    assert torch.equal(w.grad, g)
    frozen = torch.nn.Parameter(torch.randn(3), requires_grad=False)
    with pytest.raises(ValueError):
        LoRAGradReducer([frozen])
