"""The N>1 path on CPU: two processes, gloo backend, the exact GradSync / FlatGrads code bench.py uses under RCCL.

What is checked (no kernel runs: gradients are filled by hand, as the HIP kernels would fill ``param.grad``):
  * every ``param.grad`` is a view into one flat buffer per network, with the parameter's own strides;
  * ``sync('d')`` averages ONLY the discriminator buffer, ``sync('g')`` only the generator's;
  * the 12 parameters that never receive a gradient stay zero and do not stall anything;
  * ``broadcast_parameters`` / ``broadcast_buffers`` make rank 1 identical to rank 0;
  * ``FusedAdam.zero_grad`` keeps the views attached;
  * overlap: the first backward of a kind records the order in which kernels obtain their gradient destinations
    (``ops.grad_buffer``); in the next one every bucket is all-reduced as soon as the recorded sequence has passed it, the result is
    identical to the post-backward reduction (``check_exchange`` verifies every bucket bit for bit), a backward that deviates from
    the recording falls back to the late reduction, and a write into an already-launched bucket raises.
"""
import os
import socket
import sys
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

KW = dict(forecast_steps=2, output_shape=64, latent_channels=384, context_channels=192, generation_steps=1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(1)
        import skillful_nowcasting_amd as S

        torch.manual_seed(100 + rank)  # different init per rank on purpose
        model = S.DGMR(**KW)
        sync = model.attach_data_parallel()
        # --- parameters and buffers now equal rank 0's
        flat_p = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        ref = flat_p.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(flat_p, ref), "parameters differ from rank 0 after broadcast_parameters"
        bufs = torch.cat([b.reshape(-1).float() for n, b in model.named_buffers() if b.dim() > 0 and not n.endswith("_scratch")])
        refb = bufs.clone()
        dist.broadcast(refb, src=0)
        assert torch.equal(bufs, refb), "buffers differ from rank 0 after broadcast_buffers"
        # --- grads are views of the flat buffers, in the parameter's own layout
        for fg in (sync.gen, sync.disc):
            off = 0
            for p in fg.params:
                assert p.grad is not None and p.grad.stride() == p.stride()
                assert p.grad.data_ptr() == fg.flat.data_ptr() + 4 * off
                off += p.numel()
            assert off == fg.flat.numel()
        # --- fill "gradients" the way the kernels do (in place), leave the dead parameters untouched
        dead = [n for n, _ in model.named_parameters() if (".g" in n and ".conv_1x1." in n and "up_g" not in n)
                or "d6.conv_1x1." in n or "d_last.conv_1x1." in n]
        assert len(dead) == 12, dead
        dead_ids = {id(p) for n, p in model.named_parameters() if n in dead}
        for p in list(model.generator.parameters()) + list(model.discriminator.parameters()):
            if id(p) not in dead_ids:
                p.grad.fill_(float(rank + 1))
        g_before = sync.gen.flat.clone()
        sync.sync("d")
        assert torch.equal(sync.gen.flat, g_before), "sync('d') touched the generator's gradients"
        live = sync.disc.flat != 0
        assert torch.allclose(sync.disc.flat[live], torch.full_like(sync.disc.flat[live], 1.5)), "mean over ranks of {1, 2} != 1.5"
        for p in model.discriminator.parameters():
            if id(p) in dead_ids:
                assert float(p.grad.abs().sum()) == 0.0
        sync.sync("g")
        live = sync.gen.flat != 0
        assert torch.allclose(sync.gen.flat[live], torch.full_like(sync.gen.flat[live], 1.5))
        # --- optimiser zero_grad keeps the views
        g_opt, d_opt = model.optimizers()
        d_opt.zero_grad()
        assert float(sync.disc.flat.abs().sum()) == 0.0
        p0 = sync.disc.params[0]
        assert p0.grad.data_ptr() == sync.disc.flat.data_ptr()
        # --- chunked reduce == unchunked
        sync.chunk = 1000
        sync.gen.flat.copy_(torch.arange(sync.gen.flat.numel(), dtype=torch.float32) * (rank + 1))
        sync.sync("g")
        assert torch.allclose(sync.gen.flat, torch.arange(sync.gen.flat.numel(), dtype=torch.float32) * 1.5)
        # --- overlapped exchange: buckets launched during the "backward" in the order the recorded touch sequence completes them
        from skillful_nowcasting_amd import ops

        sync.chunk = 200_000  # several buckets per network
        sync.check_exchange = True
        g_params = [p for p in sync.gen.params if id(p) not in dead_ids]
        order = list(reversed(g_params))  # a backward pass walks the layers back to front

        def backward(params, val):
            for i, p in enumerate(params):
                ops.grad_buffer(p).add_(val * (1.0 + (i % 7)))  # the touch, then the "kernel"

        nb = sync._nbuckets(sync.gen)
        assert nb >= 4
        for it in range(3):
            g_opt.zero_grad()
            sync.begin("g")
            backward(order, float(rank + 1))
            launched_early = len(sync._launched)
            sync.sync("g")
            expect = torch.zeros_like(sync.gen.flat)
            for i, p in enumerate(order):
                off = sync.gen.offset[id(p)]
                expect[off:off + p.numel()] = 1.5 * (1.0 + (i % 7))
            assert torch.allclose(sync.gen.flat, expect), f"pass {it}: overlapped exchange differs from the mean over ranks"
            if it == 0:
                assert launched_early == 0 and sync._passes["g"].recorded is not None  # recording pass: nothing launched early
            else:
                assert launched_early >= nb - 2, (it, launched_early, nb)  # all but the front buckets went out during the backward
        assert sync.stats["overlapped_buckets"] > 0 and sync.stats["deviations"] == 0
        # --- a backward in another order: falls back to the late reduction, same result
        g_opt.zero_grad()
        sync.begin("g")
        backward(g_params, float(rank + 1))  # front to back: deviates at the first touch
        sync.sync("g")
        assert sync.stats["deviations"] == 1
        expect = torch.zeros_like(sync.gen.flat)
        for i, p in enumerate(g_params):
            off = sync.gen.offset[id(p)]
            expect[off:off + p.numel()] = 1.5 * (1.0 + (i % 7))
        assert torch.allclose(sync.gen.flat, expect)
        # --- a write into a bucket whose all-reduce is already under way must raise, never pass silently
        g_opt.zero_grad()
        sync.begin("g")
        backward(order, 1.0)
        raised = False
        try:
            ops.grad_buffer(order[0])  # the last layer again, long after its bucket went out
        except RuntimeError as e:
            raised = "after its all-reduce had been launched" in str(e)
        assert raised
        sync.sync("g")
        # --- buffers are views of one flat tensor: ONE broadcast, and the modules see the result
        with torch.no_grad():
            for b in sync._buffers:
                b.add_(float(rank))
        sync.broadcast_buffers()
        bufs = torch.cat([b.reshape(-1).float() for b in sync._buffers])
        refb = bufs.clone()
        dist.broadcast(refb, src=0)
        assert torch.equal(bufs, refb)
        assert all(b.data_ptr() >= sync._buf_flat.data_ptr() and b.data_ptr() < sync._buf_flat.data_ptr() + 4 * sync._buf_flat.numel()
                   for b in sync._buffers)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


@pytest.mark.timeout(300)
def test_gradsync_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def _worker_single(port, q):
    """One rank, force_exchange: every collective still runs (over one rank they are the identity) - the path tests/test_gpu_rccl.py
    drives through RCCL on the GPU, here over gloo on CPU tensors."""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=0, world_size=1)
        torch.set_num_threads(1)
        import skillful_nowcasting_amd as S
        from skillful_nowcasting_amd import ddp

        torch.manual_seed(100)
        model = S.DGMR(**KW)
        plain = ddp.GradSync(model)  # world 1 without force_exchange: nothing is exchanged
        assert plain.world == 1 and not plain.exchange
        sync = model.attach_data_parallel(chunk_mb=1, overlap=True, force_exchange=True)
        assert sync.world == 1 and sync.exchange
        sync.check_exchange = True
        for which, net in (("d", model.discriminator), ("g", model.generator)):
            params = [p for p in net.parameters() if p.requires_grad]
            for rep in range(2):  # first pass records the touch order (every bucket late), the second launches during the "backward"
                sync.flat_for(which).zero_()
                sync.begin(which)
                from skillful_nowcasting_amd import ops

                expect = []
                for i, p in enumerate(params):
                    g = ops.grad_buffer(p)
                    g.add_(float(i + 1) * 0.5 + rep)
                    expect.append(g.detach().clone())
                sync.sync(which)
                for p, e in zip(params, expect):
                    assert torch.equal(p.grad, e), "a one-rank exchange changed a gradient"
        st = sync.stats
        assert st["late_buckets"] > 0 and st["overlapped_buckets"] > 0 and st["deviations"] == 0, st
        sync.broadcast_buffers()
        dist.barrier()
        dist.destroy_process_group()
        q.put("ok")
    except Exception:
        q.put(traceback.format_exc())


def test_gradsync_one_rank_force_exchange_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_single, args=(_free_port(), q))
    p.start()
    msg = q.get(timeout=300)
    p.join(60)
    assert msg == "ok", msg
