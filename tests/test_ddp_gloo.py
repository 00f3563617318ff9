"""The N>1 path on CPU: two processes, gloo backend, the exact GradSync / FlatGrads code bench.py uses under RCCL.

What is checked (no kernel runs: gradients are filled by hand, as the HIP kernels would fill ``param.grad``):
  * every ``param.grad`` is a view into one flat buffer per network, with the parameter's own strides;
  * ``sync('d')`` averages ONLY the discriminator buffer, ``sync('g')`` only the generator's;
  * the 12 parameters that never receive a gradient stay zero and do not stall anything;
  * ``broadcast_parameters`` / ``broadcast_buffers`` make rank 1 identical to rank 0;
  * ``FusedAdam.zero_grad`` keeps the views attached.
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
