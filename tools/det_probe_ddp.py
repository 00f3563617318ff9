"""Two ranks on one GPU over gloo (tests/test_gpu_ddp.py's setting), twice: the LOCAL gradients each rank holds right before every
exchange, compared between the two runs - which parameters' gradients are not bit-reproducible when two processes share the GPU?"""
import os
import socket
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.multiprocessing as mp

KW = dict(forecast_steps=2, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)
STEPS = int(os.environ.get("PROBE_STEPS", "1"))


def worker(rank, world, port, out, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), GPU_MAX_HW_QUEUES="4")  # (two ranks on ONE device)
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        import skillful_nowcasting_amd as S
        from skillful_nowcasting_amd import ddp

        S.set_precision("mixed")
        torch.manual_seed(100 + rank)
        model = S.DGMR(**KW).to("cuda")
        sync = model.attach_data_parallel(chunk_mb=8, overlap=False)
        snaps = {}
        orig = ddp.GradSync.sync

        def spy(self, which):
            torch.cuda.synchronize()
            fg = self.flat_for(which)
            n = sum(1 for k in snaps if k.startswith(which))
            snaps[f"{which}{n}"] = fg.flat.detach().clone().cpu()
            snaps[f"state_before_sync_{which}{n}"] = {k: v.detach().clone().cpu() for k, v in model.state_dict().items()}
            return orig(self, which)

        ddp.GradSync.sync = spy
        og = model._gen_losses

        def gl(*a, **k):
            out = og(*a, **k)
            torch.cuda.synchronize()
            snaps["state_at_gen_losses"] = {"pred": a[2].detach().clone().cpu(), "gen_loss": out[0].detach().clone().cpu(), "grid": out[1].detach().clone().cpu()}
            snaps["state_at_gen_losses"].update({k: v.detach().clone().cpu() for k, v in model.state_dict().items()})
            return out

        model._gen_losses = gl
        # gradient of every module output, in the order the backward pass produces them (generator pass only: armed inside _gen_losses
        # and by the checkpoint's recompute, which runs the modules again with grad enabled)
        grad_rec = []
        names = {m: n for n, m in model.named_modules()}

        def fwd_hook(m, inp, o):
            outs = o if isinstance(o, (tuple, list)) else (o,)
            for j, t in enumerate(outs):
                if torch.is_tensor(t) and t.requires_grad and ARM[0]:
                    t.register_hook(lambda g, nm=f"{names[m]}[{j}]": grad_rec.append((nm, g.detach().float().clone().cpu())))

        ARM = [False]
        for m in model.modules():
            m.register_forward_hook(fwd_hook)
        omb = model.manual_backward

        def mb(loss):
            return omb(loss)

        og2 = model._generate

        def gen(images, draws, grad):
            ARM[0] = bool(grad)
            return og2(images, draws, grad)

        model._generate = gen
        torch.manual_seed(200 + rank)
        x = torch.rand(2, 4, 1, 128, 128, device="cuda")
        y = torch.rand(2, 2, 1, 128, 128, device="cuda")
        torch.manual_seed(300)
        for i in range(STEPS):
            model.training_step((x, y), i)
        torch.cuda.synchronize()
        names = {"g": [(n, p.numel()) for n, p in model.generator.named_parameters() if p.requires_grad],
                 "d": [(n, p.numel()) for n, p in model.discriminator.named_parameters() if p.requires_grad]}
        snaps["state"] = {k: v.detach().clone().cpu() for k, v in model.state_dict().items()}
        torch.save({"snaps": snaps, "names": names, "grad_rec": grad_rec}, f"{out}_rank{rank}.pt")
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


def run(tag):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ps = [ctx.Process(target=worker, args=(r, 2, port, tag, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=500) for _ in ps]
    for p in ps:
        p.join(60)
    for r, m in res:
        assert m == "ok", m


def compare(out):
    for rank in range(2):
        a, b = torch.load(os.path.join(out, f"a_rank{rank}.pt")), torch.load(os.path.join(out, f"b_rank{rank}.pt"))
        for key in a["snaps"]:
            if key.startswith("state"):
                bad = [k for k in a["snaps"][key] if not torch.equal(a["snaps"][key][k], b["snaps"][key][k])]
                print(f"rank {rank} {key}: {len(bad)} tensors differ", bad[:6])
                continue
            fa, fb = a["snaps"][key], b["snaps"][key]
            off, bad = 0, []
            for n, numel in a["names"][key[0]]:
                if not torch.equal(fa[off:off + numel], fb[off:off + numel]):
                    e = (fa[off:off + numel] - fb[off:off + numel]).abs().max().item() / max(fb[off:off + numel].abs().max().item(), 1e-30)
                    bad.append((n, f"{e:.1e}"))
                off += numel
            print(f"rank {rank} local gradients before exchange {key}: {len(bad)} parameters differ", bad[:4])
        ga, gb = a.get("grad_rec", []), b.get("grad_rec", [])
        shown = 0
        for idx_, ((na, ta), (nb_, tb)) in enumerate(zip(ga, gb)):
            if na != nb_ or ta.shape != tb.shape:
                print(f"rank {rank} backward order differs at {na} / {nb_}")
                break
            if not torch.equal(ta, tb):
                e = (ta.double() - tb.double()).abs().max().item() / max(tb.double().abs().max().item(), 1e-300)
                if shown == 0:
                    print(f"rank {rank} identical before it:", [n_.replace("discriminator.", "D.") for n_, _ in ga[max(0, idx_ - 4):idx_]], "next:", [n_ for n_, _ in ga[idx_:idx_ + 3]])
                    d = (ta.double() - tb.double()).abs()
                    nz = (d > 0).nonzero()
                    if nz.shape[1] == 4:
                        pix = sorted({(int(i[0]), int(i[2]), int(i[3])) for i in nz})
                        print(f"rank {rank} differing pixels (n, h, w): {pix[:16]}; channels per pixel:", [int(((nz[:, 0] == a_) & (nz[:, 2] == b_) & (nz[:, 3] == c_)).sum()) for a_, b_, c_ in pix[:16]])
                        print(f"rank {rank} differing channels:", sorted({int(i[1]) for i in nz})[:48])
                    print(f"rank {rank} first differing tensor: {nz.shape[0]} of {d.numel()} elements differ; index ranges per dim:",
                          [(int(nz[:, k].min()), int(nz[:, k].max())) for k in range(nz.shape[1])], "strides", ta.stride())
                    big = (d > 0.01 * tb.abs().max()).nonzero()
                    print(f"rank {rank} elements off by > 1 % of max: {big.shape[0]}", big[:12].tolist())
                print(f"rank {rank} GRAD-OUT DIFF #{shown} {na:60s} shape {tuple(ta.shape)} rel {e:.2e}")
                shown += 1
                if shown >= 6:
                    break
        print(f"rank {rank}: {len(ga)} module-output gradients recorded, {shown} shown as differing")
        os.remove(os.path.join(out, f"a_rank{rank}.pt"))
        os.remove(os.path.join(out, f"b_rank{rank}.pt"))


if __name__ == "__main__":
    out = os.path.join(ROOT, "gpurun_out", "r5h")
    os.makedirs(out, exist_ok=True)
    for rep in range(int(os.environ.get("PROBE_REPS", "3"))):
        print("=== repetition", rep)
        run(os.path.join(out, "a"))
        run(os.path.join(out, "b"))
        compare(out)
