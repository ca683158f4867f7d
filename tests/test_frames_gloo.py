"""N>1 path on CPU: world_size-2 gloo processes exercise frame sharding, the final gather and
the bucketed gradient all-reduce of gaussiancity_amd.frames (same code the RCCL runs use)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaussiancity_amd import frames


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_frame(f):
    return torch.full((3, 4, 5), float(f)) + torch.arange(60, dtype=torch.float32).reshape(3, 4, 5) / 100


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rendered = []

        def render(f):
            rendered.append(f)
            return _fake_frame(f)

        out = frames.render_sharded(render, 7)          # 7 frames on 2 ranks: ragged last round
        ok_gather = sorted(out) == list(range(7)) and all(torch.equal(out[f], _fake_frame(f)) for f in out)
        local_only = frames.render_sharded(_fake_frame, 7, gather=False)
        # uint8 [H,W,3] frames gathered instead of fp32 [3,H,W]: a quarter of the bytes, same frames
        u8 = frames.render_sharded(lambda f: _fake_frame(f) / 8.0 - 0.5, 7, as_uint8=True)
        ok_gather = ok_gather and sorted(u8) == list(range(7)) and all(
            u8[f].dtype == torch.uint8 and tuple(u8[f].shape) == (4, 5, 3)
            and torch.equal(u8[f], frames.InferenceLoop.to_uint8_hwc(_fake_frame(f) / 8.0 - 0.5)) for f in u8)
        # gradient exchange: three tensors, tiny buckets so that several messages are used
        g = [torch.full((1000,), float(rank + 1)), torch.full((17, 3), float(10 * (rank + 1))),
             torch.full((5,), float(-rank))]
        nb = frames.allreduce_gradients(g, bucket_bytes=4096)
        # ONE tensor much larger than the bucket (the C4 stand-in shape): reduced in place through views of
        # itself -- several messages, no staging copy (the storage pointer must not change), ragged tail
        big = torch.arange(10_007, dtype=torch.float32) * (rank + 1)
        ptr = big.data_ptr()
        nb_big = frames.allreduce_gradients([torch.full((3,), float(rank)), big, torch.full((2,), 4.0 * rank)],
                                            bucket_bytes=4096)
        big_ok = (big.data_ptr() == ptr and
                  bool(torch.allclose(big, torch.arange(10_007, dtype=torch.float32) * 1.5)))
        q.put((rank, rendered, ok_gather, sorted(local_only), nb, [float(t.reshape(-1)[0]) for t in g],
               nb_big, big_ok))
    finally:
        dist.destroy_process_group()


def test_frame_sharding_and_gradient_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]        # round-robin ownership
    assert res[0][2] and res[1][2]                                      # every rank has all frames
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5]
    assert res[0][4] == res[1][4] >= 2                                  # several buckets
    want = [1.5, 15.0, -0.5]                                            # averages over the 2 ranks
    assert np.allclose(res[0][5], want) and np.allclose(res[1][5], want)
    # 10007 floats in 1024-float views = 10 in-place messages + the two small buckets around it
    assert res[0][6] == res[1][6] == 12 and res[0][7] and res[1][7]


def test_single_process_paths():
    assert frames.shard_frames(24, 3, 8) == [3, 11, 19]
    assert frames.shard_frames(5, 7, 8) == []
    out = frames.render_sharded(_fake_frame, 3, rank=0, world=1)
    assert sorted(out) == [0, 1, 2]
    assert frames.allreduce_gradients([torch.ones(3)]) == 0


class _FakeWrapper:
    """Differentiable stand-in for GaussianRasterizerWrapper on CPU: image = f(points, pose)."""
    device = torch.device("cpu")

    def __call__(self, points, cam_pos, cam_quat):
        base = points[:, :3].mean() + points[:, 11:14].mean(dim=0).view(3, 1, 1)
        ramp = torch.linspace(0, 1, 12).view(1, 3, 4) * float(np.sum(cam_pos))
        return (base + ramp).expand(3, 3, 4)


def test_train_step_harness_optimizer_and_inference_loop():
    pts = torch.randn(50, 14, requires_grad=True)
    before = pts.detach().clone()
    h = frames.TrainStepHarness(_FakeWrapper(), n_param=1000, crop=(0, 0, 4, 3), lr=1e-2)
    target = torch.zeros(3, 3, 4)
    losses = []
    for i in range(5):  # step() clears points.grad itself (the reference zero_grads inside its G-step)
        loss, img, _ = h.step(pts, np.array([1.0, 2.0, 3.0]), np.array([0, 0, 0, 1.0]), target)
        losses.append(float(loss))
    assert not torch.equal(pts.detach(), before) and losses[-1] < losses[0]      # Adam moved the points downhill
    assert float(h.param_grad[0]) == float(h.param_grad[-1]) != 0.0
    # inference loop: order, conversion and buffer reuse
    poses = [(np.array([i, 0.0, 0.0]), np.array([0, 0, 0, 1.0])) for i in range(5)]
    loop = frames.InferenceLoop(lambda p, cp, cq: torch.full((3, 2, 2), float(cp[0]) / 2 - 1.0))
    got = loop.run(None, poses)
    assert len(got) == 5 and got[0].shape == (2, 2, 3) and got[0].dtype == np.uint8
    assert [int(f[0, 0, 0]) for f in got] == [0, 63, 127, 191, 255]
    seen = []
    loop.run(None, poses, consume=lambda i, f: seen.append((i, int(f[0, 0, 0]))))
    assert seen == [(0, 0), (1, 63), (2, 127), (3, 191), (4, 255)]
    for ns in (1, 2, 4, 7):  # any ring size keeps the order, also one longer than the pose list
        seen = []
        frames.InferenceLoop(lambda p, cp, cq: torch.full((3, 2, 2), float(cp[0]) / 2 - 1.0), n_streams=ns).run(
            None, poses, consume=lambda i, f: seen.append((i, int(f[0, 0, 0]))))
        assert seen == [(0, 0), (1, 63), (2, 127), (3, 191), (4, 255)], ns


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        gen = frames.StandInGenerator(n_param=4 * 5000 + 3, n_layers=4)
        with torch.no_grad():  # away from the identity so that every layer's gradient is non-trivial
            for i, p in enumerate(gen.layers):
                p[:28] = 0.01 * (i + 1) * torch.arange(28, dtype=torch.float32)
        step = frames.DDPTrainStep(_FakeWrapper(), gen, crop=(0, 0, 4, 3), bucket_cap_mb=0.02)  # several buckets
        base = torch.linspace(-1, 1, 50 * 14).view(50, 14) * (rank + 1)      # another frame on every rank
        pose = np.array([1.0 + rank, 2.0, 3.0])
        loss, img = step.step(base, pose, np.array([0, 0, 0, 1.0]), torch.zeros(3, 3, 4))
        grads = [p.grad.clone() for p in gen.layers]
        # the same step without DDP on both ranks' data: the rank-averaged gradient DDP must have produced
        want = [torch.zeros_like(p) for p in gen.layers]
        for r in range(world):
            ref = frames.StandInGenerator(n_param=4 * 5000 + 3, n_layers=4)
            ref.load_state_dict(gen.state_dict())
            s2 = frames.DDPTrainStep.__new__(frames.DDPTrainStep)
            s2.rw, s2.crop, s2.generator, s2.net, s2.opt = _FakeWrapper(), (0, 0, 4, 3), ref, ref, None
            s2.step(torch.linspace(-1, 1, 50 * 14).view(50, 14) * (r + 1), np.array([1.0 + r, 2.0, 3.0]),
                    np.array([0, 0, 0, 1.0]), torch.zeros(3, 3, 4))
            for w, p in zip(want, ref.layers):
                w += p.grad / world
        ok = all(torch.allclose(g, w, rtol=1e-5, atol=1e-7) for g, w in zip(grads, want))
        dense = all(g.shape == p.shape and float(g[28:].abs().max()) == 0.0 and float(g[:28].abs().max()) > 0
                    for g, p in zip(grads, gen.layers))
        q.put((rank, ok, dense, isinstance(step.net, torch.nn.parallel.DistributedDataParallel), img.shape))
    finally:
        dist.destroy_process_group()


def test_ddp_train_step_world2_gloo():
    """frames.DDPTrainStep: the generator stand-in runs under torch's DistributedDataParallel (the reference's wiring,
    core/train.py:78-87) -- its bucketed all-reduce, fired by autograd hooks during the backward, leaves the
    rank-averaged dense gradients on every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for rank, ok, dense, is_ddp, shape in res:
        assert ok and dense and is_ddp and tuple(shape) == (3, 3, 4), (rank, ok, dense, is_ddp, shape)


def _ddp_d_step_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        gen = frames.StandInGenerator(n_param=2 * 3000, n_layers=2)
        dis = frames.StandInDiscriminator(n_param=7001)
        with torch.no_grad():
            dis.p[0], dis.p[1] = 0.25, -0.1
        calls = []

        class _Spy(_FakeWrapper):  # records whether a render ran under no_grad (the D-step's forward-only frame)
            def __call__(self, points, cam_pos, cam_quat):
                calls.append(torch.is_grad_enabled())
                return super().__call__(points, cam_pos, cam_quat)

        step = frames.DDPTrainStep(_Spy(), gen, crop=(0, 0, 4, 3), bucket_cap_mb=0.01, discriminator=dis)
        base = torch.linspace(-1, 1, 50 * 14).view(50, 14) * (rank + 1)
        pose, quat, target = np.array([1.0 + rank, 2.0, 3.0]), np.array([0, 0, 0, 1.0]), torch.full((3, 3, 4), 0.3)
        loss, img = step.step(base, pose, quat, target)
        d_grad = dis.p.grad.clone()
        # the same D-step without DDP on both ranks' data: the rank-averaged gradient
        want = torch.zeros_like(dis.p)
        for r in range(world):
            d2 = frames.StandInDiscriminator(n_param=7001)
            d2.load_state_dict(dis.state_dict())
            g2 = frames.StandInGenerator(n_param=2 * 3000, n_layers=2)
            s2 = frames.DDPTrainStep(_FakeWrapper(), g2, crop=(0, 0, 4, 3), discriminator=d2)
            s2.net, s2.dnet = g2, d2  # (no DDP wrapper: a single-process reference of rank r's frame)
            s2.d_step(torch.linspace(-1, 1, 50 * 14).view(50, 14) * (r + 1), np.array([1.0 + r, 2.0, 3.0]), quat, target)
            want += d2.p.grad / world
        q.put((rank, bool(torch.allclose(d_grad, want, rtol=1e-5, atol=1e-7)), float(d_grad[:2].abs().max()) > 0,
               tuple(d_grad.shape) == (7001,), calls, isinstance(step.dnet, torch.nn.parallel.DistributedDataParallel),
               all(p.grad is not None for p in gen.parameters()), float(step.last_d_loss)))
    finally:
        dist.destroy_process_group()


def test_ddp_train_step_with_discriminator_world2_gloo():
    """The C4 step as the reference runs it with the discriminator enabled (core/train.py:227-295): D-step = a forward-only
    render under no_grad + the discriminator's backward with its own DDP all-reduce (73.1 MB at full size), then the G-step
    with the GAN term.  World 2 over gloo: the discriminator's gradients come out rank-averaged and dense, the first render
    of the step ran without autograd, the second with it, and the generator still gets its gradients."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_d_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for rank, ok, nonzero, dense, calls, is_ddp, gen_grads, d_loss in res:
        assert ok and nonzero and dense and is_ddp and gen_grads and d_loss > 0, (rank, ok, nonzero, dense, is_ddp, gen_grads, d_loss)
        assert calls == [False, True], calls


def test_rank_affinity_from_sysfs(tmp_path, monkeypatch):
    """gaussiancity_amd.affinity (counterpart of utils/distributed.py:19-62): cpulist parsing, the sysfs lookup on a fake
    tree, the binding restricted to the process's own mask, and the no-GPU fallbacks."""
    from gaussiancity_amd import affinity as A
    assert A.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and A.parse_cpulist("") == []
    dev = tmp_path / "0000:c1:00.0"
    dev.mkdir()
    mine = sorted(os.sched_getaffinity(0))
    (dev / "local_cpulist").write_text("%d,4000-4001\n" % mine[0])
    (dev / "numa_node").write_text("1\n")
    monkeypatch.setattr(A, "gpu_pci_address", lambda i: "0000:c1:00.0")
    cpus, node, why = A.gpu_local_cpus(0, sysfs=str(tmp_path))
    assert cpus == sorted({mine[0], 4000, 4001}) and node == 1 and why == "ok"
    try:
        r = A.bind_rank_to_gpu(0, sysfs=str(tmp_path))
        assert r == {"bound": True, "cpus": 1, "numa_node": 1, "why": "ok", "cpulist": str(mine[0]),
                     "pci": "0000:c1:00.0"} and sorted(os.sched_getaffinity(0)) == [mine[0]]
        assert A.format_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11" and A.format_cpulist([]) == ""
        summary = A.summarize_rccl_log("NCCL INFO RCCL version 2.22.3+hip7.0\nNCCL INFO Channel 00/16 : 0 1 2 3\n"
                                       "NCCL INFO Trees [0] 1/-1/-1->0->-1\nNCCL INFO Channel 00 : 0[0] -> 1[1] via P2P/IPC\n"
                                       "NCCL INFO 16 coll channels, 16 p2p channels\n")
        assert summary["coll_channels"] == 16 and summary["transports"] == {"p2p_xgmi_or_ipc": 1}
        assert summary["ring_lines"] == 1 and summary["tree_lines"] == 1 and summary["version"].startswith("2.22")
    finally:
        os.sched_setaffinity(0, mine)
    # binding before HIP exists: the GPU agents' PCI addresses from a (fake) KFD topology, CPU agents skipped,
    # *_VISIBLE_DEVICES index lists honoured
    topo = tmp_path / "kfd"
    for n, (simd, loc) in enumerate([(0, 0), (0, 0), (1024, 0xc100), (1024, 0x8b00)]):
        d = topo / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count %d\nlocation_id %d\ndomain 0\n" % (simd, loc))
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert A.kfd_gpu_pci_addresses(str(topo)) == ["0000:c1:00.0", "0000:8b:00.0"]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert A.kfd_gpu_pci_addresses(str(topo)) == ["0000:8b:00.0"]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    try:
        r = A.bind_rank_early(0, sysfs=str(tmp_path), topology=str(topo))
        assert r["bound"] is True and r["pci"] == "0000:c1:00.0" and r["when"].startswith("before HIP")
    finally:
        os.sched_setaffinity(0, mine)
    assert A.bind_rank_early(5, sysfs=str(tmp_path), topology=str(topo))["bound"] is False
    assert A.kfd_gpu_pci_addresses(str(tmp_path / "nowhere")) == []
    monkeypatch.setenv("GCR_NO_AFFINITY", "1")
    assert A.bind_rank_to_gpu(0, sysfs=str(tmp_path))["bound"] is False
    monkeypatch.delenv("GCR_NO_AFFINITY")
    monkeypatch.setattr(A, "gpu_pci_address", lambda i: None)
    assert A.bind_rank_to_gpu(0)["bound"] is False
    monkeypatch.setattr(A, "gpu_pci_address", lambda i: "0000:ff:1f.0")
    assert "sysfs" in A.bind_rank_to_gpu(0, sysfs=str(tmp_path))["why"]
