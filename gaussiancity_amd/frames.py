"""Frame loops around the rasterizer, one process per GPU.

The hot path shards by FRAME: a frame is one camera over a replicated Gaussian set, with no
cross-frame state (reference utils/helpers.py:255-260, scripts/inference.py:655-667).  So

  * inference: rank r renders frames r, r+world, ... -- no collective on the data path; frames
    are only gathered at the end (all_gather of [3,H,W] images) if the caller wants the video
    on one rank;
  * training: the reference is plain DDP (core/train.py:78-87): one frame per rank per step,
    then an all-reduce(avg) of the generator gradients (69.8M fp32 = 279 MB for the BG config,
    SURVEY.md 2.2).  `allreduce_gradients` is that collective, bucketed flat buffers over
    torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).
    The rasterizer's own backward needs no collective.

Nothing here touches CUDA/HIP directly; it works on whatever device the tensors live on, which
is what lets the world_size-2 gloo tests cover the N>1 logic on CPU.
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """Round-robin frame ownership: frame f belongs to rank f % world."""
    return list(range(rank, n_frames, world))


def render_sharded(render_fn, n_frames, rank=None, world=None, gather=True, group=None, as_uint8=False):
    """Each rank renders its own frames with `render_fn(frame_index) -> Tensor[3,H,W]`.

    Returns {frame_index: image}.  With gather=True every rank ends up with all frames
    (one all_gather of equally sized stacks; the last round is padded), which is the only
    collective and is off the per-frame critical path.  as_uint8=True converts every frame to the
    uint8 [H,W,3] the video writer wants (scripts/inference.py:665) BEFORE the gather: a quarter of the bytes
    over xGMI (a 4K frame: 24.9 MB instead of 99.5 MB).
    """
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_frames(n_frames, rank, world)
    local = {f: (InferenceLoop.to_uint8_hwc(render_fn(f)) if as_uint8 else render_fn(f)) for f in mine}
    if not gather or world == 1:
        return local
    per_rank = (n_frames + world - 1) // world
    sample = next(iter(local.values())) if local else None
    shape = torch.tensor(list(sample.shape) if sample is not None else [0, 0, 0], dtype=torch.int64)
    if sample is not None and sample.is_cuda:
        shape = shape.to(sample.device)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX, group=group)
    c, h, w = [int(v) for v in shape.tolist()]
    ref = sample if sample is not None else torch.zeros(1, device=shape.device)
    stack = torch.zeros((per_rank, c, h, w), dtype=torch.uint8 if as_uint8 else torch.float32, device=ref.device)
    for slot, f in enumerate(mine):
        stack[slot] = local[f]
    gathered = [torch.empty_like(stack) for _ in range(world)]
    dist.all_gather(gathered, stack, group=group)
    out = {}
    for r in range(world):
        for slot, f in enumerate(shard_frames(n_frames, r, world)):
            out[f] = gathered[r][slot]
    return out


def allreduce_gradients(tensors, group=None, bucket_bytes=64 << 20, average=True):
    """DDP-style gradient exchange, in place.  Returns the number of all-reduce messages issued.

    * A contiguous tensor of at least `bucket_bytes` is reduced IN PLACE through `bucket_bytes`-sized
      views of itself -- no staging copy in, no copy back (a 279 MB gradient set = five 64 MB
      messages, issued asynchronously so message k+1 is queued while k is on the links).
    * Smaller tensors are packed, in order, into one flat staging bucket of about `bucket_bytes`
      (the usual DDP bucket), reduced, and scattered back.
    Bucket size is a knob because a ring all-reduce over xGMI is per-link bound (7 links x ~153 GB/s
    per GPU): few large messages amortise the launch + protocol latency."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    scale = 1.0 / world
    messages = 0
    pending = []  # (work handle, view to scale) of the in-place messages

    def flush(bucket):
        nonlocal messages
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        messages += 1
        if average:
            flat.mul_(scale)
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n

    cur, cur_bytes = [], 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if nbytes >= bucket_bytes and t.is_contiguous():
            flush(cur)  # keep the reduction order = the tensor order on every rank
            cur, cur_bytes = [], 0
            flat = t.view(-1)
            step = max(1, bucket_bytes // t.element_size())
            for off in range(0, flat.numel(), step):
                view = flat[off:off + step]
                pending.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group, async_op=True), view))
                messages += 1
            continue
        if cur and cur_bytes + nbytes > bucket_bytes:
            flush(cur)
            cur, cur_bytes = [], 0
        cur.append(t)
        cur_bytes += nbytes
    flush(cur)
    for work, view in pending:
        work.wait()
        if average:
            view.mul_(scale)
    return messages


class TrainStepHarness:
    """Shape of the reference's G-step (core/train.py:263-295) around the rasterizer only:
    points -> rasterize (wrapper API) -> crop -> stand-in loss -> backward -> gradient
    all-reduce of a stand-in parameter set.  The generator/discriminator/VGG are out of scope
    (plain torch modules); `n_param` fp32 values stand in for their gradients so the RCCL
    message sizes match the real training step (BG generator: 69,809,101 parameters).  The
    stand-in buffer is FILLED with one scalar derived from d(loss)/d(points) -- it has the
    generator's message size, not its values; the real gradients of the step are `points.grad`."""

    def __init__(self, rasterizer_wrapper, n_param=69_809_101, crop=None, device=None, group=None, lr=None):
        self.rw = rasterizer_wrapper
        self.crop = crop
        self.group = group
        self.device = device if device is not None else rasterizer_wrapper.device
        self.param_grad = torch.zeros(n_param, dtype=torch.float32, device=self.device)
        self.lr = lr          # with a learning rate, step() also applies Adam to `points`
        self._opt = None      # (the reference steps its generator's optimizer, core/train.py:293-295)

    def step(self, points, cam_pos, cam_quat, target=None):
        """points: [N,14] leaf tensor requiring grad.  Returns (loss, image, n_messages)."""
        points.grad = None  # the reference zero_grads inside its G-step (core/train.py:263-295)
        img = self.rw(points, cam_pos, cam_quat)
        if self.crop is not None:
            x, y, w, h = self.crop
            img = img[:, y:y + h, x:x + w]
        loss = (img - target).abs().mean() if target is not None else img.abs().mean()
        loss.backward()
        # stand-in for "the generator's gradients depend on d(loss)/d(points)"
        self.param_grad.fill_(points.grad.abs().mean())
        n_buckets = allreduce_gradients([self.param_grad], group=self.group)
        if self.lr is not None:
            # data-parallel update: every rank applies the rank-averaged gradient, so replicas stay equal
            allreduce_gradients([points.grad], group=self.group)
            if self._opt is None or self._opt.param_groups[0]["params"][0] is not points:
                self._opt = torch.optim.Adam([points], lr=self.lr, betas=(0.0, 0.999), eps=1e-7)  # cfg.TRAIN.GAUSSIAN
            self._opt.step()
        return loss.detach(), img.detach(), n_buckets


class StandInGenerator(torch.nn.Module):
    """A parameter set with the BG generator's size (69,809,101 fp32 values) in front of the rasterizer:
    `n_layers` chained elementwise layers, each owning an equal share of the parameters, of which the first 28 act on the
    points -- x <- x * (1 + p[:14]) + p[14:28], the identity at initialisation.  Its gradients are DENSE tensors of the
    full size (zeros but for 28 values per layer), produced layer by layer while the backward runs, which is what
    DistributedDataParallel needs to see: message sizes and readiness order of a real generator, none of its arithmetic
    (the generator itself is out of scope: plain torch modules that run unchanged on ROCm)."""

    def __init__(self, n_param=69_809_101, n_layers=4, device=None):
        super().__init__()
        n_layers = max(1, min(int(n_layers), n_param // 28))
        base = n_param // n_layers
        sizes = [base] * (n_layers - 1) + [n_param - base * (n_layers - 1)]
        self.layers = torch.nn.ParameterList(
            [torch.nn.Parameter(torch.zeros(sz, dtype=torch.float32, device=device)) for sz in sizes])

    def forward(self, base_points):
        x = base_points
        for p in self.layers:
            x = x * (1.0 + p[:14]) + p[14:28]
        return x


class StandInDiscriminator(torch.nn.Module):
    """A parameter set with the discriminator's size (18,262,793 fp32 values = 73.1 MB) behind
    the rasterizer: one scalar "label" per image from the image mean and the first two parameters; every other parameter
    gets a dense zero gradient (p.sum() * 0), so DistributedDataParallel all-reduces the real message size.  As with
    StandInGenerator: the message sizes and the place of the module in the step, none of its arithmetic (the
    discriminator is a plain torch module that runs unchanged on ROCm)."""

    def __init__(self, n_param=18_262_793, device=None):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(max(2, int(n_param)), dtype=torch.float32, device=device))

    def forward(self, img):
        return img.mean() * (1.0 + self.p[0]) + self.p[1] + self.p.sum() * 0.0


class DDPTrainStep:
    """The reference's training step (core/train.py:227-295) with its real data-parallel wiring (core/train.py:78-87): the
    generator stand-in is wrapped in torch's DistributedDataParallel(find_unused_parameters=True), so the gradient
    buckets are all-reduced over RCCL/xGMI WHILE the backward is still running (autograd hooks), exactly as the
    reference's DDP does -- not in a separate pass after loss.backward() as TrainStepHarness does.  One frame per
    rank per step through helpers.get_gaussian_rasterization (wrapper -> one autograd node on the [N,14] tensor).
    Buckets of `bucket_cap_mb` (default 64): few large messages, a ring all-reduce over xGMI is per-link bound.

    `discriminator` (optional, round 6): the D-step in front of the G-step as the reference runs it when
    cfg.TRAIN.GAUSSIAN.DISCRIMINATOR.ENABLED (core/train.py:227-257) -- a forward-only render of the same frame under
    no_grad (the rasterizer then writes no backward state), the discriminator on the fake and on the real image, its
    backward with the 73.1 MB all-reduce of ITS gradients, its optimizer step; the G-step then adds the GAN term to its
    loss with the discriminator frozen (:260-283).  Without it the step is the G-step alone (round 1-5 behaviour)."""

    discriminator = dnet = opt_d = last_d_loss = None  # (the G-step-only object of rounds 1-5)

    def __init__(self, rasterizer_wrapper, generator, crop=None, lr=None, group=None, bucket_cap_mb=64, discriminator=None):
        self.rw = rasterizer_wrapper
        self.crop = crop
        self.generator = generator
        self.net = generator
        self.discriminator = discriminator
        self.dnet = discriminator
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            first = next(generator.parameters())
            self.net = torch.nn.parallel.DistributedDataParallel(
                generator, device_ids=[first.device.index] if first.is_cuda else None, process_group=group,
                find_unused_parameters=True, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
            if discriminator is not None:
                dfirst = next(discriminator.parameters())
                self.dnet = torch.nn.parallel.DistributedDataParallel(
                    discriminator, device_ids=[dfirst.device.index] if dfirst.is_cuda else None, process_group=group,
                    find_unused_parameters=True, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
        # cfg.TRAIN.GAUSSIAN optimizer (core/train.py:293-295); None = the step ends with the reduced gradients
        self.opt = (torch.optim.Adam(generator.parameters(), lr=lr, betas=(0.0, 0.999), eps=1e-7)
                    if lr is not None else None)
        self.opt_d = (torch.optim.Adam(discriminator.parameters(), lr=lr, betas=(0.0, 0.999), eps=1e-7)
                      if (lr is not None and discriminator is not None) else None)
        self.last_d_loss = None

    def _render(self, pts, cam_pos, cam_quat):
        from . import helpers
        box = None
        if self.crop is not None:
            x, y, w, h = self.crop
            box = [{"x": x, "y": y, "w": w, "h": h}]
        return helpers.get_gaussian_rasterization(pts[None], self.rw, [cam_pos], [cam_quat], crop_bboxes=box)[0]

    @staticmethod
    def _requires_grad(module, flag):  # utils/helpers.requires_grad
        for p in module.parameters():
            p.requires_grad_(flag)

    def d_step(self, base_points, cam_pos, cam_quat, target=None):
        """core/train.py:227-257.  Returns the discriminator loss (detached)."""
        self._requires_grad(self.generator, False)
        self._requires_grad(self.discriminator, True)
        with torch.no_grad():  # the forward-only render: an inference frame for the rasterizer
            fake = self._render(self.net(base_points), cam_pos, cam_quat).detach()
        real = target if target is not None else torch.zeros_like(fake)
        fake_label, real_label = self.dnet(fake), self.dnet(real)
        # hinge stand-ins for gan_loss(fake, False, dis_update=True) + gan_loss(real, True, dis_update=True)
        loss_d = torch.relu(1.0 + fake_label) + torch.relu(1.0 - real_label)
        for p in self.discriminator.parameters():
            p.grad = None
        loss_d.backward()  # DDP: the discriminator's 73.1 MB of gradients are all-reduced here
        if self.opt_d is not None:
            self.opt_d.step()
        self._requires_grad(self.discriminator, False)
        self._requires_grad(self.generator, True)
        self.last_d_loss = loss_d.detach()
        return self.last_d_loss

    def step(self, base_points, cam_pos, cam_quat, target=None):
        """base_points: [N,14] (no gradient needed).  Returns (loss, image)."""
        if self.discriminator is not None:
            self.d_step(base_points, cam_pos, cam_quat, target)
        for p in self.generator.parameters():
            p.grad = None
        pts = self.net(base_points)
        img = self._render(pts, cam_pos, cam_quat)
        loss = (img - target).abs().mean() if target is not None else img.abs().mean()
        if self.discriminator is not None:  # the GAN term, discriminator frozen (core/train.py:274-283)
            loss = loss + 0.5 * torch.relu(1.0 - self.discriminator(img))
        loss.backward()
        if self.opt is not None:
            self.opt.step()
        return loss.detach(), img.detach()


class InferenceLoop:
    """Shape of the reference's render loop (scripts/inference.py:655-667) around the rasterizer only: for
    every pose, points [N,14] -> image -> uint8 HWC frame on the host.  The reference does this on the
    legacy default stream with a blocking `tensor_to_image(...).cpu()` per frame; here frames alternate
    over `n_streams` side streams (default 3: the measured optimum of the frame loop, bench.py) and land in as
    many pinned host buffers, so frame f's device->host copy and the host-side consumer overlap the next
    frames' renders.  `render_fn(points, cam_pos, cam_quat) -> [3,H,W]`
    is the rasterizer wrapper (or any stand-in on CPU, which degrades to a plain loop)."""

    def __init__(self, render_fn, device=None, n_streams=3, render_uint8_fn=None, static_scene=False):
        """`render_uint8_fn(points, cam_pos, cam_quat) -> uint8 [H,W,3]` (optional): a renderer that produces the video
        frame itself (GaussianRasterizerWrapper(..., as_uint8=True): the blend kernel stores the bytes to_uint8_hwc would
        compute); used instead of render_fn + to_uint8_hwc when given.  `static_scene`: the frames of a run() share one
        point tensor that nobody edits meanwhile AND most of it is off screen in every frame (a whole city's Gaussians
        flown through; not the reference's own loop, which feeds each pose its visible points only) -- the rasterizer
        may keep its cull cache for it (gaussiancity_amd/cull_cache.py: the same frames, the cull streams 16 instead of
        56 bytes per Gaussian; slower than the stateless path when most points are visible).  Frames must be rendered
        under no_grad for it to apply."""
        self.static_scene = bool(static_scene)
        self.render_fn = render_fn
        self.render_uint8_fn = render_uint8_fn
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.n = max(1, int(n_streams))
        self.streams = [torch.cuda.Stream(device=self.device) if self.cuda else None for _ in range(self.n)]
        self._pinned = [None] * self.n

    @staticmethod
    def to_uint8_hwc(img):
        """utils/helpers.tensor_to_image(img, "RGB") * 255 -> uint8 (scripts/inference.py:665): [3,H,W] in
        [-1,1] -> [H,W,3] in [0,255]."""
        return ((img.clamp(-1, 1) / 2 + 0.5) * 255).permute(1, 2, 0).to(torch.uint8)

    def run(self, points, poses, consume=None):
        """Renders every (cam_pos, cam_quat) of `poses`; returns the list of uint8 [H,W,3] numpy frames, or
        calls `consume(index, frame)` per frame (the frame buffer is reused `n_streams` frames later)."""
        if self.static_scene and self.cuda:
            from . import cull_cache
            with cull_cache.scoped(True):
                return self._run(points, poses, consume)
        return self._run(points, poses, consume)

    def _run(self, points, poses, consume):
        out = []
        pending = [None] * self.n  # (index, event) per slot

        def drain(slot):
            if pending[slot] is None:
                return
            idx, ev = pending[slot]
            if ev is not None:
                ev.synchronize()
            frame = self._pinned[slot].numpy()
            if consume is not None:
                consume(idx, frame)
            else:
                out.append((idx, frame.copy()))
            pending[slot] = None

        caller = torch.cuda.current_stream(self.device) if self.cuda else None
        for i, (cam_pos, cam_quat) in enumerate(poses):
            slot = i % self.n
            drain(slot)  # the buffer this frame will land in must have been consumed
            if self.cuda:
                # `points` is normally produced on the caller's stream just before run(): the side
                # stream must not read it earlier, and the allocator must not recycle it while a
                # side stream still uses it
                self.streams[slot].wait_stream(caller)
                if isinstance(points, torch.Tensor) and points.is_cuda:
                    points.record_stream(self.streams[slot])
                with torch.cuda.stream(self.streams[slot]):
                    frame = (self.render_uint8_fn(points, cam_pos, cam_quat) if self.render_uint8_fn is not None else
                             self.to_uint8_hwc(self.render_fn(points, cam_pos, cam_quat)))
                    if self._pinned[slot] is None or self._pinned[slot].shape != frame.shape:
                        self._pinned[slot] = torch.empty(frame.shape, dtype=torch.uint8, pin_memory=True)
                    self._pinned[slot].copy_(frame, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.streams[slot])
                pending[slot] = (i, ev)
            else:
                frame = self.to_uint8_hwc(self.render_fn(points, cam_pos, cam_quat))
                self._pinned[slot] = frame.contiguous()
                pending[slot] = (i, None)
        n = len(poses)
        for k in range(self.n):  # oldest first
            drain((n + k) % self.n)
        if self.cuda:  # whatever the caller does next with `points` is ordered after the last frame
            for st in self.streams:
                caller.wait_stream(st)
        return [f for _, f in sorted(out, key=lambda t: t[0])] if consume is None else None
