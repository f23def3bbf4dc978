"""Fused training iterations for both GANs (the hot loops GAN_DES/SIMNN.py:276-334 and
MMGAN_MIDI_DES/network_tests.py:281-321), restructured for MI355X:

  * discriminator parameters, their gradients and Adam moments live in three flat fp32 buffers (the modules'
    ``.data`` / ``.grad`` become views), so the optimizer is ONE kernel launch and the data-parallel exchange is ONE
    RCCL all-reduce over a flat bucket (gradients + the discriminator loss scalar);
  * D(real) and D(fake) of the discriminator step share weights, so they run as one 2B batch: one forward, one
    backward, gradients arrive already summed (per-sample results are identical, D has no batch statistics);
  * no autograd graph, no host synchronisation: losses stay on the device until the caller reads them;
  * what the reference computes but never uses (``gen_loss.backward()`` only fills D's .grad, which the next
    ``disc_opt.zero_grad()`` wipes; ``gen_opt.step()`` sees no gradients) is executed in faithful mode and skipped
    with ``elide_dead_backward=True`` -- both modes leave identical parameters and losses.

Generators are never updated by the reference loops (no gradient crosses the DES bridge); only their BatchNorm
running statistics move, once (model 1) or twice (model 2) per iteration.
"""
import torch

from . import dp
from . import functional as Fn
from . import ops


class FlatBuffers:
    """Re-home a list of fp32 parameters into one flat buffer with parallel gradient / Adam-moment buffers.

    Physical layout of the gradient bucket (what the data-parallel exchange sees):

        [ rank-local scalars | reduced scalars | gradients of the small parameters | gradient of the largest parameter ]

    so that the exchange can be split in two contiguous collectives: the big tail (model 1: fc1.weight, 99.9 % of the
    bytes) as soon as its gradient exists -- overlapped with the convolution backward -- and the small head (the
    reduced scalars = the discriminator loss + the small gradients) at the end.  The first ``local`` of the ``extra``
    scalar slots are NEVER part of a collective (gen_loss lives there: under the pipelined schedule it is written
    before the exchange of the following iteration and must not be summed over ranks).  ``views`` / ``grad_views`` keep
    the caller's parameter order.
    """

    ALIGN = 64        # floats (256 bytes): start of the largest parameter in every flat buffer

    def __init__(self, params, extra=0, local=0):
        params = [p for p in params]
        assert params, "no parameters"
        dev = params[0].device     # the buffers can be laid out anywhere; the fused step itself needs a HIP device
        self.params = params
        self.numel = sum(p.numel() for p in params)                 # true parameter count
        big = max(range(len(params)), key=lambda i: params[i].numel())
        order = [i for i in range(len(params)) if i != big] + [big]
        n_small = self.numel - params[big].numel()
        # The largest parameter starts on a 256-byte boundary in all four buffers (GEMMs write its gradient in place and
        # take their 16-byte-vector path only for aligned operands -- a 4-byte-aligned slot sent fc1's weight-gradient
        # GEMM down the generic kernel, 200 us instead of 25).  The gap holds zeros: Adam leaves them zero.
        assert extra % 4 == 0 and 0 <= local <= extra
        self.n_small = (n_small + self.ALIGN - 1) // self.ALIGN * self.ALIGN        # incl. the zero padding
        self.n_pad = self.n_small - n_small
        self.n_extra = extra
        self.n_local = local
        total = self.n_small + params[big].numel()
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        # bucket: [extra | small (+ pad) | big]; the extra slots are padded so that the gradients start aligned too
        self._lead = (extra + self.ALIGN - 1) // self.ALIGN * self.ALIGN if extra else 0
        self._bucket_store = torch.zeros(self._lead + total, dtype=torch.float32, device=dev)
        self.bucket = self._bucket_store[self._lead - extra:]
        self.extra = self.bucket[:extra]
        self.grad = self.bucket[extra:]
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, self.grad_views = [None] * len(params), [None] * len(params)
        off = 0
        for i in order:
            p = params[i]
            assert p.dtype == torch.float32 and p.device == dev
            n = p.numel()
            if i == big:
                off = self.n_small
            v = self.flat[off:off + n].view(p.shape)
            v.copy_(p.data)
            p.data = v
            g = self.grad[off:off + n].view(p.shape)
            p.grad = g
            self.views[i] = v
            self.grad_views[i] = g
            off += n
        self.step_count = 0
        self._hyper = None
        self.derived_record = None   # device record holding values derived from _hyper (gdm_simnn_adam_step's)
        self._hyper_host = None

    # the two contiguous pieces of the data-parallel exchange
    def bucket_big(self):
        return self.bucket[self.n_extra + self.n_small:]

    def bucket_head(self):
        """[reduced scalars | small gradients] -- starts after the rank-local scalars."""
        return self.bucket[self.n_local: self.n_extra + self.n_small]

    def bucket_reduced(self):
        """Everything that crosses ranks, as one contiguous piece: [reduced scalars | small | big gradients]."""
        return self.bucket[self.n_local:]

    def sync_hyper(self, lr, betas, eps, grad_scale=1.0):
        """Bring the device hyper-parameter record up to date (lr schedule, ...) WITHOUT touching the step counter;
        a captured graph reads the record, so ``replay`` calls this before every launch."""
        want = (float(lr), float(betas[0]), float(betas[1]), float(eps), float(grad_scale))
        if self._hyper is not None and want != self._hyper_host:
            if torch.cuda.is_current_stream_capturing():
                raise ops.GdmError("optimizer hyper-parameters changed inside a captured step: re-capture the graph")
            self._hyper[1:6].copy_(torch.tensor(want, dtype=torch.float32), non_blocking=False)
            self._hyper_host = want
            if self.derived_record is not None:
                self.derived_record.zero_()          # terms cached on the device for the old lr / betas
        return want

    def adam(self, lr, betas, eps, grad_scale=1.0, big_pc=None):
        """One fused Adam step over the flat buffer.  Step counter and hyper-parameters live in an 8-float device
        record (so a captured hipGraph replays correctly); the host only rewrites it when lr/scale change.

        big_pc = (n, c, pix, shadow): the largest parameter is an (n, c, pix) tensor whose gradient slot holds the
        (n, pix, c) "channels-last" layout its weight-gradient GEMM produces, and ``shadow`` (n, pix, c) receives the
        updated value as the GEMM operand copy (model 1's fc1.weight): the small parameters take the plain kernel, the
        big one the transposing kernel -- no separate permute passes."""
        want = self.sync_hyper(lr, betas, eps, grad_scale)
        if self._hyper is None:
            self._hyper = ops.adam_hyper(self.flat.device, *want, step=self.step_count)
            self._hyper_host = want
        self.step_count += 1
        if big_pc is None:
            ops.adam_step_dev(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self._hyper)
            return
        ns = self.n_small
        n, c, pix, shadow = big_pc
        assert n * c * pix == self.flat.numel() - ns and ns > 0
        ops.adam_step_dev(self.flat[:ns], self.grad[:ns], self.exp_avg[:ns], self.exp_avg_sq[:ns], self._hyper)
        ops.adam_step_dev_pc(self.flat[ns:], self.grad[ns:], self.exp_avg[ns:], self.exp_avg_sq[ns:], n, c, pix, shadow,
                             self._hyper, advance_step=False)


class _TrainerBase:
    def _init_common(self, d_params, lr, betas, eps, compute_dtype, elide_dead_backward, process_group):
        self.dt = Fn.get_compute_dtype() if compute_dtype is None else Fn._NAMES[compute_dtype]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.elide = elide_dead_backward
        self.pg = process_group
        self.world = dp.world_size(process_group)
        # extra[0] = gen_loss (rank-local, never reduced: the pipelined schedule writes it before the NEXT iteration's
        # exchange), [1:4] rank-local scratch; extra[4] = disc_loss (rides the all-reduce), [5:8] reduced padding so
        # that the reduced range and the gradients start on 16-byte boundaries
        self.d = FlatBuffers(d_params, extra=8, local=4)
        self.loss_g = self.d.extra[0:1]
        self.loss_d = self.d.extra[4:5]
        self.iterations = 0

    def _reduce_big_async(self):
        """Start the SUM all-reduce of the largest gradient (call on the stream that produced it); no-op on 1 rank."""
        self._pending = dp.allreduce_async_(self.d.bucket_big(), self.pg)

    def _reduce(self):
        """The data-parallel exchange of everything that is not yet in flight: SUM over ranks (1/world is folded into
        Adam and into the loss read-out).  No-op on one rank."""
        pending = getattr(self, "_pending", None)
        if pending is None:
            red = self.d.bucket_reduced()
            dp.allreduce_bucket_(red, red.numel(), self.pg)                         # everything in one collective
        else:
            # head of the bucket = [disc_loss | small gradients]; the big tail has been in flight since its GEMM finished
            dp.allreduce_bucket_(self.d.bucket_head(), self.d.bucket_head().numel(), self.pg)
            pending.wait()                                                          # current stream waits for the tail
            self._pending = None

    def _adam(self):
        self.d.adam(self.lr, self.betas, self.eps, grad_scale=1.0 / self.world)

    def _reduce_and_step(self):
        self._reduce()
        self._adam()

    def _sync_hyper(self):
        self.d.sync_hyper(self.lr, self.betas, self.eps, 1.0 / self.world)

    @torch.no_grad()
    def load_discriminator_state(self, params=None, exp_avg=None, exp_avg_sq=None, step=None):
        """Overwrite the discriminator's parameters and/or Adam state in place (lists in the trainer's parameter
        order; resume from a checkpoint that kept optimizer state, teacher-forced parity tests).  Storage does not
        move, so a captured graph stays valid; the packed / permuted operands derived from the weights are rebuilt."""
        d = self.d
        for src, flat in ((params, d.flat), (exp_avg, d.exp_avg), (exp_avg_sq, d.exp_avg_sq)):
            if src is None:
                continue
            assert len(src) == len(d.views)
            for v, s_ in zip(d.views, src):
                off = (v.data_ptr() - d.flat.data_ptr()) // 4
                flat[off:off + v.numel()].copy_(s_.detach().reshape(-1).to(flat.device, torch.float32))
        if step is not None:
            d.step_count = int(step)
            if d._hyper is not None:
                d._hyper.view(torch.int32)[0:1].copy_(torch.tensor([int(step)], dtype=torch.int32))
        if params is not None:
            self._refresh_operands()

    # ---- non-finite values -------------------------------------------------------------------------------------------
    # The reference runs under torch.autograd.set_detect_anomaly(True) (network_tests.py:211): a NaN raises there.  The
    # kernels here are compiled without NaN semantics (a NaN / Inf in an input or weight has NO defined effect on the
    # results -- it may propagate or be flushed by an fmaxf), so the defined behaviour is: (a) with anomaly mode on,
    # every eager ``step`` counts the non-finite elements of its inputs, losses and discriminator parameters on the
    # device and raises ops.NonFiniteError before returning; (b) ``check_finite(*tensors)`` does the same on request
    # (graph replays, anomaly mode off).
    def _watch(self, *tensors):
        if torch.is_anomaly_enabled():
            if getattr(self, "_nf", None) is None:
                self._nf = torch.zeros(1, dtype=torch.int32, device=self.d.flat.device)
            ops.nonfinite_count([t for t in tensors if isinstance(t, torch.Tensor)], self._nf)

    def check_finite(self, *tensors):
        """Raise ops.NonFiniteError if the last losses, the discriminator's parameters / Adam moments, anything watched
        since the last check (anomaly mode) or any of ``tensors`` (e.g. the batch just used) holds a NaN or Inf.
        Synchronises."""
        if getattr(self, "_nf", None) is None:
            self._nf = torch.zeros(1, dtype=torch.int32, device=self.d.flat.device)
        ops.nonfinite_count([self.loss_g, self.loss_d, self.d.flat, self.d.exp_avg, self.d.exp_avg_sq, *tensors], self._nf)
        n = int(self._nf.item())
        self._nf.zero_()
        if n:
            raise ops.NonFiniteError(f"{type(self).__name__}: {n} non-finite value(s) in inputs, losses, discriminator "
                                     f"parameters or optimizer state after iteration {self.iterations}")

    def _anomaly_check(self):
        if torch.is_anomaly_enabled() and not torch.cuda.is_current_stream_capturing():
            self.check_finite()

    def disc_loss_value(self):
        """Global-batch mean of the last discriminator loss (the SUM over ranks rode the gradient all-reduce)."""
        return self.loss_d.item() / self.world

    def gen_loss_value(self):
        """THIS RANK's batch mean of the last generator loss (it is computed after the exchange and stays local)."""
        return self.loss_g.item()

    def gen_loss_global(self):
        """Global-batch mean of the last generator loss.  A collective: every rank has to call it."""
        if self.world == 1:
            return self.loss_g.item()
        t = self.loss_g.clone()
        dp.allreduce_bucket_(t, 1, self.pg)
        return t.item() / self.world


class SimnnTrainer(_TrainerBase):
    """One object per (Generator, Discriminator) pair of model 1; ``step`` = one iteration of SIMNN.py:276-334.

    overlap=True runs the independent branches of the iteration on side HIP streams (generator forward beside the
    discriminator step; fc1's weight gradient and conv2's weight gradient beside the data-gradient chain); all
    branches re-join before the optimizer and before ``step`` returns.  ``capture``/``replay`` record the whole
    iteration (all branches) into one hipGraph for fixed input buffers.

    ``step_pipelined`` is the same iteration scheduled across two calls: the "generator" half of iteration i (the
    discriminator pass on fake_i with the weights Adam(i) produced, SIMNN.py:322-331) only reads state that nothing
    modifies until Adam(i+1), so it runs on a stream of its own BESIDE the discriminator step of iteration i+1 (its
    memory-bound GEMMs fill the issue-bound conv kernels of the other chain and vice versa).  Every iteration still
    executes exactly the reference's work in the reference's dependency order; ``flush`` runs the half that is
    still pending.  After ``flush`` all state equals the sequential ``step``'s bit for bit.
    """

    def __init__(self, gen, disc, lr=0.00002, betas=(0.5, 0.999), eps=1e-8, compute_dtype=None,
                 elide_dead_backward=False, process_group=None, overlap=True, one_launch_optimizer=True):
        self.gen, self.disc = gen, disc
        self.one_launch_optimizer = one_launch_optimizer     # False: adam_prep, Adam(small), Adam(fc1.weight), re-pack
        self._init_common([disc.conv1.weight, disc.conv1.bias, disc.conv2.weight, disc.conv2.bias, disc.fc1.weight,
                           disc.fc1.bias, disc.fc2.weight, disc.fc2.bias], lr, betas, eps, compute_dtype,
                          elide_dead_backward, process_group)
        self._last_generated = None
        self._graph_gen = None     # pipelined capture: the generator forward as a graph of its own (see capture)
        self._gen_event = None     # recorded behind the last replay of that graph
        self._prepared = None      # (packed conv2 images, permuted fc1 weight) for the current weights
        self.overlap = overlap
        self._tm_cache = {}        # tap-major copies of the generator's ConvTranspose2d weights (per weight version)
        self._side = None
        self._graph = None
        self._scratch_grads = None
        self._pending_fake = None  # step_pipelined: fake batch whose generator half has not run yet (= _fake_buf)
        self._fake_buf = None      # trainer-owned copy of that batch

    @property
    def last_generated(self):
        """The generator's output of the last iteration; after a pipelined ``replay`` it is produced on a stream of the
        trainer's own, and reading it makes the current stream wait for that stream's last replay."""
        if self._gen_event is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_event(self._gen_event)
        return self._last_generated

    def invalidate_weights(self):
        """Call after changing discriminator weights from outside (e.g. load_state_dict)."""
        self._prepared = None

    def _refresh_operands(self):
        if self._prepared is not None:
            Fn.simnn_disc_prepare(self.d.views[2], self.d.views[4], self.dt, out=self._prepared)

    def _adam(self):
        """Adam, with fc1.weight's two layout changes folded in: its gradient slot holds the channels-last (128, P, 32)
        layout the weight-gradient GEMM writes (``disc.fc1.weight.grad`` is therefore NOT in the parameter's order while
        a trainer owns the module), and the kernel writes the updated weight's operand copy ``self._prepared[1]`` itself;
        conv2's packed images are rebuilt right after."""
        wf1 = self.d.views[4]
        n, k = wf1.shape
        d = self.d
        if not self.one_launch_optimizer:
            d.adam(self.lr, self.betas, self.eps, grad_scale=1.0 / self.world, big_pc=(n, 32, k // 32, self._prepared[1]))
            ops.simnn_conv2_pack(d.views[2], self.dt, out=self._prepared[0])
            return
        # one launch: adam_prep + Adam(small) + Adam(fc1.weight) + conv2 re-pack (gdm_simnn_adam_step)
        if getattr(self, "_adam_done", None) is None:
            # completion counters + cached bias-correction terms; sync_hyper zeroes it when it rewrites the record
            self._adam_done = d.derived_record = torch.zeros(ops.SIMNN_ADAM_RECORD_INTS, dtype=torch.int32,
                                                             device=d.flat.device)
        want = d.sync_hyper(self.lr, self.betas, self.eps, 1.0 / self.world)
        if d._hyper is None:
            d._hyper = ops.adam_hyper(d.flat.device, *want, step=d.step_count)
            d._hyper_host = want
        d.step_count += 1
        ns = d.n_small
        ops.simnn_adam_step(d.flat[ns:], d.grad[ns:], d.exp_avg[ns:], d.exp_avg_sq[ns:], n, 32, k // 32, self._prepared[1],
                            d.flat[:ns], d.grad[:ns], d.exp_avg[:ns], d.exp_avg_sq[:ns], d.views[2], self._prepared[0],
                            d._hyper, self._adam_done)

    def _gen_state(self):
        g = self.gen
        ws = [g.conv1.weight.detach(), g.conv2.weight.detach(), g.conv3.weight.detach(), g.conv4.weight.detach()]
        bns = [(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.num_batches_tracked)
               for bn in (g.batch_norm1, g.batch_norm2, g.batch_norm3)]
        return ws, bns

    def _streams(self, dev):
        if self._side is None:
            self._side = [torch.cuda.Stream(dev) for _ in range(3)]
        return self._side

    def _d_backward(self, saved, dh, pack, wf1p, outs, x_pair, keep, fork=True):
        """Discriminator backward below the head: fc1's weight gradient runs beside the data-gradient chain
        (fork=False: everything on the current stream -- a branch of a captured graph must not fork again: a
        second-level fork crashed hipStreamEndCapture)."""
        dt = self.dt
        x, p1, code1, flat, code2 = saved[:5]
        b = p1.shape[0]
        n, k = wf1p.shape
        main = torch.cuda.current_stream()
        side = self._streams(p1.device) if (self.overlap and fork) else None
        # branch A: fc1 weight gradient
        if side:
            side[1].wait_stream(main)
        with torch.cuda.stream(side[1] if side else main):
            # (128, P*32): the channels-last layout of the feature map, kept as it is -- Adam transposes on the fly
            ops.gemm(dh.t(), flat, compute=dt, out=outs[4].view(n, k))
            if self.world > 1 and outs is self.d.grad_views:
                # 99.9 % of the exchange (fc1.weight's gradient) starts now and overlaps the convolution backward
                self._reduce_big_async()
        # main: fc1 data gradient = pooled gradient of the conv trunk
        dflat = ops.gemm(dh, wf1p, compute=dt, out_dtype=dt)
        h1s, w1s = p1.shape[1], p1.shape[2]
        dp2 = dflat.view(b, h1s // 2, w1s // 2, 32)
        keep.append(dflat)
        # conv2 weight gradient, then conv2 data gradient with conv1's weight gradient fused in: both on the main
        # stream -- they are issue-bound persistent kernels that fill the chip, side by side they only take turns
        # (tools/overlap_probe.py), and one after the other each runs at its own speed
        ops.simnn_conv2_bwd_weight(dp2, code2, p1, out=(outs[2], outs[3]))
        x0, x1 = x_pair if x_pair is not None else (x, None)
        ops.simnn_conv2_bwd_fused(dp2, code2, pack, code1, x0, x1, out=(outs[0], outs[1]))
        if side:
            main.wait_stream(side[1])

    @torch.no_grad()
    def step(self, real, noise, fake):
        """real (B,H,W) fp32 on the device; noise (B,noise_dim,1,1); fake: (B,H,W) tensor, or a callable
        ``fake(generated (B,1,20,20) device tensor) -> (B,H,W) tensor`` standing in for matrix_to_wav (SIMNN.py:301).
        Returns (disc_loss, gen_loss) as 1-element device tensors: gen_loss is this rank's batch mean; disc_loss is this
        rank's batch mean on one rank and the SUM over ranks of the batch means with world > 1 (it rides the gradient
        all-reduce; ``disc_loss_value()`` divides by world)."""
        if self._pending_fake is not None:
            self.flush()
        dt = self.dt
        w1, b1, w2, b2, wf1, bf1, wf2, bf2 = self.d.views
        gv = self.d.grad_views
        real = Fn._f32c(real)
        self._watch(real, noise, fake)
        b, h, w = real.shape
        main = torch.cuda.current_stream()
        side = self._streams(real.device) if self.overlap else None
        keep = []
        ws, bns = self._gen_state()

        def generator_forward():
            # SIMNN.py:293-296; the output only feeds the (external) bridge -> own stream
            if side:
                side[0].wait_stream(main)
            with torch.cuda.stream(side[0] if side else main):
                generated, gsaved = Fn.simnn_gen_forward(noise, ws, bns, self.gen.training, dt, cache=self._tm_cache,
                                                           need_backward=False)
                keep.append(gsaved)
            self._last_generated = generated
            return generated

        bridge = callable(fake)
        if bridge:
            generated = generator_forward()
            if side:
                main.wait_stream(side[0])
            fake = fake(generated)
        fake = Fn._f32c(fake.to(real.device))
        assert fake.shape == real.shape, (fake.shape, real.shape)
        # --- discriminator step on the 2B batch [real ; fake] (SIMNN.py:282-316)
        if self._prepared is None:
            self._prepared = Fn.simnn_disc_prepare(w2, wf1, dt)
        pack, wf1p = self._prepared
        h1, w1s = (h + 1) // 2, (w + 1) // 2
        adt = ops.torch_dtype(dt)
        p1 = torch.empty((2 * b, h1, w1s, 16), dtype=adt, device=real.device)
        code1 = torch.empty((2 * b, h1, ops.simnn_code1_width(w1s)), dtype=torch.int64, device=real.device)
        # two B launches, not one 2B launch (ops.simnn_conv1_fwd(..., x1=fake)): inside the iteration the 2B launch takes
        # 64-66 us against 2 x 30 (same-box A/B of the step: 0.612-0.614 vs 0.608-0.610 ms)
        ops.simnn_conv1_fwd(real, w1, b1, dt, out=(p1[:b], code1[:b]))
        ops.simnn_conv1_fwd(fake, w1, b1, dt, out=(p1[b:], code1[b:]))
        if not bridge:
            # tensor stand-in for the bridge: the generator is independent of the discriminator step and runs beside
            # it.  Forked after the first main-stream launch: a branch that forks at the very root of a captured graph
            # was observed to run BEFORE the main branch instead of beside it.
            generator_forward()
        hid, saved = Fn.simnn_disc_features(None, w1, b1, pack, b2, wf1p, bf1, dt, trunk_out=(p1, code1))
        # fc2 + sigmoid + both BCE terms (labels 0.9 / 0.1, SIMNN.py:284-311) + head backward: one launch pair
        _prob, dh, _ = ops.simnn_head(hid, wf2, bf2, b, 0.9, 0.1, loss_out=self.loss_d, grad_out=(gv[6], gv[7], gv[5]),
                                      dh_dtype=dt)
        self._d_backward(saved, dh, pack, wf1p, gv, (real, fake), keep)
        # Adam also rewrites the weight-derived operands in place (fc1's operand copy inside the kernel, conv2's packed
        # images right after): they are cross-iteration state, so their storage must be stable under graph replay
        self._reduce_and_step()
        # --- "generator" step (SIMNN.py:322-331): D forward on fake with the updated weights, label 1.0
        p1g = torch.empty((b, h1, w1s, 16), dtype=adt, device=real.device)
        code1g = torch.empty((b, h1, ops.simnn_code1_width(w1s)), dtype=torch.int64, device=real.device)
        ops.simnn_conv1_fwd(fake, w1, b1, dt, out=(p1g, code1g))
        hid_g, saved_g = Fn.simnn_disc_features(None, w1, b1, pack, b2, wf1p, bf1, dt, trunk_out=(p1g, code1g))
        _prob, dh_g, _ = ops.simnn_head(hid_g, wf2, bf2, b, 1.0, 1.0, loss_out=self.loss_g, want_grad=not self.elide,
                                        dh_dtype=dt)
        if not self.elide:
            # dead values: gen_loss.backward() only fills D's .grad, which the next zero_grad() wipes (SIMNN.py:330,
            # 282); they are computed (faithful mode) into a scratch set of gradient buffers
            if self._scratch_grads is None:
                self._scratch_grads = [torch.empty_like(g) for g in gv]
            self._d_backward(saved_g, dh_g, pack, wf1p, self._scratch_grads, (fake, None), keep)
        if side:
            main.wait_stream(side[0])
        # gen_opt.step(): every generator .grad is None -> no-op
        self.iterations += 1
        del keep
        self._anomaly_check()
        return self.loss_d, self.loss_g

    # ---- the same iteration, generator half of iteration i beside the discriminator step of iteration i+1 ----------
    def _generator_half(self, fake, keep):
        """SIMNN.py:322-331 on the current stream: D forward on fake with the current weights, label 1.0, and (faithful
        mode) the dead backward into scratch gradient buffers.  Needs fresh ``self._prepared``."""
        dt = self.dt
        w1, b1, w2, b2, wf1, bf1, wf2, bf2 = self.d.views
        pack, wf1p = self._prepared
        b = fake.shape[0]
        hid_g, saved_g = Fn.simnn_disc_features(fake, w1, b1, pack, b2, wf1p, bf1, dt)
        _prob, dh_g, _ = ops.simnn_head(hid_g, wf2, bf2, b, 1.0, 1.0, loss_out=self.loss_g, want_grad=not self.elide,
                                        dh_dtype=dt)
        if not self.elide:
            if self._scratch_grads is None:
                self._scratch_grads = [torch.empty_like(g) for g in self.d.grad_views]
            self._d_backward(saved_g, dh_g, pack, wf1p, self._scratch_grads, (fake, None), keep, fork=False)
        keep.append((hid_g, saved_g, dh_g))

    @torch.no_grad()
    def step_pipelined(self, real, noise, fake, with_generator=True):
        """Like ``step`` for tensor inputs, but the generator half of THIS iteration is left pending and the pending
        half of the previous iteration runs beside this iteration's discriminator step.  Returns (disc_loss of this
        iteration, gen_loss of the previous one) as device tensors; call ``flush`` after the last iteration."""
        if callable(fake):
            raise ops.GdmError("step_pipelined needs a tensor for the fake batch (the bridge is host code)")
        dt = self.dt
        w1, b1, w2, b2, wf1, bf1, wf2, bf2 = self.d.views
        gv = self.d.grad_views
        real = Fn._f32c(real)
        fake = Fn._f32c(fake.to(real.device))
        assert fake.shape == real.shape, (fake.shape, real.shape)
        b, h, w = real.shape
        main = torch.cuda.current_stream()
        side = self._streams(real.device) if self.overlap else None
        keep = []
        ws, bns = self._gen_state()
        pending = self._pending_fake
        if self._prepared is None:
            self._prepared = Fn.simnn_disc_prepare(w2, wf1, dt)
        pack, wf1p = self._prepared
        if self._scratch_grads is None:
            self._scratch_grads = [torch.empty_like(g) for g in gv]
        h1, w1s = (h + 1) // 2, (w + 1) // 2
        adt = ops.torch_dtype(dt)
        p1 = torch.empty((2 * b, h1, w1s, 16), dtype=adt, device=real.device)
        code1 = torch.empty((2 * b, h1, ops.simnn_code1_width(w1s)), dtype=torch.int64, device=real.device)
        # two B launches, not one 2B launch (ops.simnn_conv1_fwd(..., x1=fake)): inside the iteration the 2B launch takes
        # 64-66 us against 2 x 30 (same-box A/B of the step: 0.612-0.614 vs 0.608-0.610 ms)
        ops.simnn_conv1_fwd(real, w1, b1, dt, out=(p1[:b], code1[:b]))
        ops.simnn_conv1_fwd(fake, w1, b1, dt, out=(p1[b:], code1[b:]))
        # branches fork after the first main-stream launch (see step)
        if with_generator:
            if side:
                side[0].wait_stream(main)
            with torch.cuda.stream(side[0] if side else main):
                generated, gsaved = Fn.simnn_gen_forward(noise, ws, bns, self.gen.training, dt, cache=self._tm_cache,
                                                               need_backward=False)
                keep.append(gsaved)
            self._last_generated = generated
        # generator half of the previous iteration: reads the weights / prepared operands that stay untouched until
        # this call's Adam, writes only gen_loss and scratch buffers.  Then, on the same stream (so after the half's last
        # read of it), the trainer's own copy of the fake batch is refreshed with THIS iteration's: the caller may
        # refill ``fake`` as soon as this call returns (loaders / bridges that reuse their output buffer do).
        if side:
            side[2].wait_stream(main)
        with torch.cuda.stream(side[2] if side else main):
            if pending is not None:
                self._generator_half(pending, keep)
            if self._fake_buf is None or self._fake_buf.shape != fake.shape:
                if torch.cuda.is_current_stream_capturing():
                    raise ops.GdmError("step_pipelined saw a new batch geometry inside a graph capture")
                self._fake_buf = torch.empty_like(fake)
            self._fake_buf.copy_(fake)
        hid, saved = Fn.simnn_disc_features(None, w1, b1, pack, b2, wf1p, bf1, dt, trunk_out=(p1, code1))
        _prob, dh, _ = ops.simnn_head(hid, wf2, bf2, b, 0.9, 0.1, loss_out=self.loss_d, grad_out=(gv[6], gv[7], gv[5]),
                                      dh_dtype=dt)
        self._d_backward(saved, dh, pack, wf1p, gv, (real, fake), keep)
        if side:
            main.wait_stream(side[2])
        self._reduce_and_step()         # (Adam refreshes the weight-derived operands in place)
        if side and with_generator:
            main.wait_stream(side[0])
        self._pending_fake = self._fake_buf
        self.iterations += 1
        del keep
        return self.loss_d, self.loss_g

    @torch.no_grad()
    def flush(self):
        """Run the generator half left pending by ``step_pipelined``; returns its gen_loss (device tensor)."""
        if self._pending_fake is None:
            return self.loss_g
        keep = []
        self._generator_half(self._pending_fake, keep)
        self._pending_fake = None
        del keep
        return self.loss_g

    # ---- hipGraph capture of the whole iteration for fixed input buffers -------------------------------------------
    def capture(self, real, noise, fake, pipelined=False, generator_graph=True, pieces=False):
        """Record one iteration on (real, noise, fake) -- tensors whose storage is re-used for every replay -- into a
        hipGraph.  Not available with a callable bridge.  With more than one rank (or ``pieces=True``) the pipelined
        iteration is recorded as FIVE fork-free graphs replayed around the two eager collectives (``_capture_pieces``).

        pipelined + generator_graph (the default): the generator forward (6 launches that feed nothing inside the
        iteration) is a graph of its OWN, replayed on a stream of the trainer's own beside the main graph: as a branch
        of the main graph its fork and join were cross-queue dependencies inside the iteration (0.699 -> 0.684 ms;
        generator_graph=False keeps it inside the main graph).  The two graphs run concurrently, so each is captured
        under a scratch-buffer namespace of its own (ops.workspace_namespace)."""
        if callable(fake):
            raise ops.GdmError("graph capture needs tensor inputs")
        if self.world > 1 or pieces:
            if not pipelined:
                raise ops.GdmError("with more than one rank only the pipelined schedule is captured (pipelined=True)")
            return self._capture_pieces(real, noise, fake)
        self._pieces = None
        self._static = (Fn._f32c(real), noise, Fn._f32c(fake))
        fn = self.step_pipelined if pipelined else self.step
        warm = torch.cuda.Stream(real.device)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for _ in range(2):
                fn(*self._static)
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        self._graph_gen = None
        self._graph = torch.cuda.CUDAGraph()
        if pipelined and generator_graph:
            ws, bns = self._gen_state()
            self._graph_gen = torch.cuda.CUDAGraph()
            with ops.workspace_namespace(("graph", id(self._graph_gen))), torch.cuda.graph(self._graph_gen):
                self._last_generated, _gs = Fn.simnn_gen_forward(self._static[1], ws, bns, self.gen.training, self.dt,
                                                                 cache=self._tm_cache, need_backward=False)
            self._gen_replay_stream = torch.cuda.Stream(real.device)
            with ops.workspace_namespace(("graph", id(self._graph))), torch.cuda.graph(self._graph):
                self.step_pipelined(*self._static, with_generator=False)   # finds a pending half and leaves one
        else:
            with ops.workspace_namespace(("graph", id(self._graph))), torch.cuda.graph(self._graph):
                fn(*self._static)
        self.d.step_count -= 1     # the captured call did not execute: the device-side step counter did not move
        self.iterations -= 1
        return self._graph

    # ---- the pipelined iteration as graphs around the data-parallel exchange (world > 1) ------------------------------
    def _capture_pieces(self, real, noise, fake):
        """What N > 1 ranks replay per iteration (a collective is not part of a graph here):

            stream sg : [G generator forward]                                                     (feeds nothing inside)
            stream sh : [H generator half of the previous iteration, then the copy of this fake batch]
            main      : [A forward of the 2B batch + head + fc1 dW] [B rest of the backward] all-reduce(head) [C Adam]
            stream s1 :                                              all-reduce(fc1.weight.grad, async) .....^
            joins     : C waits for H (it rewrites the weights H reads) and for the big all-reduce

        5 graph launches + 2 collectives instead of ~60 kernel launches.  Each graph has its own scratch namespace and
        memory pool (H and G run beside everything); tensors that cross a graph boundary stay referenced in
        ``self._pieces`` so their storage is never handed out again.  Bit-identical to eager ``step_pipelined`` calls."""
        real, fake = Fn._f32c(real), Fn._f32c(fake)
        self._static = (real, noise, fake)
        dev = real.device
        warm = torch.cuda.Stream(dev)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for _ in range(2):
                self.step_pipelined(*self._static)
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        dt = self.dt
        w1, b1, w2, b2, wf1, bf1, wf2, bf2 = self.d.views
        gv = self.d.grad_views
        pack, wf1p = self._prepared
        b, h, w = real.shape
        h1, w1s = (h + 1) // 2, (w + 1) // 2
        adt = ops.torch_dtype(dt)
        g = {k: torch.cuda.CUDAGraph() for k in ("gen", "half", "a", "b", "c")}
        ctx = {}

        def cap(name):
            return ops.workspace_namespace(("graph", id(g[name])))

        with cap("gen"), torch.cuda.graph(g["gen"]):
            ws, bns = self._gen_state()
            self._last_generated, _gs = Fn.simnn_gen_forward(noise, ws, bns, self.gen.training, dt, cache=self._tm_cache,
                                                             need_backward=False)
        with cap("half"), torch.cuda.graph(g["half"]):
            keep = []
            self._generator_half(self._pending_fake, keep)
            self._fake_buf.copy_(fake)
            ctx["half_keep"] = keep
        with cap("a"), torch.cuda.graph(g["a"]):
            p1 = torch.empty((2 * b, h1, w1s, 16), dtype=adt, device=dev)
            code1 = torch.empty((2 * b, h1, ops.simnn_code1_width(w1s)), dtype=torch.int64, device=dev)
            ops.simnn_conv1_fwd(real, w1, b1, dt, out=(p1[:b], code1[:b]))
            ops.simnn_conv1_fwd(fake, w1, b1, dt, out=(p1[b:], code1[b:]))
            hid, saved = Fn.simnn_disc_features(None, w1, b1, pack, b2, wf1p, bf1, dt, trunk_out=(p1, code1))
            _prob, dh, _ = ops.simnn_head(hid, wf2, bf2, b, 0.9, 0.1, loss_out=self.loss_d, grad_out=(gv[6], gv[7], gv[5]),
                                          dh_dtype=dt)
            ctx.update(saved=saved, dh=dh, hid=hid, prob=_prob)
            _x, p1, code1, flat, code2 = saved[:5]
            n, k = wf1p.shape
            # fc1's weight gradient closes this graph: its all-reduce (99.9 % of the exchange) then runs beside graph B.
            # (As a graph of its own on a third stream -- the eager schedule's branch -- every iteration paid two more
            # cross-queue dependencies: 0.785 ms against 0.666 eager on one rank.)
            ops.gemm(dh.t(), flat, compute=dt, out=gv[4].view(n, k))
        with cap("b"), torch.cuda.graph(g["b"]):
            dflat = ops.gemm(ctx["dh"], wf1p, compute=dt, out_dtype=dt)
            dp2 = dflat.view(2 * b, h1 // 2, w1s // 2, 32)
            ops.simnn_conv2_bwd_weight(dp2, code2, p1, out=(gv[2], gv[3]))
            ops.simnn_conv2_bwd_fused(dp2, code2, pack, code1, real, fake, out=(gv[0], gv[1]))
            ctx["dflat"] = dflat
        with cap("c"), torch.cuda.graph(g["c"]):
            self._adam()
        self.d.step_count -= 1     # the captured Adam did not execute
        self._pieces = (g, ctx)
        self._graph, self._graph_gen = g["a"], None
        self._piece_streams = tuple(torch.cuda.Stream(dev) for _ in range(3))      # sg, sh, s1
        return g

    def _replay_pieces(self):
        g, _ctx = self._pieces
        sg, sh, s1 = self._piece_streams
        main = torch.cuda.current_stream()
        # G and H start behind everything the caller has enqueued (refills of the static inputs, the previous Adam)
        sg.wait_stream(main)
        with torch.cuda.stream(sg):
            g["gen"].replay()
            self._gen_event = sg.record_event()
        sh.wait_stream(main)
        with torch.cuda.stream(sh):
            g["half"].replay()
            half_ev = sh.record_event()
        g["a"].replay()
        if self.world > 1:
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                self._reduce_big_async()      # 99.9 % of the exchange starts now, beside the convolution backward
        g["b"].replay()
        self._reduce()                        # head all-reduce on this stream + wait for the big one (no-op on 1 rank)
        main.wait_event(half_ev)              # Adam rewrites the weights the generator half reads
        g["c"].replay()
        main.wait_event(self._gen_event)      # a refill of ``noise`` comes after the generator's reads
        self.d.step_count += 1
        self.iterations += 1
        return self.loss_d, self.loss_g

    def replay(self):
        self._sync_hyper()         # lr schedule etc.: the captured Adam reads the device record
        if getattr(self, "_pieces", None) is not None:
            return self._replay_pieces()
        main = torch.cuda.current_stream()
        if self._graph_gen is not None:
            # behind everything the caller has enqueued so far (its refill of ``noise``, the previous iteration) and
            # beside this iteration's main graph
            sg = self._gen_replay_stream
            sg.wait_stream(main)
            with torch.cuda.stream(sg):
                self._graph_gen.replay()
                self._gen_event = sg.record_event()
        self._graph.replay()
        if self._graph_gen is not None:
            # whatever the caller enqueues next (a refill of the static inputs, the next replay) comes after the
            # generator's reads; the generator graph is the shorter one, so this wait is normally already satisfied
            main.wait_event(self._gen_event)
        self.d.step_count += 1
        self.iterations += 1
        return self.loss_d, self.loss_g


class MmganTrainer(_TrainerBase):
    """``step`` = one iteration of network_tests.py:281-321 for a MultiModalGAN.

    bf16 mode (and a roll length the kernel supports, T = 50 does): the discriminator's forward, loss and backward run
    as ONE persistent kernel per pass that keeps a sample and all its activations in LDS (``ops.dcnn_fused``), and each
    generator block is one fused Linear+BatchNorm+Sigmoid launch.  fp32 mode: GEMM + im2col lowering (parity path).
    """

    def __init__(self, mmgan, lr=0.01, betas=(0.9, 0.999), eps=1e-8, compute_dtype=None, elide_dead_backward=False,
                 process_group=None, fuse_optimizer=True, exact_bn=False):
        self.mm = mmgan
        self.fuse_optimizer = fuse_optimizer      # one rank, fused bf16 path: Adam + re-pack inside the gradient's slab sum
        # N > 1 ranks: generators' BatchNorm1d statistics over the GLOBAL batch (per-layer exchange of Welford partials)
        # instead of per rank -- generated matrices and running statistics then equal a single process on the whole batch
        self.exact_bn = exact_bn
        d = mmgan.discriminator
        self._init_common([d.conv1.weight, d.conv1.bias, d.conv2.weight, d.conv2.bias, d.fc.weight, d.fc.bias], lr,
                          betas, eps, compute_dtype, elide_dead_backward, process_group)
        self._last_g1 = self._last_g2 = None
        self._pack = None          # packed weight images of the fused discriminator kernel (persistent buffer)
        self._graph = None
        self._graph_gen = None     # one rank: the generators' launches as graphs of their own (see capture)
        self._graph_gen_in = None
        self._gen_event = None     # recorded behind the last replay of that graph
        self._gen_stream = None
        self._gen_replay_stream = None

    # The generators' outputs of the last iteration.  After ``replay`` on one rank they are produced on a stream of the
    # trainer's own: reading them here makes the CURRENT stream wait for that stream's last replay.
    def _join_generators(self):
        if self._gen_event is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_event(self._gen_event)

    @property
    def last_g1(self):
        self._join_generators()
        return self._last_g1

    @property
    def last_g2(self):
        self._join_generators()
        return self._last_g2

    def invalidate_weights(self):
        self._pack = None

    def _refresh_operands(self):
        if self._pack is not None:
            ops.dcnn_pack(*self.d.views, self._pack_t, out=self._pack)

    @staticmethod
    def _layers(gen):
        return [(blk[0].weight.detach(), blk[0].bias.detach(), blk[1].weight.detach(), blk[1].bias.detach(),
                 blk[1].running_mean, blk[1].running_var, blk[1].num_batches_tracked) for blk in gen.gen]

    def _generators_forward(self, noise1, noise2, beats, g1_input, streams=None):
        """Both generators (network_tests.py:186-187).  They are independent chains of four latency-bound
        Linear+BN+Sigmoid launches: with ``streams`` = (s1, s2) each chain runs on its own stream."""
        mm, dt = self.mm, self.dt
        if g1_input is None:   # network_tests.py:83-84: drawn on the CPU generator, then moved
            g1_input = torch.randn(len(noise1), mm.generator1.input_tensor_dim).to(noise1.device)
        s1, s2 = streams if streams is not None else (None, None)
        with torch.cuda.stream(s1 if s1 is not None else torch.cuda.current_stream()):
            x1 = torch.cat((noise1, g1_input), dim=1)
            o1, _ = Fn.mlp_bn_sigmoid_forward(x1, self._layers(mm.generator1), mm.generator1.training, dt,
                                              need_backward=False)
        with torch.cuda.stream(s2 if s2 is not None else torch.cuda.current_stream()):
            x2 = torch.cat((noise2, beats), dim=1)
            o2, _ = Fn.mlp_bn_sigmoid_forward(x2, self._layers(mm.generator2), mm.generator2.training, dt,
                                              need_backward=False)
        a = mm.generator1.adj_size
        return o1.view(len(noise1), -1, a[0], a[1]), o2

    def _fused_ok(self, t):
        return self.dt == ops.BF16 and ops.dcnn_fused_supported(t)

    def _fused_adam_record(self):
        """What ops.dcnn_fused needs to apply Adam itself (one rank): per-parameter views of the flat parameter / moment
        buffers, the device hyper-parameter record (brought up to date; its step counter is advanced by the kernel) and
        the finished-workgroups counter.  None with more than one rank (the gradient is exchanged first)."""
        self._adam_in_kernel = self.world == 1 and self.fuse_optimizer
        if not self._adam_in_kernel:
            return None
        d = self.d
        want = d.sync_hyper(self.lr, self.betas, self.eps, 1.0)
        if d._hyper is None:
            d._hyper = ops.adam_hyper(d.flat.device, *want, step=d.step_count)
            d._hyper_host = want
        if getattr(self, "_adam_views", None) is None:
            offs = [((v.data_ptr() - d.flat.data_ptr()) // 4, v.numel()) for v in d.views]
            self._adam_views = ([d.flat[o:o + n] for o, n in offs], [d.exp_avg[o:o + n] for o, n in offs],
                                [d.exp_avg_sq[o:o + n] for o, n in offs])
            self._adam_done = torch.zeros(1, dtype=torch.int32, device=d.flat.device)
        d.step_count += 1
        p, m, v = self._adam_views
        return dict(params=p, exp_avg=m, exp_avg_sq=v, hyper=d._hyper, done=self._adam_done)

    def _gen_fused_ok(self, b):
        """Whether an iteration's generator work takes the fused chain (staged inputs + one launch per block depth)."""
        mm = self.mm
        return (self.dt == ops.BF16 and 1 < b <= ops.linear_bn_act_max_rows() and mm.generator1.training
                and mm.generator2.training)

    def _gen_inputs(self, noise1, noise2, beats, g1_in_a, g1_in_b):
        """The generators' input rows [noise | conditioning] (network_tests.py:87, 119) in one launch: x1 (2B rows: the
        two forwards of generator 1), x2 (B rows).  The ONLY reader of the caller's noise / beat tensors."""
        b = len(noise1)
        n1, n2, bt = Fn._f32c(noise1), Fn._f32c(noise2), Fn._f32c(beats)
        ia, ib = Fn._f32c(g1_in_a), Fn._f32c(g1_in_b)
        x1 = torch.empty((2 * b, n1.shape[1] + ia.shape[1]), dtype=torch.float32, device=n1.device)
        x2 = torch.empty((b, n2.shape[1] + bt.shape[1]), dtype=torch.float32, device=n1.device)
        ops.concat_cols_multi([(n1, ia), (n1, ib), (n2, bt)], outs=[x1[:b], x1[b:], x2])
        return x1, x2

    def _generators_forward_both(self, noise1, noise2, beats, g1_in_a, g1_in_b, streams, staged=None):
        """Both forwards each generator makes in one iteration (network_tests.py:294 and 312) in ONE chain of four
        launches per generator: generator 1's two input batches are stacked (two BatchNorm groups per launch, running
        statistics updated in call order); the beat generator sees identical inputs both times, so its second forward
        is its first one with the running-statistics update applied twice.  The outputs do not depend on the
        discriminator, so the whole chain runs at the start of the iteration, beside the discriminator step.
        Returns (g1_a, g2_a, g1_b, g2_b); falls back to two sequential forwards where the fused block does not apply."""
        mm, dt = self.mm, self.dt
        b = len(noise1)
        if self.exact_bn and self.world > 1 and mm.generator1.training and mm.generator2.training:
            if g1_in_a is None:
                g1_in_a = torch.randn(b, mm.generator1.input_tensor_dim).to(noise1.device)
            if g1_in_b is None:
                g1_in_b = torch.randn(b, mm.generator1.input_tensor_dim).to(noise1.device)
            a = mm.generator1.adj_size
            outs = []
            for g1_in in (g1_in_a, g1_in_b):     # the reference's call order: G1, G2 (294), then G1, G2 again (312)
                o1 = Fn.mlp_bn_sigmoid_forward_global(torch.cat((noise1, g1_in), dim=1), self._layers(mm.generator1), self.pg)
                o2 = Fn.mlp_bn_sigmoid_forward_global(torch.cat((noise2, beats), dim=1), self._layers(mm.generator2), self.pg)
                outs += [o1.view(b, -1, a[0], a[1]), o2]
            return tuple(outs)
        fused = self._gen_fused_ok(b)
        if not fused:
            g1a, g2a = self._generators_forward(noise1, noise2, beats, g1_in_a, streams)
            g1b, g2b = self._generators_forward(noise1, noise2, beats, g1_in_b, streams)
            return g1a, g2a, g1b, g2b
        if g1_in_a is None:    # network_tests.py:83-84: drawn on the CPU generator, call by call
            g1_in_a = torch.randn(b, mm.generator1.input_tensor_dim).to(noise1.device)
        if g1_in_b is None:
            g1_in_b = torch.randn(b, mm.generator1.input_tensor_dim).to(noise1.device)
        s1 = streams[0] if streams is not None else None
        l1, l2 = self._layers(mm.generator1), self._layers(mm.generator2)
        with torch.cuda.stream(s1 if s1 is not None else torch.cuda.current_stream()):
            # the three input concatenations in one launch, then the k-th blocks of BOTH generators in one launch
            # (same depth, independent): 5 launches for an iteration's generator work
            x1, x2 = self._gen_inputs(noise1, noise2, beats, g1_in_a, g1_in_b) if staged is None else staged
            if len(l1) == len(l2):
                for (w1_, b1_, g1_, be1_, rm1, rv1, nb1), (w2_, b2_, g2_, be2_, rm2, rv2, nb2) in zip(l1, l2):
                    (x1, _, _), (x2, _, _) = ops.linear_bn_act_fwd_multi(
                        [dict(x=x1, w=w1_, bias=b1_, gamma=g1_, beta=be1_, running_mean=rm1, running_var=rv1, nbt=nb1,
                              groups=2),
                         dict(x=x2, w=w2_, bias=b2_, gamma=g2_, beta=be2_, running_mean=rm2, running_var=rv2, nbt=nb2,
                              stat_repeats=2)], act=ops.ACT_SIGMOID, training=True)
                o1, o2 = x1, x2
            else:
                o1, _ = Fn.mlp_bn_sigmoid_forward(x1, l1, True, dt, need_backward=False, groups=2)
                o2, _ = Fn.mlp_bn_sigmoid_forward(x2, l2, True, dt, need_backward=False, stat_repeats=2)
        a = mm.generator1.adj_size
        g1 = o1.view(2, b, -1, a[0], a[1])
        return g1[0], o2, g1[1], o2

    @torch.no_grad()
    def step(self, piano_roll, durations, beats, noise1, noise2, fake_a, fake_b, g1_in_a=None, g1_in_b=None):
        """fake_a / fake_b: (B,2,128,T) tensors or callables ``f(g1_out, g2_out) -> tensor`` standing in for the
        DES bridge of the D-step and G-step forwards (network_tests.py:294, 312).

        The iteration is three pieces -- everything up to the gradient (``_part_a``), the data-parallel exchange, Adam and
        the generator step (``_part_b``) -- so that with more than one rank the two compute pieces can be replayed as
        hipGraphs around the eager collective (``capture`` / ``replay``)."""
        self._watch(piano_roll, durations, beats, noise1, noise2, fake_a, fake_b, g1_in_a, g1_in_b)
        self._part_a(piano_roll, durations, beats, noise1, noise2, fake_a, g1_in_a, g1_in_b)
        self._reduce()
        self._part_b(piano_roll, beats, noise1, noise2, fake_b, g1_in_b)
        self.iterations += 1
        self._anomaly_check()
        return self.loss_d, self.loss_g

    def _sides(self, dev):
        if self._gen_stream is None:
            self._gen_stream = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        return self._gen_stream

    def _part_a(self, piano_roll, durations, beats, noise1, noise2, fake_a, g1_in_a, g1_in_b, with_generators=True):
        """D step up to the gradient (network_tests.py:293-307): the generators (both forwards of the iteration, see
        _generators_forward_both) on side streams beside the discriminator's forward + loss + backward; every branch
        is joined before returning (one rank: at the end of ``_part_b``).  ``with_generators=False`` leaves the
        generators out (``capture`` records them as a graph of their own)."""
        dt = self.dt
        w1, b1, w2, b2, wf, bf = self.d.views
        gv = self.d.grad_views
        b = piano_roll.shape[0]
        t = piano_roll.shape[2]
        dev = piano_roll.device
        fused = self._fused_ok(t)
        # the generators only feed the (external) bridge: each runs on a side stream of its own beside the
        # discriminator kernels (a generator's second forward follows its first one: BN running statistics)
        main = torch.cuda.current_stream()
        sides = self._sides(dev)

        def generators():
            for sd in sides:
                sd.wait_stream(torch.cuda.current_stream())
            g1, g2, g1b, g2b = self._generators_forward_both(noise1, noise2, beats, g1_in_a, g1_in_b, sides)
            self._last_g1, self._last_g2 = g1, g2
            self._gen_b = (g1b, g2b)       # second forward's outputs: the bridge of the generator step consumes them
            return g1, g2

        # A callable bridge needs the generators' output first.  With tensor stand-ins nothing in the iteration reads
        # it, and the chain is forked AFTER the discriminator kernel's launch: that kernel owns the LDS of every CU it
        # runs on, so generator launches beside it only wait for CUs; behind it they fill the small-kernel stretch of
        # the iteration (slab sums, Adam, re-pack).
        gen_late = not callable(fake_a)
        if not with_generators:
            gen_late = False
        elif not gen_late:
            # (a branch forked at the very root of a captured graph was observed to run before, not beside, the main
            # branch: fork after a first small launch on the main stream)
            self.d.extra[1:4].zero_()
            g1, g2 = generators()
            for sd in sides:
                main.wait_stream(sd)
            fake_a = fake_a(g1, g2)
        if fused:
            if self._pack is None:
                self._pack = ops.dcnn_pack(w1, b1, w2, b2, wf, bf, t)
                self._pack_t = t
            fa = Fn._f32c(fake_a)
            # batch [fake ; real] with labels 0 / 1 (304-305); real_data is read as two planes: no stack/permute copy.
            # One rank: disc_opt.step() (308) and the refresh of the packed weights ride the gradient's final summation
            # (slab sum -> adam_prep -> Adam -> re-pack were four launches on the iteration's critical chain).
            ops.dcnn_fused(fa, (Fn._f32c(piano_roll), Fn._f32c(durations)), t, 0.0, 1.0, self._pack,
                           loss_out=self.loss_d, grad_out=gv, adam=self._fused_adam_record())
            if gen_late:
                generators()
        else:
            if gen_late:
                generators()
            x = torch.empty((2 * b, 2) + tuple(piano_roll.shape[1:]), dtype=torch.float32, device=dev)
            x[:b].copy_(fake_a)                        # [fake ; real]: same order as the two loss terms (304-305)
            x[b:, 0].copy_(piano_roll)                 # real_data = stack([roll, dur]).permute(1,0,2,3) (290)
            x[b:, 1].copy_(durations)
            logits, saved = Fn.dcnn_forward(x, w1, b1, w2, b2, wf, bf, dt)
            lg = logits.view(-1)
            dl = torch.empty(2 * b, dtype=torch.float32, device=dev)
            ops.bce_with_logits(lg[:b], 0.0, loss_out=self.loss_d, dx_out=dl[:b])
            ops.bce_with_logits(lg[b:], 1.0, loss_out=self.loss_d, dx_out=dl[b:], accumulate_loss=True)
            grads = Fn.dcnn_backward(saved, dl, w2, wf, dt)[:6]
            for gview, g in zip(gv, grads):
                gview.copy_(g.view(gview.shape))
        # one rank: the late generator chains are joined at the end of the iteration (_part_b); with more ranks this
        # piece is a graph of its own and has to join its branches itself
        self._gen_join_pending = gen_late and self.world == 1
        if with_generators and not self._gen_join_pending:
            for sd in sides:
                main.wait_stream(sd)

    def _part_b(self, piano_roll, beats, noise1, noise2, fake_b, g1_in_b):
        """Adam (308), then the "G" step (311-315): D forward on the bridge's output for the generators' second forward
        (which ran at the start of the iteration: _generators_forward_both) + the dead backward in faithful mode."""
        dt = self.dt
        w1, b1, w2, b2, wf, bf = self.d.views
        gv = self.d.grad_views
        t = piano_roll.shape[2]
        fused = self._fused_ok(t)
        if not (fused and getattr(self, "_adam_in_kernel", False)):      # (else: already applied by _part_a's kernel)
            self._adam()
            if fused:
                ops.dcnn_pack(w1, b1, w2, b2, wf, bf, t, out=self._pack)     # weights changed: refresh in place
        if callable(fake_b):
            if getattr(self, "_gen_join_pending", False):
                # mixed bridge (tensor fake_a, callable fake_b): the generator chains were forked late and are still
                # running on their side streams -- the bridge reads their outputs on this stream
                main = torch.cuda.current_stream()
                for sd in self._sides(piano_roll.device):
                    main.wait_stream(sd)
                self._gen_join_pending = False
            fake_b = fake_b(*self._gen_b)      # the generators' second forward ran at the start of the iteration
        if fused:
            if self.elide:
                ops.dcnn_fused(Fn._f32c(fake_b), None, t, 1.0, 1.0, self._pack, loss_out=self.loss_g, want_grad=False)
            else:   # dead values (only D's .grad in the reference, wiped by the next zero_grad): scratch buffers
                if getattr(self, "_scratch_grads", None) is None:
                    self._scratch_grads = [torch.empty_like(g) for g in gv]
                ops.dcnn_fused(Fn._f32c(fake_b), None, t, 1.0, 1.0, self._pack, loss_out=self.loss_g,
                               grad_out=self._scratch_grads)
        else:
            logits_g, saved_g = Fn.dcnn_forward(fake_b, w1, b1, w2, b2, wf, bf, dt)
            if self.elide:
                ops.bce_with_logits(logits_g.view(-1), 1.0, loss_out=self.loss_g, want_grad=False)
            else:
                _, dlg = ops.bce_with_logits(logits_g.view(-1), 1.0, loss_out=self.loss_g)
                Fn.dcnn_backward(saved_g, dlg, w2, wf, dt)     # dead values (only D's .grad in the reference)
        if getattr(self, "_gen_join_pending", False):
            main = torch.cuda.current_stream()
            for sd in self._sides(piano_roll.device):
                main.wait_stream(sd)
            self._gen_join_pending = False

    # ---- hipGraph capture for fixed input buffers --------------------------------------------------------------------
    def capture(self, piano_roll, durations, beats, noise1, noise2, fake_a, fake_b, g1_in_a, g1_in_b):
        """Record one iteration on fixed tensors (the iteration is ~130 small launches: replay removes the host launch
        cost).  Needs tensor bridge outputs and explicit generator-1 inputs (the reference draws them on the CPU
        generator, network_tests.py:83-84, which cannot be part of a device graph).

        One rank: the whole iteration is ONE hipGraph.  More ranks: TWO graphs -- everything up to the gradient, and
        Adam + the generator step -- with the gradient all-reduce issued eagerly between them (a collective is not
        part of a graph here), so a data-parallel rank replays 2 graphs + 1 collective per iteration instead of ~130
        eager launches."""
        args = (piano_roll, durations, beats, noise1, noise2, fake_a, fake_b, g1_in_a, g1_in_b)
        if any(callable(a) for a in args) or any(a is None for a in args):
            raise ops.GdmError("graph capture needs tensor inputs (incl. g1_in_a/g1_in_b)")
        self._static = args
        warm = torch.cuda.Stream(piano_roll.device)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for _ in range(2):
                self.step(*args[:7], g1_in_a=g1_in_a, g1_in_b=g1_in_b)
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        if self.world == 1:
            # TWO graphs without a single fork or join: the discriminator chain (zero, kernel, slab sum, Adam, re-pack,
            # kernel, slab sum) and the generators' eight launches.  Inside one multi-branch graph every edge between
            # branches became a cross-queue dependency on the GPU: 20-35 us each on the discriminator chain (66 us
            # of a 174-us iteration were such gaps).  ``replay`` starts the generator graph on a stream of the
            # trainer's own beside this iteration's discriminator graph; the caller's stream waits for it only after
            # the discriminator chain (the generators' graph is the shorter one).
            # (the generators' graph is two: the one-launch input staging, the only reader of the caller's tensors, and
            # the four block launches -- the caller's stream waits for the FIRST only, see replay)
            # Only the fused generator chain reads the staged rows; the fallback chain (fp32, B > the fused block's row
            # limit, eval-mode generators) concatenates the caller's tensors itself, inside the block graph -- then
            # there is no staging graph and ``replay`` makes the caller's stream wait for the WHOLE generator graph
            # before a refill of the inputs may follow.
            self._gen_staged = self._gen_fused_ok(len(noise1))
            self._graph_gen_in = None
            self._graph_gen = torch.cuda.CUDAGraph()
            self._graph = torch.cuda.CUDAGraph()
            ns_gen, ns_main = ("graph", id(self._graph_gen)), ("graph", id(self._graph))
            staged, pool = None, None
            if self._gen_staged:
                self._graph_gen_in = torch.cuda.CUDAGraph()
                with ops.workspace_namespace(ns_gen), torch.cuda.graph(self._graph_gen_in):
                    staged = self._gen_inputs(noise1, noise2, beats, g1_in_a, g1_in_b)
                pool = self._graph_gen_in.pool()
            with ops.workspace_namespace(ns_gen), torch.cuda.graph(self._graph_gen, pool=pool):
                g1, g2, g1b, g2b = self._generators_forward_both(noise1, noise2, beats, g1_in_a, g1_in_b, None,
                                                                 staged=staged)
                self._last_g1, self._last_g2 = g1, g2
                self._gen_b = (g1b, g2b)
            with ops.workspace_namespace(ns_main), torch.cuda.graph(self._graph):
                self._part_a(piano_roll, durations, beats, noise1, noise2, fake_a, g1_in_a, g1_in_b,
                             with_generators=False)
                self._part_b(piano_roll, beats, noise1, noise2, fake_b, g1_in_b)
            self._gen_replay_stream = torch.cuda.Stream(piano_roll.device)
        else:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with ops.workspace_namespace(("graph", id(ga))), torch.cuda.graph(ga):
                self._part_a(piano_roll, durations, beats, noise1, noise2, fake_a, g1_in_a, g1_in_b)
            pool = ga.pool()
            with ops.workspace_namespace(("graph", id(ga))), torch.cuda.graph(gb, pool=pool):   # (sequential: shared)
                self._part_b(piano_roll, beats, noise1, noise2, fake_b, g1_in_b)
            self._graph = (ga, gb)
        self.d.step_count -= 1     # the captured call did not execute: the device-side step counter did not move
        return self._graph

    def replay(self):
        self._sync_hyper()         # lr schedule etc.: the captured Adam reads the device record
        if isinstance(self._graph, tuple):
            self._graph[0].replay()
            self._reduce()
            self._graph[1].replay()
        else:
            # the generator graph: behind everything the caller has enqueued so far (its refill of the static inputs, the
            # previous iteration), beside this iteration's discriminator graph
            main = torch.cuda.current_stream()
            sg = self._gen_replay_stream
            sg.wait_stream(main)
            with torch.cuda.stream(sg):
                staged_ev = None
                if self._graph_gen_in is not None:
                    self._graph_gen_in.replay()
                    staged_ev = sg.record_event()
                self._graph_gen.replay()
                self._gen_event = sg.record_event()
            self._graph.replay()
            # whatever the caller enqueues next (a refill of the inputs, the next replay) comes after the generators'
            # READS of the caller's tensors: the one staging launch at the head of their stream.  (Waiting for their whole
            # graph cost 4 % at B = 256 and 17 % at B = 16, where the generators' chain is the longer one.)  Without a
            # staging graph (fallback generator chain) the block graph itself reads them: wait for all of it.
            main.wait_event(staged_ev if staged_ev is not None else self._gen_event)
        self.d.step_count += 1
        self.iterations += 1
        return self.loss_d, self.loss_g


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size, gamma) for a trainer (network_tests.py:257-258, 328-329)."""

    def __init__(self, trainer, step_size, gamma=0.1):
        self.trainer, self.step_size, self.gamma = trainer, step_size, gamma
        self.base_lr = trainer.lr
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        self.trainer.lr = self.base_lr * self.gamma ** (self.last_epoch // self.step_size)

    def get_last_lr(self):
        return [self.trainer.lr]
