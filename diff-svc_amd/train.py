"""``DiffusionTrainerHip`` -- the training step of the reference (BASELINE configs[4]) on the HIP kernels:
``GaussianDiffusion.forward(infer=False)`` -> ``p_losses`` (network/diff/diffusion.py:200-225,237-241) with the gradients of
every ``denoise_fn.*`` parameter and of ``fs2.pitch_embed.weight``, gradient-norm clipping (utils/pl_utils.py:1081-1084),
AdamW and the StepLR schedule (training/task/SVC_task.py:60-66,116-125), and data-parallel training as the reference does it
(utils/pl_utils.py:179-221: one process per GPU, gradients averaged over ranks) through ``torch.distributed`` (backend ``nccl`` = RCCL
over xGMI; ``gloo`` in the CPU tests): the flat gradient buffer is all-reduced in BUCKETS -- the tail, groups of residual layers from the
top down, the head -- each launched as soon as ``dsvc_trainer_step_begin / _layers / _end`` has made its slice final, so the exchange of
layers 19..15 runs on RCCL's stream under the backward pass of layers 14..0 (what the reference's DDP reducer does, pl_utils.py:187-221).

Parameters and gradients are two flat fp32 device tensors owned by this object; the C ABI (include/dsvc.h, dsvc_trainer_*)
reads / writes them in place, ``state_dict()`` exposes the reference's key names and shapes, so a checkpoint written from it
loads into the reference (and into the inference drop-ins) unchanged.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .cond import CondBuilder


class TrainerHandle:
    def __init__(self, hp, loss_type="l2", pitch_vocab=300):
        self._h = ctypes.c_void_p(0)
        self._L = lib()                                    # the library this handle lives in (product, or the test-hooks build: _lib.hooks_build)
        self._inflight = None                              # tensors of the step between step_begin() and step_end()
        self.cfg = _lib.TrainerCfg(hp["audio_num_mel_bins"], hp["hidden_size"], hp["residual_channels"], hp["residual_layers"],
                                   hp["dilation_cycle_length"], int(hp.get("timesteps", 1000)), 1 if loss_type == "l1" else 0, pitch_vocab)
        self._ck(self._L.dsvc_trainer_create(ctypes.byref(self.cfg), ctypes.byref(self._h)))
        nt, nf = ctypes.c_int64(0), ctypes.c_int64(0)
        self._ck(self._L.dsvc_trainer_param_count(self._h, ctypes.byref(nt), ctypes.byref(nf)))
        self.n_floats = nf.value
        self.layout = []                                   # (name, offset, numel) in flat order
        for i in range(nt.value):
            name, off, n = ctypes.c_char_p(), ctypes.c_int64(0), ctypes.c_int64(0)
            self._ck(self._L.dsvc_trainer_param_info(self._h, i, ctypes.byref(name), ctypes.byref(off), ctypes.byref(n)))
            self.layout.append((name.value.decode(), off.value, n.value))

    def _ck(self, rc):
        check(rc, self._L)

    def check(self):
        """Wait for the current stream and raise if a step since the last check was given a diffusion step outside [0, timesteps) (steps are
        clamped on the device; the reference's extract() would raise an IndexError)."""
        self._ck(self._L.dsvc_trainer_check(self._h, stream_ptr()))

    def debug_set(self, key, value):
        """Test support (include/dsvc_debug.h: dsvc_trainer_debug_set), e.g. ``("wgrad_fm", 0)``: the k_split_t weight-gradient path."""
        self._ck(self._L.dsvc_trainer_debug_set(self._h, key.encode(), int(value)))

    def bind(self, params, grads):
        assert params.is_cuda and grads.is_cuda and params.numel() == self.n_floats == grads.numel()
        self._keep = (params, grads)
        self._ck(self._L.dsvc_trainer_bind(self._h, ptr(params), ptr(grads)))

    def set_schedule(self, sqrt_ac, sqrt_1mac, spec_min, spec_max):
        a, b = sqrt_ac.detach().cpu().float().contiguous(), sqrt_1mac.detach().cpu().float().contiguous()
        lo, hi = spec_min.detach().cpu().float().reshape(-1).contiguous(), spec_max.detach().cpu().float().reshape(-1).contiguous()
        self._ck(self._L.dsvc_trainer_set_schedule(self._h, ptr(a), ptr(b), a.numel(), ptr(lo), ptr(hi), lo.numel()))

    def _args(self, mel, cond, t, pitch, mel2ph, seed, first_clip, clip_ids):
        B, T, M = mel.shape
        i32 = lambda x: x.to(torch.int32).contiguous() if x is not None else None
        mel, cond = mel.contiguous().float(), cond.contiguous().float()
        t, pitch, mel2ph, clip_ids = i32(t), i32(pitch), i32(mel2ph), i32(clip_ids)
        for x in (mel, cond, t, pitch, mel2ph, clip_ids):
            if x is not None and not x.is_cuda:
                raise RuntimeError("diffsvc_amd: training tensors must live on the HIP device; there is no CPU path")
        # (diffusion steps outside [0, timesteps) are clamped inside dsvc_trainer_step -- k_make_xt reads the noise schedule at t --
        #  without a device-to-host check; train_step() draws them in range)
        if tuple(cond.shape) != (B, self.cfg.hidden, T) or t.numel() != B:
            raise ValueError("shape mismatch: mel %s cond %s t %s" % (tuple(mel.shape), tuple(cond.shape), tuple(t.shape)))
        a = _lib.TrainArgs(B, T, mel.data_ptr(), cond.data_ptr(), t.data_ptr(), pitch.data_ptr() if pitch is not None else None,
                           mel2ph.data_ptr() if mel2ph is not None else None, seed, first_clip,
                           clip_ids.data_ptr() if clip_ids is not None else None)
        return a, (mel, cond, t, pitch, mel2ph, clip_ids)            # the tensors stay referenced while the step is enqueued

    def step(self, mel, cond, t, pitch=None, mel2ph=None, seed=0, first_clip=0, clip_ids=None, loss_out=None):
        a, keep = self._args(mel, cond, t, pitch, mel2ph, seed, first_clip, clip_ids)
        loss = loss_out if loss_out is not None else torch.empty(1, device=mel.device, dtype=torch.float32)
        self._ck(self._L.dsvc_trainer_step(self._h, ctypes.byref(a), ptr(loss), stream_ptr()))
        return loss

    # the same step in phases (include/dsvc.h): after each call a contiguous slice of the gradient buffer is final
    def step_begin(self, mel, cond, t, pitch=None, mel2ph=None, seed=0, first_clip=0, clip_ids=None):
        a, self._inflight = self._args(mel, cond, t, pitch, mel2ph, seed, first_clip, clip_ids)
        self._ck(self._L.dsvc_trainer_step_begin(self._h, ctypes.byref(a), stream_ptr()))

    def step_layers(self, l_hi, l_lo):
        self._ck(self._L.dsvc_trainer_step_layers(self._h, int(l_hi), int(l_lo), stream_ptr()))

    def step_end(self, loss_out=None):
        if self._inflight is None:
            raise RuntimeError("diffsvc_amd: step_end() without a step in flight (call step_begin first)")
        loss = loss_out if loss_out is not None else torch.empty(1, device=self._inflight[0].device, dtype=torch.float32)
        self._ck(self._L.dsvc_trainer_step_end(self._h, ptr(loss), stream_ptr()))
        self._inflight = None
        return loss

    def __del__(self):
        try:
            if self._h:
                self._L.dsvc_trainer_destroy(self._h)
                self._h = ctypes.c_void_p(0)
        except Exception:
            pass


def gradient_buckets(layout, n_layers, layers_per_bucket=5):
    """The order in which the phased step makes the flat gradient buffer final, as (phase, l_hi, l_lo, [(offset, numel), ...]):
    ('begin', -, -, tail slice), ('layers', hi, lo, slice) from the top layer down, ('end', -, -, head slice + pitch embedding).
    The slices partition [0, n_floats) (tests/test_host.py)."""
    off = {name: (o, n) for name, o, n in layout}
    span = lambda first, last: (off[first][0], off[last][0] + off[last][1] - off[first][0])
    out = [("begin", 0, 0, [span("denoise_fn.skip_projection.weight", "denoise_fn.output_projection.bias")])]
    hi = n_layers
    while hi > 0:
        lo = max(0, hi - layers_per_bucket)
        out.append(("layers", hi, lo, [span("denoise_fn.residual_layers.%d.dilated_conv.weight" % lo,
                                            "denoise_fn.residual_layers.%d.output_projection.bias" % (hi - 1))]))
        hi = lo
    out.append(("end", 0, 0, [span("denoise_fn.input_projection.weight", "denoise_fn.mlp.2.bias"), off["fs2.pitch_embed.weight"]]))
    return out


def allreduce_mean_(flat, group=None):
    """The one collective of data-parallel training: average the flat gradient buffer over the ranks in place (what the
    reference's DDP reducer does bucket by bucket, utils/pl_utils.py:187-221).  One fp32 all-reduce of 32 M floats = 128 MB."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    return flat


class DiffusionTrainerHip:
    """state: a GaussianDiffusion state dict (no ``model.`` prefix).  hp keys read beyond the architecture: ``diff_loss_type``,
    ``lr``, ``optimizer_adam_beta1/2``, ``weight_decay``, ``clip_grad_norm``, ``decay_steps`` (training/config_nsf.yaml)."""

    def __init__(self, hp, state, device="cuda", group=None):
        if not torch.cuda.is_available():
            raise RuntimeError("DiffusionTrainerHip needs a HIP device (there is no CPU path)")
        self.hp, self.group, self.device = hp, group, device
        self.h = TrainerHandle(hp, hp.get("diff_loss_type", "l2"), state["fs2.pitch_embed.weight"].shape[0])
        n = self.h.n_floats
        self.params = torch.zeros(n, device=device, dtype=torch.float32)
        self.grads = torch.zeros(n, device=device, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=device, dtype=torch.float32)
        self._aux = torch.zeros(4, device=device, dtype=torch.float32)          # [0] ||g||^2, [1] clip coefficient
        self.shapes = {}
        for name, off, numel in self.h.layout:
            v = state[name]
            assert v.numel() == numel, (name, tuple(v.shape), numel)
            self.params[off:off + numel].copy_(v.reshape(-1).float())
            self.shapes[name] = tuple(v.shape)
        self.other = {k: v.clone() for k, v in state.items() if k not in self.shapes}     # buffers + the fs2.* tensors no gradient reaches
        self.h.bind(self.params, self.grads)
        self.h.set_schedule(state["sqrt_alphas_cumprod"], state["sqrt_one_minus_alphas_cumprod"], state["spec_min"], state["spec_max"])
        self.fs2 = CondBuilder(hp).to(device)
        off, numel = next((o, m) for nme, o, m in self.h.layout if nme == "fs2.pitch_embed.weight")
        self.fs2.pitch_embed.weight.data = self.params[off:off + numel].view(self.shapes["fs2.pitch_embed.weight"])   # shares the flat storage
        self.global_step = 0
        self.lr0 = float(hp.get("lr", 0.0004))

    def view(self, flat, name):
        off, numel = next((o, m) for nme, o, m in self.h.layout if nme == name)
        return flat[off:off + numel].view(self.shapes[name])

    def state_dict(self):
        sd = {k: v.clone() for k, v in self.other.items()}
        for name, off, numel in self.h.layout:
            sd[name] = self.params[off:off + numel].view(self.shapes[name]).detach().clone().cpu()
        return sd

    def lr(self):
        """The rate the NEXT optimisation step runs at.  The reference steps its StepLR(decay_steps, gamma=0.5) AFTER optimizer.step() and
        with an explicit epoch -- ``scheduler.step(global_step)`` (SVC_task.py:119-125, global_step still k while step k is being
        taken, pl_utils.py:1426) -- which sets lr0 * 0.5 ** (k // decay_steps) for step k + 1: step k runs at
        lr0 * 0.5 ** ((k - 1) // decay_steps), one step later than a plain per-step StepLR would halve."""
        return self.lr0 * 0.5 ** (max(self.global_step - 1, 0) // int(self.hp.get("decay_steps", 40000)))

    @torch.no_grad()
    def forward_backward(self, hubert, mel2ph, f0, mels, t, seed=0, first_clip=0, clip_ids=None):
        """loss (device scalar) of one batch; the gradients land in ``self.grads`` (this rank's, before any all-reduce)."""
        ret = self.fs2(hubert, mel2ph, None, None, f0.clone(), None, None, infer=False)
        cond = ret["decoder_inp"].transpose(1, 2).contiguous()
        return self.h.step(mels, cond, t, pitch=ret["pitch_pred"].squeeze(-1), mel2ph=mel2ph, seed=seed, first_clip=first_clip, clip_ids=clip_ids)

    @torch.no_grad()
    def forward_backward_overlapped(self, hubert, mel2ph, f0, mels, t, seed=0, first_clip=0, clip_ids=None, layers_per_bucket=5):
        """forward_backward + the gradient all-reduce (mean), bucket by bucket under the backward pass: every bucket's all-reduce is
        launched asynchronously (RCCL's own stream, ordered after the kernels that made the slice final) and waited for at the end.
        ``self.grads`` holds the averaged gradients on return."""
        import torch.distributed as dist
        ret = self.fs2(hubert, mel2ph, None, None, f0.clone(), None, None, infer=False)
        cond = ret["decoder_inp"].transpose(1, 2).contiguous()
        world = dist.get_world_size(self.group)
        works, loss = [], None
        for phase, hi, lo, slices in gradient_buckets(self.h.layout, int(self.hp["residual_layers"]), layers_per_bucket):
            if phase == "begin":
                self.h.step_begin(mels, cond, t, pitch=ret["pitch_pred"].squeeze(-1), mel2ph=mel2ph, seed=seed, first_clip=first_clip, clip_ids=clip_ids)
            elif phase == "layers":
                self.h.step_layers(hi, lo)
            else:
                loss = self.h.step_end()
            for o, n in slices:
                works.append(dist.all_reduce(self.grads[o:o + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if world > 1:
            self.grads.mul_(1.0 / world)
        return loss

    CHECK_EVERY = 100              # optimiser steps between two reads of the trainer's sticky argument flag (dsvc_trainer_check synchronises)

    @torch.no_grad()
    def train_step(self, hubert, mel2ph, f0, mels, t=None, seed=None, first_clip=0, clip_ids=None, overlap=None):
        """One optimisation step: forward + backward, all-reduce (mean) of the gradients over the ranks, gradient-norm clip, AdamW.
        overlap: all-reduce bucket by bucket under the backward pass (default: whenever a process group with more than one rank is up)."""
        import torch.distributed as dist
        B = mels.shape[0]
        if t is None:
            t = torch.randint(0, int(self.hp.get("K_step", self.hp["timesteps"])), (B,), device=mels.device)       # train_pipeline.py:233
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        ddp = dist.is_available() and dist.is_initialized()
        if overlap is None:
            overlap = ddp and dist.get_world_size(self.group) > 1
        if overlap and ddp:
            loss = self.forward_backward_overlapped(hubert, mel2ph, f0, mels, t, seed=seed, first_clip=first_clip, clip_ids=clip_ids)
            self.optimizer_step(reduced=True)
        else:
            loss = self.forward_backward(hubert, mel2ph, f0, mels, t, seed=seed, first_clip=first_clip, clip_ids=clip_ids)
            self.optimizer_step()
        return loss

    @torch.no_grad()
    def optimizer_step(self, reduced=False):
        """What follows the backward pass: all-reduce (mean) of ``self.grads`` over the ranks (unless the overlapped backward pass has done
        it already), clip_grad_norm_, AdamW at the StepLR rate."""
        if not reduced:
            allreduce_mean_(self.grads, self.group)
        lr = self.lr()                                       # step k (0-based) runs at lr0 * 0.5 ** (max(k - 1, 0) // decay_steps)
        self.global_step += 1
        if self.global_step % self.CHECK_EVERY == 0:
            self.h.check()                                   # the trainer's sticky "diffusion step outside the schedule" flag: one sync per 100 steps
        hp, n = self.hp, self.h.n_floats
        clip = float(hp.get("clip_grad_norm", 1.0))
        coef = None
        if clip > 0:
            check(lib().dsvc_grad_clip_coef(ptr(self.grads), n, clip, ptr(self._aux[0:1]), ptr(self._aux[1:2]), stream_ptr()))
            coef = self._aux[1:2]
        check(lib().dsvc_adamw_step(ptr(self.params), ptr(self.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), n, lr,
                                    float(hp.get("optimizer_adam_beta1", 0.9)), float(hp.get("optimizer_adam_beta2", 0.98)), 1e-8,
                                    float(hp.get("weight_decay", 0.0)), self.global_step, ptr(coef), 1.0, stream_ptr()))
