"""Training side of the hot path (BASELINE config 4): ``Trainer.training_step`` of the reference
(train.py:258-268) = q_sample -> ``model(x=x_t, step=t, image=images, pred_type="denoise")`` under autograd -> loss ->
backward -> AdamW, one process per GPU with the gradients averaged over RCCL.

Everything between the parameters and the logits runs on this package's HIP kernels inside ``torch.autograd.Function``s
(autograd is the tape, not the arithmetic): every 3x3x3 convolution forward / data gradient / weight gradient,
InstanceNorm + LeakyReLU + temb / embedding adds forward and backward, MaxPool forward and backward, the k2s2
transposed convolution with its concat in place, the 1x1x1 head, and the fused mse/bce/dice loss.  Activations are
channels-last [N, D, H, W, C] in the compute dtype (fp16 with fp32 master weights and dynamic loss scaling, or fp32).
``Diffusion.denoise`` dispatches here whenever grad mode is on, so the reference's training loop runs unchanged through
``DiffUNet.forward``.  The timestep embedding (TimeStepEmbedder + the nine temb_proj, forward and backward), the loss tail,
q_sample on 2 * label - 1, the overflow check, AdamW and the loss-scale update are library kernels too (csrc/train_glue.hip);
torch supplies the tape, the allocator, the random noise and the collectives.  There is no torch / MIOpen convolution path.
"""
from __future__ import annotations

import torch

uses_native_kernels = True


class _TembState:
    """What the TwoConv blocks of one evaluation share about their timestep-embedding adds: every block reads its [N, cout]
    rows of ONE block-major buffer (ops.temb_train_fwd) and, in backward, writes the gradient of its rows into ONE buffer
    of the same layout; the block that ran FIRST in forward -- whose backward node can only run after every other block's,
    since all of them consume its output -- hands that buffer to autograd as the gradient of the whole add tensor (the
    others return None: the engine counts a node's dependencies whether or not a gradient came with them).  Nine slice
    views would have cost nine zero-fills, copies and accumulations of a [N, 1536] tensor per step instead."""

    def __init__(self, N, couts):
        self.N, self.couts = N, list(couts)
        self.offs = [sum(self.couts[:i]) for i in range(len(self.couts))]
        self.P = sum(self.couts)
        self.dadd = None
        self.next = 0          # forward call counter: block i of this evaluation
        self.written = 0

    def rows(self, flat, i):
        a = self.N * self.offs[i]
        return flat[a:a + self.N * self.couts[i]].view(self.N, self.couts[i])


class _TembAdds(torch.autograd.Function):
    """add_b = temb_proj_b(swish(TimeStepEmbedder(t))) for every TwoConv block b (models/diffusion/utils.py:5-54,
    denoiser.py:51-52,65): three launches forward (dua_temb_train_fwd), two backward (dua_temb_train_bwd) for the eleven Linear
    layers.  Inputs after the state: dense[0].weight, .bias, dense[1].weight, .bias, then (weight, bias) of each temb_proj."""

    @staticmethod
    def forward(ctx, t, half, state, *params):
        from . import ops
        w0, b0, w1, b1 = (p.detach() for p in params[:4])
        pw = [p.detach() for p in params[4::2]]
        pb = [p.detach() for p in params[5::2]]
        add, saved = ops.temb_train_fwd(t.contiguous(), half, w0, b0, w1, b1, pw, pb)
        ctx.save_for_backward(saved, w1, *pw)
        ctx.half = half
        return add

    @staticmethod
    def backward(ctx, dadd):
        from . import ops
        saved, w1, *pw = ctx.saved_tensors
        dw0, db0, dw1, db1, dws, dbs = ops.temb_train_bwd(dadd.contiguous(), saved, ctx.half, w1, pw)
        out = [None, None, None, dw0, db0, dw1, db1]
        for dw, db in zip(dws, dbs):
            out += [dw, db]
        return tuple(out)


def parse_losses(losses="mse,bce,dice", loss_combine="sum"):
    """losses/loss.py:25-62: comma-separated loss names and the combine rule; the names the HIP loss kernels cover."""
    names = tuple(losses.split(","))
    from .ops import LOSS_NAMES
    for n in names:
        if n not in LOSS_NAMES:
            raise NotImplementedError(f"Loss ({n}) is not listed yet")
    if loss_combine not in ("sum", "mean", "log"):
        raise NotImplementedError("Unsupported value for loss_combine. Please choose from 'sum', 'mean', or 'log'.")
    return names, loss_combine


def allreduce_mean_(tensors, group=None):
    """Average ``tensors`` over the ranks of ``group`` with ONE flat all-reduce (RCCL on a GPU node, gloo in the CPU
    tests); in place.  No-op outside torch.distributed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()


# ------------------------------------------------------------------------------------------------------------
# ---- weight gradients on a side stream -------------------------------------------------------------------------------
# A layer's weight gradient needs its input and its output gradient and nothing downstream needs it before the
# optimizer: the MFMA-bound wgrad kernels (4.5 ms of a step) run on a second stream next to the rest of the backward
# chain, whose InstanceNorm / pooling / concat passes are HBM-bound and whose coarse levels leave the chip idle.  The
# operands are kept referenced until the join (the caching allocator must not hand their memory to the main stream while
# the side stream still reads it); the accumulator itself is what autograd takes over as ``param.grad`` without touching it.
_WGRAD = {"stream": None, "keep": []}


def _wgrad(x, dy, cout, dw):
    from . import ops
    st = _WGRAD["stream"]
    if st is None:
        ops.conv3d_k3_wgrad(x, x.shape[-1], 0, dy, cout, 0, dw)
        return
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ops.conv3d_k3_wgrad(x, x.shape[-1], 0, dy, cout, 0, dw)
    _WGRAD["keep"].append((x, dy))


class _wgrad_side:
    """with _wgrad_side(stream): backward()  -- joins the side stream and drops the kept operands on exit."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        _WGRAD["stream"] = self.stream

    def __exit__(self, *exc):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        _WGRAD["stream"] = None
        _WGRAD["keep"].clear()
        return False


def _channel_sums(t, c, c_off):
    """fp32 [c] = sum over batch and voxels of channels [c_off, c_off + c) of a channels-last tensor: a bias gradient.  One
    streaming pass of the InstanceNorm statistics kernel (fp64 accumulators) instead of torch's reduction over a strided
    fp16 view (128 us for the 96^3 x 64 x 2 gradient of the last transposed convolution; the kernel streams it in ~50)."""
    from . import ops
    st = ops.stats_buffer(t.shape[0], c, t.device)
    ops.instnorm_stats(t, c, st, c_off=c_off)
    return ops.stats_channel_sums(st, c)            # decode + sum over the samples in one launch (twelve torch launches before)


class _Conv3dK3(torch.autograd.Function):
    """y = conv3d(x, w, b), 3x3x3 / pad 1, channels-last.  forward: dua_conv3d_k3_fwd; backward: the same kernel on
    dy with the weights flipped and transposed (data gradient) + dua_conv3d_k3_wgrad (weight gradient)."""

    @staticmethod
    def _run(x, w, bias, cout):
        from . import ops
        N, D, H, W, cs = x.shape
        wp, bp = ops.pack_conv3_weights(w, bias, x.dtype, cin_packed=cs, pad_bias=False)
        y = torch.empty((N, D, H, W, cout), dtype=x.dtype, device=x.device)
        stats = ops.stats_buffer(N, cout, x.device)
        ops.conv3d_k3(x, cs, 0, wp, bp, cout, y, 0, stats, workspace=ops.splitk_ws(x.dtype, N, D, H, W, cs, cout, x.device))
        return y

    @staticmethod
    def _dgrad(dy, w, cin_padded, packs=None):
        """dx [.., cin_padded] = data gradient: the forward kernel on dy with W' packed straight from w (or taken from the
        evaluation's batch-packed weights, ops.ConvPacks)."""
        from . import ops
        N, D, H, W, cs = dy.shape
        wp = packs.get(w, "dgrad", cs) if packs is not None else None
        if wp is not None:
            bp = ops.zero_bias(w.shape[1], w.device)
        else:
            wp, bp = ops.pack_conv3_weights_dgrad(w, dy.dtype, cout_packed=cs)
        dx = torch.empty((N, D, H, W, cin_padded), dtype=dy.dtype, device=dy.device)
        ops.conv3d_k3(dy, cs, 0, wp, bp, cin_padded, dx, 0, ops.stats_buffer(N, cin_padded, dy.device),
                      workspace=ops.splitk_ws(dy.dtype, N, D, H, W, cs, cin_padded, dy.device))
        return dx

    @staticmethod
    def forward(ctx, x, weight, bias):
        assert x.is_contiguous() and x.shape[-1] % 8 == 0 and weight.shape[0] % 8 == 0 and weight.shape[1] <= x.shape[-1]
        ctx.save_for_backward(x, weight)
        return _Conv3dK3._run(x, weight.detach().float().contiguous(), bias.detach().float(), weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        cout, cin = weight.shape[:2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _Conv3dK3._dgrad(dy, weight.detach().float().contiguous(), x.shape[-1])
        if ctx.needs_input_grad[1]:
            dw = ops.zeros((cout, cin, 3, 3, 3), torch.float32, x.device)
            _wgrad(x, dy, cout, dw)
            dw = dw.to(weight.dtype)
        if ctx.needs_input_grad[2]:
            db = _channel_sums(dy, cout, 0)
        return dx, dw, db


def _cl_pad(parts, dtype):
    """torch.cat(parts, 1) (NCDHW tensors) -> channels-last with the channel count padded to a multiple of 8 (zeros).  Inputs that
    carry no gradient (the training step's image and x_t) go through dua_to_channels_last, one pass per part; the cat / fill /
    strided-copy route of torch stays for anything else."""
    parts = list(parts) if isinstance(parts, (list, tuple)) else [parts]
    N = parts[0].shape[0]
    C = sum(p.shape[1] for p in parts)
    cp = -(-C // 8) * 8
    if all(p.is_cuda and p.dtype == torch.float32 and not p.requires_grad for p in parts):
        from . import ops
        out = torch.empty((N, *parts[0].shape[2:], cp), dtype=dtype, device=parts[0].device)
        if len(parts) <= 2 and cp * out.element_size() <= 64 and (cp * out.element_size()) % 16 == 0:
            return ops.to_channels_last_rows([p.contiguous() for p in parts], out)        # whole rows, 16-byte stores, one launch
        off = 0
        for i, p in enumerate(parts):
            last = i == len(parts) - 1
            ops.to_channels_last(p.contiguous(), out, off, c_fill=(cp - off) if last else None)
            off += p.shape[1]
        return out
    x = torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]
    out = x.new_zeros((N, *x.shape[2:], cp), dtype=dtype)
    out[..., :C] = x.permute(0, 2, 3, 4, 1)
    return out


def _slice_of(t):
    """(buffer, channel offset) with t == buffer[..., off:off+C], without a copy when t is a channel slice of a contiguous
    channels-last buffer (the gradient of a concat half): the backward kernels take (stride, offset) operands."""
    if t.is_contiguous():
        return t, 0
    N, D, H, W, Cc = t.shape
    st = t.stride()
    ct = st[3]
    if st == (D * H * W * ct, H * W * ct, W * ct, ct, 1) and ct % 8 == 0:
        off = t.storage_offset() % ct
        if off % 8 == 0 and off + Cc <= ct:
            return t.as_strided((N, D, H, W, ct), st, t.storage_offset() - off), off
    return t.contiguous(), 0


class _UpCat(torch.autograd.Function):
    """cat([skip, ConvTranspose3d_k2s2(lo)], channel axis) (UpCat.forward, denoiser.py:184-191): the transposed convolution
    writes its half of the concat buffer in place; backward reads its half of the buffer's gradient in place (data gradient
    + weight gradient kernels) and hands the skip half on as a view."""

    @staticmethod
    def forward(ctx, lo, skip, weight, bias):
        from . import ops
        N, D, H, W, cs = skip.shape
        cin, cout = weight.shape[:2]
        assert lo.is_contiguous() and lo.shape[-1] == cin and cin % 8 == 0 and cout % 8 == 0 and cs % 8 == 0
        buf, off = _slice_of(skip)
        if buf is not skip and off == 0 and buf.shape[-1] == cs + cout:
            cat = buf                # the skip was materialised straight into its concat buffer (_ConvNormAct, cat_extra): no copy
        else:
            cat = torch.empty((N, D, H, W, cs + cout), dtype=skip.dtype, device=skip.device)
            cat[..., :cs].copy_(skip)
        w32 = weight.detach().float().contiguous()
        wp, bp = ops.pack_deconv_weights(w32, bias.detach().float(), skip.dtype, alias_bias=True)
        ops.deconv_k2s2(lo, cin, 0, wp, bp, cout, cat, cs)
        ctx.save_for_backward(lo, w32)
        ctx.cs = cs
        return cat

    @staticmethod
    def backward(ctx, dcat):
        from . import ops
        lo, w32 = ctx.saved_tensors
        dcat = dcat.contiguous()
        cs, (cin, cout) = ctx.cs, w32.shape[:2]
        dx, dw = ops.deconv_k2s2_bwd(lo, cin, 0, dcat, cout, cs, w32, need_dx=ctx.needs_input_grad[0],
                                     need_dw=ctx.needs_input_grad[2])
        db = _channel_sums(dcat, cout, cs) if ctx.needs_input_grad[3] else None
        return dx, (dcat[..., :cs] if ctx.needs_input_grad[1] else None), dw, db


class _BwdLink:
    """Hand-over between the two blocks of a TwoConv in backward: block b's data-gradient launch produces block a's dA and can take
    the reduce pass of a's InstanceNorm backward along (ops.conv3d_k3_dgrad_reduce).  a.forward deposits what that needs; b.backward
    (which runs first) leaves the sums and the address of the dA they belong to; a.backward uses them if that is the dA it gets."""

    def __init__(self):
        self.raw = self.stats = self.g32 = self.b32 = self.sums = None
        self.count = 0
        self.da_ptr = None


class _ConvNormAct(torch.autograd.Function):
    """a = LeakyReLU(InstanceNorm(conv3d(x, w, b))) [+ add[n, c]] [+ emb] -- one MONAI Convolution block (+ the temb bias /
    the encoder embedding that follow it in TwoConv.forward / BasicUNetRDenoiser.forward), all on the HIP kernels:
    forward  = conv (raw output + statistics in its epilogue) -> materialize;
    backward = norm/activation backward (reduce + apply) -> data gradient (conv kernel) + weight gradient kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, add, emb, pool=False, cat_extra=0, temb_state=None, temb_index=0, packs=None,
                fold=None, link_out=None, link_in=None):
        """``temb_state``: ``add`` is the block-major add buffer of the whole evaluation (_TembAdds) and this block uses rows
        ``temb_index`` of it (_TembState); otherwise ``add`` is this block's own [N, cout] tensor or None.
        ``cat_extra`` > 0: the activation is the skip of a decoder level -- it is written into channels [0, cout) of a buffer
        with ``cat_extra`` more channels (the half the transposed convolution fills later, _UpCat) and returned as that view:
        torch.cat((x_e, upsampled)) (denoiser.py:190) costs no copy (226 MB moved per step at level 0 otherwise).
        ``fold`` = (lo, deconv weight, deconv bias, skip channels): ``x`` is the concat buffer _UpCat filled from ``lo``; the
        FORWARD convolution then runs as the folded launch on (skip half of x, lo) with composed weights (dua_upconv_k3_fwd, 58
        instead of 196 GFLOP per sample for the upsampled half at level 0); backward is unchanged and reads x as before."""
        from . import ops
        N, D, H, W, cs = x.shape
        cout = weight.shape[0]
        assert x.is_contiguous() and cs % 8 == 0 and cout % 8 == 0 and weight.shape[1] <= cs
        wp = packs.get(weight.detach(), "fwd", cs) if packs is not None else None          # batch-packed at the start of the evaluation
        if wp is not None:
            bp = bias.detach().float().contiguous()
        else:
            wp, bp = ops.pack_conv3_weights(weight.detach().float().contiguous(), bias.detach().float(), x.dtype, cin_packed=cs,
                                            pad_bias=False)
        ctx.packs = packs
        raw = torch.empty((N, D, H, W, cout), dtype=x.dtype, device=x.device)
        stats = ops.stats_buffer(N, cout, x.device)
        if fold is not None:
            lo, wd, bd, cskip = fold
            w_skip, wu, btab = ops.pack_upconv_weights(weight.detach().float().contiguous(), bias.detach().float(),
                                                       wd.detach().float().contiguous(), bd.detach().float(), cskip, x.dtype)
            ops.upconv_k3(x, cskip, 0, lo, lo.shape[-1], 0, None, w_skip, wu, btab, cout, raw, 0, stats)
        else:
            ops.conv3d_k3(x, cs, 0, wp, bp, cout, raw, 0, stats, workspace=ops.splitk_ws(x.dtype, N, D, H, W, cs, cout, x.device))
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        if add is not None and temb_state is not None:
            a32 = temb_state.rows(add.detach(), temb_index)
        else:
            a32 = add.detach().float().contiguous() if add is not None else None
        norm = ops.Norm(stats, g32, b32, D * H * W, add=a32, add_stride=cout)
        if cat_extra:
            cat = torch.empty((N, D, H, W, cout + cat_extra), dtype=x.dtype, device=x.device)
            act = cat[..., :cout]
        else:
            cat = act = torch.empty_like(raw)
        pooled = torch.empty((N, D // 2, H // 2, W // 2, cout), dtype=x.dtype, device=x.device) if pool else None
        ops.materialize(raw, cout, norm, cat, 0, emb=emb.detach() if emb is not None else None, pooled=pooled)
        ctx.save_for_backward(x, weight, raw, stats, g32, b32, act if pool else None)
        ctx.link_out, ctx.link_in = link_out, link_in           # _BwdLink: this block's output feeds link_out's consumer / its input is link_in's
        if link_out is not None:
            link_out.raw, link_out.stats, link_out.g32, link_out.b32, link_out.count = raw, stats, g32, b32, D * H * W
        ctx.has_add, ctx.has_emb, ctx.pool = add is not None, emb is not None, pool
        ctx.temb_state, ctx.temb_index = (temb_state, temb_index) if add is not None else (None, 0)
        return (act, pooled) if pool else act

    @staticmethod
    def backward(ctx, dA, dP=None):
        from . import ops
        x, weight, raw, stats, g32, b32, act = ctx.saved_tensors
        N, D, H, W, cout = raw.shape
        buf, off = _slice_of(dA) if dA is not None else (None, 0)       # a concat half's gradient is read in place
        if ctx.pool and dP is not None:       # MaxPool3d(2) backward + the skip-path gradient in one pass
            abuf, aoff = _slice_of(act)         # the activation may live in its decoder's concat buffer (cat_extra)
            dA = buf = ops.maxpool2_bwd_add(abuf, aoff, cout, buf, off, dP.contiguous())
            off = 0
        norm = ops.Norm(stats, g32, b32, D * H * W)
        sums = None
        lo_ = ctx.link_out
        if lo_ is not None and lo_.sums is not None and not (ctx.pool and dP is not None) and off == 0 and buf.data_ptr() == lo_.da_ptr:
            sums = lo_.sums                 # the launch that produced dA already took this layer's reduce pass
        dY = torch.empty_like(raw)
        ts = ctx.temb_state
        dadd_out = None
        if ts is not None:
            if ts.dadd is None:           # zero-initialised: a block whose backward never runs (its output unused) contributes nothing
                ts.dadd = ops.zeros((ts.N * ts.P,), torch.float32, raw.device)
            dadd_out = ts.rows(ts.dadd, ctx.temb_index)
        dgamma, dbeta, dadd = ops.instnorm_bwd(buf, off, raw, cout, norm, dY, want_add=ctx.has_add, dadd_out=dadd_out, sums=sums)
        if ts is not None:
            ts.written += 1
            # the evaluation's first block runs last in backward and hands the shared buffer on (see _TembState)
            dadd = ts.dadd if ctx.temb_index == 0 else None
        dx = dw = None
        if ctx.needs_input_grad[0]:
            li = ctx.link_in
            if (li is not None and li.raw is not None and ops.TRAIN_DGRAD_REDUCE and x.shape[-1] == li.raw.shape[-1]
                    and ops.conv3d_k3_dgrad_reduce_supported(dY.dtype, N, D, H, W, cout, x.shape[-1])):
                # x is the activation of the block in front: dx is its dA, and this launch takes its norm-backward sums along
                wp = ctx.packs.get(weight.detach(), "dgrad", cout) if ctx.packs is not None else None
                if wp is None:
                    wp, _ = ops.pack_conv3_weights_dgrad(weight.detach().float().contiguous(), dY.dtype, cout_packed=cout)
                dx = torch.empty((N, D, H, W, x.shape[-1]), dtype=dY.dtype, device=dY.device)
                pnorm = ops.Norm(li.stats, li.g32, li.b32, li.count)
                li.sums = ops.instnorm_bwd_sums(li.raw, pnorm)
                ops.conv3d_k3_dgrad_reduce(dY, cout, wp, ops.zero_bias(x.shape[-1], x.device), x.shape[-1], dx, li.raw, pnorm, li.sums)
                li.da_ptr = dx.data_ptr()
            else:
                dx = _Conv3dK3._dgrad(dY, weight.detach().float().contiguous(), x.shape[-1], ctx.packs)
        if ctx.needs_input_grad[1]:
            dw = ops.zeros(tuple(weight.shape), torch.float32, x.device)
            _wgrad(x, dY, cout, dw)
        db = ops.zeros((cout,), torch.float32, x.device)      # bias before InstanceNorm: sum(dY) == 0 exactly
        return dx, dw, db, dgamma, dbeta, dadd, (dA if ctx.has_emb else None), None, None, None, None, None, None, None, None


def _two_conv_cl(block, x, temb, emb=None, pool=False, cat_extra=0, packs=None, fold=None):
    """``temb``: None (the encoder's blocks) or (add, state): the evaluation's block-major add buffer (_TembAdds) and its
    _TembState; the blocks take their rows in call order."""
    add = state = None
    index = 0
    if temb is not None:
        add, state = temb
        index = state.next
        state.next += 1
        assert block.temb_proj.weight.shape[0] == state.couts[index]
    c0, c1 = block.conv_0, block.conv_1
    link = _BwdLink() if torch.is_grad_enabled() else None       # conv_1's data gradient is conv_0's dA (see _BwdLink)
    h = _ConvNormAct.apply(x, c0.conv.weight, c0.conv.bias, c0.adn.N.weight, c0.adn.N.bias, add, None, False, 0, state, index, packs,
                           fold, link, None)
    return _ConvNormAct.apply(h, c1.conv.weight, c1.conv.bias, c1.adn.N.weight, c1.adn.N.bias, None, emb, pool, cat_extra, None, 0,
                              packs, None, None, link)


class _Head(torch.autograd.Function):
    """final_conv (1x1x1): dua_head_fwd / dua_head_bwd (du, dW, db in one pass over the activation)."""

    @staticmethod
    def forward(ctx, u, weight, bias):
        from . import ops
        w2 = weight.detach().float().reshape(weight.shape[0], -1).contiguous()
        ctx.save_for_backward(u, w2)
        ctx.wshape = weight.shape
        return ops.head_fwd(u, w2, bias.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dlogits):
        from . import ops
        u, w2 = ctx.saved_tensors
        du, dW, db = ops.head_bwd(dlogits.contiguous(), u, w2)
        return du, dW.view(ctx.wshape), db


class _SegLoss(torch.autograd.Function):
    """losses/loss.py:25-86 for any subset of mse / bce / dice under "sum" / "mean" / "log" on channels-last logits: one
    reduce pass forward, one gradient pass backward."""

    @staticmethod
    def forward(ctx, logits, labels, names=("mse", "bce", "dice"), combine="sum"):
        from . import ops
        L, sums, dcomb = ops.seg_loss_reduce(logits, labels, names, combine)
        ctx.save_for_backward(logits, labels, sums, dcomb)
        ctx.names = names
        return L

    @staticmethod
    def backward(ctx, g):
        from . import ops
        logits, labels, sums, dcomb = ctx.saved_tensors
        return ops.seg_loss_grad(logits, labels, sums, g * dcomb, ctx.names), None, None, None


def native_conv_denoise(net, image, x, step, dtype=torch.float16):
    """Diffusion.denoise (diffusion.py:71-84) for training: HIP convolutions under torch autograd.  Returns fp32
    logits [N, C, D, H, W]."""
    return native_logits_cl(net, image, x, step, dtype).permute(0, 4, 1, 2, 3).float()


def native_logits_cl(net, image, x, step, dtype=torch.float16):
    """Same network, logits left channels-last [N, D, H, W, C] in the compute dtype (what the fused loss consumes)."""
    if image.device.type != "cuda":
        raise RuntimeError("DiffUNet runs on an MI355X (device 'cuda'); there is no CPU path in this package")
    from . import _native
    _native.lib()            # fails loudly when libdua_hip.so is missing
    enc, den = net.embed_model, net.model
    from . import ops
    # the nine denoiser blocks in the order they run: their temb_proj rows come out of one launch chain (_TembAdds)
    blocks = [den.conv_0, den.down_1.convs, den.down_2.convs, den.down_3.convs, den.down_4.convs, den.upcat_4.convs,
              den.upcat_3.convs, den.upcat_2.convs, den.upcat_1.convs]
    # every layer's fp16 weights, forward and data-gradient layout, packed by one launch per 64 tensors; an ordinary layer's input
    # buffer has exactly its Cin channels and its output gradient exactly its Cout (the first layers -- one or seventeen input
    # channels in a padded buffer -- and fp32 plans pack per layer as before)
    packs = ops.ConvPacks(dtype)
    for blk in [enc.conv_0] + [d.convs for d in enc.down] + blocks:
        for conv in (blk.conv_0.conv, blk.conv_1.conv):
            w = conv.weight.detach()
            packs.add(w, "fwd", w.shape[1])
            if torch.is_grad_enabled():
                packs.add(w, "dgrad", w.shape[0])
    packs.run()
    img = _cl_pad(image, dtype)
    e, pe = _two_conv_cl(enc.conv_0, img, None, None, True, packs=packs)
    emb = [e]
    for i, d in enumerate(enc.down):                 # the pooled copy of each level comes out of its materialize pass
        last = i == len(enc.down) - 1
        r = _two_conv_cl(d.convs, pe, None, None, not last, packs=packs)
        e, pe = (r, None) if last else r
        emb.append(e)
    state = _TembState(image.shape[0], [b.temb_proj.weight.shape[0] for b in blocks])
    tparams = [den.temb.dense[0].weight, den.temb.dense[0].bias, den.temb.dense[1].weight, den.temb.dense[1].bias]
    for b in blocks:
        tparams += [b.temb_proj.weight, b.temb_proj.bias]
    temb = (_TembAdds.apply(step.to(torch.int64), den.temb.embedding_dim // 2, state, *tparams), state)
    h = _cl_pad([image, x], dtype)
    up_c = [blk.upsample.deconv.weight.shape[1] for blk in (den.upcat_1, den.upcat_2, den.upcat_3, den.upcat_4)]   # channels each
    x0, p0 = _two_conv_cl(den.conv_0, h, temb, emb[0], True, up_c[0], packs)                                     # decoder adds to the skip
    x1, p1 = _two_conv_cl(den.down_1.convs, p0, temb, emb[1], True, up_c[1], packs)
    x2, p2 = _two_conv_cl(den.down_2.convs, p1, temb, emb[2], True, up_c[2], packs)
    x3, p3 = _two_conv_cl(den.down_3.convs, p2, temb, emb[3], True, up_c[3], packs)
    x4 = _two_conv_cl(den.down_4.convs, p3, temb, emb[4], packs=packs)

    def up(block, lo, skip):
        dc = block.upsample.deconv
        cat = _UpCat.apply(lo, skip, dc.weight, dc.bias)
        fold = None
        N, D, H, W, cs = skip.shape
        cout = block.convs.conv_0.conv.weight.shape[0]
        # the forward convolution over the concat as ONE folded launch where the level has rounds of tiles to fill (level 0)
        if (ops.TRAIN_FOLD_UPCONV and dtype == torch.float16 and N * (D // 8) * (H // 8) * (W // 8) >= ops.TRAIN_FOLD_MIN_TILES
                and D % 8 == 0 and H % 8 == 0 and W % 8 == 0 and lo.is_contiguous()
                and ops.upconv_supported(dtype, N, D, H, W, cs, cat.shape[-1], lo.shape[-1], lo.shape[-1], cout, cout)):
            fold = (lo.detach(), dc.weight, dc.bias, cs)
        return _two_conv_cl(block.convs, cat, temb, packs=packs, fold=fold)

    u4 = up(den.upcat_4, x4, x3)
    u3 = up(den.upcat_3, u4, x2)
    u2 = up(den.upcat_2, u3, x1)
    u1 = up(den.upcat_1, u2, x0)
    wf = den.final_conv.weight
    if wf.shape[0] <= ops.HEAD_MAX_K and u1.shape[-1] <= ops.HEAD_MAX_C:
        return _Head.apply(u1, wf, den.final_conv.bias)
    # more classes / channels than the head kernel holds in registers: plain library GEMM
    return u1 @ wf.reshape(wf.shape[0], -1).t().to(u1.dtype) + den.final_conv.bias.to(u1.dtype)


class _NativeModule(torch.nn.Module):
    """nn.Module face of native_logits_cl, so torch's DistributedDataParallel reducer can hook the parameters."""

    def __init__(self, net, dtype):
        super().__init__()
        self.net, self.dtype = net, dtype

    def forward(self, images, x_t, t):
        return native_logits_cl(self.net, images, x_t, t, self.dtype)


class NativeAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW (train.py:121-122: lr, weight_decay; betas (0.9, 0.999), eps 1e-8) on the library's multi-tensor kernel:
    one launch per 64 tensors instead of torch's five-per-step ``multi_tensor_apply`` launches plus a separate unscale pass;
    1 / grad_scale is applied while the gradient is read, a set ``found_inf`` skips the update on the device, and the update
    counter lives on the device (one int32 for all tensors), so a captured step replays it.  It is a torch ``Optimizer``:
    param_groups drive the reference's LR scheduler (schedule.py), and ``state_dict()`` has torch.optim.AdamW's layout
    (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), so checkpoints move between the two (engine.py:118-135)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._count = None          # int32 device scalar: updates applied so far
        self.lr_dev = None          # fp32 device scalar that overrides the groups' lr when set (captured steps: the host value
                                    # would be baked into the graph)

    def _counter(self, device):
        if self._count is None:
            self._count = torch.zeros((), dtype=torch.int32, device=device)
        return self._count

    def _moments(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = torch.zeros((), dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st["exp_avg"], st["exp_avg_sq"]

    def materialize_state(self):
        """Allocate the moments of every parameter now (before a stream capture, which must not allocate persistent state)."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad:
                    self._moments(p)
                    self._counter(p.device)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None, found_inf=None, store_grad=False, advance=True):
        """``grad_scale`` / ``found_inf``: fp32 device scalars of a loss-scaled step (see ops.adamw_step); ``advance=False`` leaves
        the update counter to the caller's ops.adamw_advance (which also applies the loss-scale rule)."""
        from . import ops
        assert closure is None
        count = None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            mv = [self._moments(p) for p in ps]
            count = self._counter(ps[0].device)
            ops.adamw_step([p.data for p in ps], [p.grad for p in ps], [m for m, _ in mv], [v for _, v in mv], count, group["lr"],
                           group["betas"], group["eps"], group["weight_decay"], lr_dev=self.lr_dev, grad_scale=grad_scale,
                           found_inf=found_inf, store_grad=store_grad)
        if advance and count is not None:
            ops.adamw_advance(count)

    def state_dict(self):
        if self._count is not None:
            k = float(self._count.item())
            for st in self.state.values():
                st["step"] = torch.tensor(k, dtype=torch.float32)
        return super().state_dict()

    def snapshot_state(self):
        """Copies of the moments and the update counter (``restore_state`` writes them back INTO the live tensors)."""
        return ({p: (st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"].clone()) for p, st in self.state.items()
                 if "exp_avg" in st}, None if self._count is None else self._count.clone())

    def restore_state(self, snap):
        moments, count = snap
        for p, st in self.state.items():
            if p in moments:
                st["exp_avg"].copy_(moments[p][0]); st["exp_avg_sq"].copy_(moments[p][1]); st["step"] = moments[p][2].clone()
            elif "exp_avg" in st:
                st["exp_avg"].zero_(); st["exp_avg_sq"].zero_(); st["step"].zero_()
        if self._count is not None:
            if count is None:
                self._count.zero_()
            else:
                self._count.copy_(count)

    def load_state_dict(self, state_dict):
        # Moments that already exist keep their ADDRESSES: a captured step (NativeConvTrainer(graph=True)) updates those tensors
        # on every replay, and the base class would rebind state[p] to fresh ones -- the graph would go on updating the old
        # moments while state_dict() saved tensors that never move.  Loaded values are copied into the live tensors instead.
        live = {p: (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.state.items() if "exp_avg" in st}
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for p, (m, v) in live.items():
                st = self.state.get(p)
                if st is None or "exp_avg" not in st:
                    m.zero_(); v.zero_()
                    self.state[p].update(step=torch.zeros((), dtype=torch.float32), exp_avg=m, exp_avg_sq=v)
                    continue
                if st["exp_avg"] is not m:
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                    st["exp_avg"], st["exp_avg_sq"] = m, v
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        if steps:
            dev = next(iter(self.state)).device
            self._counter(dev).fill_(int(steps[0]))


class NativeConvTrainer:
    """One-process-per-GPU trainer on the native-convolution path: fp32 master weights, fp16 activations and gradients
    with dynamic loss scaling (or plain fp32), AdamW as train.py:121-126 (``NativeAdamW``; ``fused_optimizer=False`` keeps
    torch.optim.AdamW itself for eager steps -- the reference's exact optimizer object, used as the comparison in the tests).
    Under torch.distributed the gradients are
    averaged by torch's DDP reducer (``overlap=True``: 32 MB buckets all-reduced while the rest of backward still
    runs; RCCL on a GPU node) or by one flat all-reduce after backward (``overlap=False``).
    ``graph=True`` replays the step from HIP graphs: one graph single-process; under torch.distributed TWO graphs
    (forward + backward | unscale + AdamW + loss-scale update) with the flat gradient all-reduce between them -- the
    collective stays an ordinary call, so any backend works (RCCL on a node, gloo in the tests)."""

    def __init__(self, net, lr=2e-4, weight_decay=1e-4, losses="mse,bce,dice", loss_combine="sum",
                 dtype=torch.float16, init_scale=2.0 ** 12, overlap=True, graph=False, fused_optimizer=True,
                 wgrad_overlap=True):
        import torch.distributed as dist
        self.net, self.dtype = net, dtype
        self.lr, self.weight_decay, self.init_scale = lr, weight_decay, init_scale
        self.use_graph, self._graph, self._graph2 = graph, None, None
        self._qtab = None
        self._amp = None
        self.module = _NativeModule(net, dtype)
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.overlap = overlap and self.distributed and not graph        # the DDP reducer's hooks cannot be captured
        if self.overlap:
            from torch.nn.parallel import DistributedDataParallel
            dev = next(net.parameters()).device
            self.module = DistributedDataParallel(self.module, device_ids=[dev.index], bucket_cap_mb=32)
        self.loss_names, self.loss_combine = parse_losses(losses, loss_combine)
        self.params = [p for p in net.parameters() if p.requires_grad]
        # graph mode always uses the library's AdamW; ``fused_optimizer=False`` selects torch.optim.AdamW for eager steps
        self.native_opt = bool(fused_optimizer or graph)
        self.optimizer = (NativeAdamW(self.params, lr=lr, weight_decay=weight_decay) if self.native_opt
                          else torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay))
        self.scale, self.good_steps = (init_scale if dtype == torch.float16 else 1.0), 0
        # every zero-initialised buffer of a step (weight-gradient accumulators, statistics rows, reduction scratch) comes
        # out of one arena that a single fill re-zeroes: gradients are 4 bytes per parameter, the rest is small
        from . import ops
        dev = self.params[0].device
        if dev.type == "cuda":
            from . import _native
            _native.prepare(dev)         # every kernel's function attributes: before the first launch, before any capture, and
                                         # before autograd's worker thread (which issues the backward launches) exists
        self.arena = ops.ZeroArena(sum(p.numel() for p in self.params) * 4 + (96 << 20), dev) if dev.type == "cuda" else None
        # weight gradients on a side stream (see _wgrad); not under the DDP reducer, whose hooks read them on the main stream --
        # and NOT in graph mode: a captured fork / join makes the step a multi-branch HIP graph, and hipGraphLaunch of such a
        # graph can fault inside the ROCm 7 runtime (hip::Graph::UpdateStreams reads past the end of the exec's parallel-stream
        # list whenever more than one of those streams shares the launch stream's queue: which ones do depends on how many
        # streams the process created before -- the host segfaults of test_graph_replayed_training_step_equals_eager in full
        # test-suite runs).  The captured step is therefore a single-stream graph; the side stream was worth
        # 0.03 ms of a 17.7 ms step there (17.67 vs 17.70, same-process A/B of round 2).
        self.wgrad_stream = (torch.cuda.Stream(device=dev)
                             if (dev.type == "cuda" and not self.overlap and wgrad_overlap and not graph) else None)
        # graph mode captures on a stream of this trainer's own, after warming up on that same stream (so that the per-stream
        # scratch buffers of ops exist before the capture and do not land in the graph's private memory pool)
        self.capture_stream = torch.cuda.Stream(device=dev) if (graph and dev.type == "cuda") else None

    def _allreduce(self):
        if not self.distributed or self.overlap:          # the DDP reducer already averaged them during backward
            return
        allreduce_mean_([p.grad for p in self.params])

    # ---- whole-step HIP graph (single process): the eager step issues ~3,000 launches and is host-bound once the kernels
    # take < 22 ms; one replay per step removes the host from the loop.  Everything a step decides stays on the device:
    # the timestep -> (sqrt(a_bar), sqrt(1 - a_bar)) gather, the fp16 overflow check (the flag feeds NativeAdamW,
    # which skips the update itself) and the loss-scale update (dua_adamw_advance: torch._amp_update_scale_'s rule).
    def _graph_fwd_bwd(self):
        from . import ops
        g = self._g
        self.arena.reset()
        with self.arena:
            x_t = ops.q_sample_affine(g["labels"], 2.0, -1.0, g["noise"], g["qtab"], g["t"])      # q_sample(label * 2 - 1, t, noise)
            self.optimizer.zero_grad(set_to_none=True)
            with torch.enable_grad():
                loss = _SegLoss.apply(self.module(g["images"], x_t, g["t"]), g["labels"], self.loss_names, self.loss_combine)
                with _wgrad_side(self.wgrad_stream):
                    (loss * g["scale"]).backward()
        return loss.detach()

    def _graph_update(self):
        g = self._g
        from . import ops
        # found_inf is zero on entry (zero-initialised, cleared again by adamw_advance at the end of every step)
        # (g["found_inf"] is what the host reads: the flag the last completed step saw)
        ops.grads_nonfinite([p.grad for p in self.params], g["found_flag"])
        self.optimizer.step(grad_scale=g["scale"], found_inf=g["found_flag"], advance=False)
        ops.adamw_advance(self.optimizer._count, g["found_flag"], g["scale"], g["growth"], 2.0, 0.5, 200, seen=g["found_inf"])

    def _graph_body(self):
        loss = self._graph_fwd_bwd()
        if self.distributed:
            allreduce_mean_([p.grad for p in self.params])
        self._graph_update()
        return loss

    def _build_graph(self, images, labels):
        assert labels.dtype == torch.float32
        dev = images.device
        d = self.net.diffusion
        self.optimizer.materialize_state()
        self.optimizer.lr_dev = torch.full((), float(self.optimizer.param_groups[0]["lr"]), dtype=torch.float32, device=dev)
        self._lr_host = float(self.optimizer.param_groups[0]["lr"])
        self._g = dict(images=images.clone(), labels=labels.contiguous().clone(), noise=torch.zeros_like(labels),
                       t=torch.zeros(labels.shape[0], dtype=torch.long, device=dev),
                       qtab=torch.stack([torch.as_tensor(d.sqrt_alphas_cumprod), torch.as_tensor(d.sqrt_one_minus_alphas_cumprod)],
                                        dim=1).float().to(dev).contiguous(),
                       scale=torch.full((), self.init_scale if self.dtype == torch.float16 else 1.0, device=dev),
                       growth=torch.zeros((), dtype=torch.int32, device=dev), found_inf=torch.zeros((), device=dev),
                       found_flag=torch.zeros((), device=dev))
        saved = [p.detach().clone() for p in self.params]
        # ... and the optimizer state as it is NOW: a checkpoint loaded before the first step (load_checkpoint ->
        # optimizer.load_state_dict) must survive the warm-up updates below
        saved_opt = self.optimizer.snapshot_state()
        # Warm-up ON the capture stream: the per-(device, stream) scratch buffers of ops (split-K, weight-gradient partials)
        # are then allocated here, from the ordinary pool, and the capture finds them -- allocated inside the capture they
        # would live in this graph's private pool while the module-level cache hands them to every later capture.
        side = self.capture_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):                       # warm-up: workspaces, optimizer state tensors
                self._g["noise"].normal_()
                self._graph_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(dev)
        self.optimizer.zero_grad(set_to_none=True)
        if not self.distributed:
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, stream=self.capture_stream):
                self._g["loss"] = self._graph_body()
        else:
            # two graphs sharing one memory pool; the gradients allocated while capturing the first stay alive (p.grad)
            # and are what the eager all-reduce and the second graph see on every replay
            self._graph, self._graph2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, stream=self.capture_stream):
                self._g["loss"] = self._graph_fwd_bwd()
            with torch.cuda.graph(self._graph2, pool=self._graph.pool(), stream=self.capture_stream):
                self._graph_update()
        # the warm-up steps were real updates: put weights, Adam moments/step counter and the loss scale back
        with torch.no_grad():
            for p, q in zip(self.params, saved):
                p.copy_(q)
            self.optimizer.restore_state(saved_opt)
            self._g["found_inf"].zero_(); self._g["found_flag"].zero_()
            self._g["scale"].fill_(self.init_scale if self.dtype == torch.float16 else 1.0)
            self._g["growth"].zero_()

    def _graph_step(self, images, labels, noise, t):
        if self._graph is None:
            self._build_graph(images, labels)
        g = self._g
        g["images"].copy_(images); g["labels"].copy_(labels)
        if t is None:
            t, _ = self.net.sampler.sample(labels.shape[0], labels.device)
        g["t"].copy_(t)
        if noise is None:
            g["noise"].normal_()
        else:
            g["noise"].copy_(noise)
        lr = float(self.optimizer.param_groups[0]["lr"])          # an LR scheduler's new value reaches the captured kernels here
        if lr != self._lr_host:
            self.optimizer.lr_dev.fill_(lr)
            self._lr_host = lr
        self._graph.replay()
        if self._graph2 is not None:
            allreduce_mean_([p.grad for p in self.params])
            self._graph2.replay()
        return g["loss"]

    def _amp_state(self):
        """Device-resident loss-scaling state of eager fp16 steps on the library's AdamW: scale, growth counter, this step's
        overflow flag and the flag the last completed step saw (``loss_scale()`` / ``last_step_overflowed()`` read them back)."""
        if self._amp is None:
            dev = self.params[0].device
            self._amp = dict(scale=torch.full((), float(self.init_scale), device=dev), growth=torch.zeros((), dtype=torch.int32, device=dev),
                             flag=torch.zeros((), device=dev), seen=torch.zeros((), device=dev))
        return self._amp

    def loss_scale(self):
        """The current loss scale (a device read-back in the modes that keep it on the device)."""
        if self.use_graph and self._graph is not None:
            return float(self._g["scale"])
        return float(self._amp["scale"]) if self._amp is not None else float(self.scale)

    def last_step_overflowed(self):
        if self.use_graph and self._graph is not None:
            return bool(self._g["found_inf"].item())
        return bool(self._amp["seen"].item()) if self._amp is not None else False

    def step(self, images, labels, noise=None, t=None):
        if self.use_graph:
            return self._graph_step(images, labels, noise, t)
        if t is None:
            t, _ = self.net.sampler.sample(labels.shape[0], labels.device)
        noise = torch.randn_like(labels) if noise is None else noise
        if labels.is_cuda and labels.dtype == torch.float32:
            from . import ops
            if self._qtab is None:
                d = self.net.diffusion
                self._qtab = torch.stack([torch.as_tensor(d.sqrt_alphas_cumprod), torch.as_tensor(d.sqrt_one_minus_alphas_cumprod)],
                                         dim=1).float().to(labels.device).contiguous()
            # x_start = label * 2 - 1 and q_sample (train.py:258-262) in one pass
            x_t = ops.q_sample_affine(labels.contiguous(), 2.0, -1.0, noise.float().contiguous(), self._qtab,
                                      t.to(device=labels.device, dtype=torch.int64).contiguous())
        else:
            x_t = self.net.diffusion.q_sample(labels * 2 - 1, t, noise)
        self.optimizer.zero_grad(set_to_none=True)
        import contextlib
        if self.arena is not None:
            self.arena.reset()
        with (self.arena if self.arena is not None else contextlib.nullcontext()), torch.enable_grad():
            loss = _SegLoss.apply(self.module(images, x_t, t), labels.float().contiguous(), self.loss_names,
                                  self.loss_combine)
            amp = self._amp_state() if (self.dtype == torch.float16 and self.native_opt) else None
            with _wgrad_side(self.wgrad_stream):
                (loss * (amp["scale"] if amp is not None else self.scale)).backward()
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        self._allreduce()
        if amp is not None:
            # dynamic loss scaling without a host decision (the eager step used to read found_inf back -- one synchronisation per
            # step, which also kept the host from running ahead of the device under the DDP reducer): overflow flag, skipped update,
            # scale and growth counter live on the device exactly as in the captured step
            from . import ops
            ops.grads_nonfinite([p.grad for p in self.params], amp["flag"])
            self.optimizer.step(grad_scale=amp["scale"], found_inf=amp["flag"], store_grad=True, advance=False)
            ops.adamw_advance(self.optimizer._count, amp["flag"], amp["scale"], amp["growth"], 2.0, 0.5, 200, seen=amp["seen"])
            return loss.detach()
        if self.dtype == torch.float16:
            dev = self.params[0].device
            found_inf = torch.zeros(1, dtype=torch.float32, device=dev)
            grads = [p.grad for p in self.params]
            inv = torch.full((1,), 1.0 / self.scale, dtype=torch.float32, device=dev)
            torch._amp_foreach_non_finite_check_and_unscale_(grads, found_inf, inv)      # torch.optim.AdamW path (fused_optimizer=False)
            if bool(found_inf.item()):                                           # overflow: skip, halve the scale
                self.scale, self.good_steps = max(self.scale / 2, 2.0 ** -14), 0
                return loss.detach()
            self.good_steps += 1
            if self.good_steps >= 200:
                self.scale, self.good_steps = self.scale * 2, 0
        self.optimizer.step()
        return loss.detach()
