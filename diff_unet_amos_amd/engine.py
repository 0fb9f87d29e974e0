"""Launch plan of the two Diff-UNet networks on one MI355X.

Owns, per (batch, patch shape, dtype): packed weights, channels-last workspaces sized for HBM
residency (nothing is freed between steps; a 96^3 x 16-class patch keeps ~3 GB resident), the
timestep-embedding table, and the fixed kernel sequence of one encoder pass and one denoiser
evaluation.  The denoiser evaluation + sampler update touches the host only to enqueue; the
sampling loops capture it once into a HIP graph and replay it per step.

Call sites in the reference that this replaces: BasicUNetEncoder.forward
(models/basic_unet/pretrained/basic_unet.py:496-512), BasicUNetRDenoiser.forward
(models/basic_unet/denoiser.py:284-312), and the loop bodies of
GaussianDiffusion.p_sample_loop_progressive / ddim_sample_loop_progressive
(guided_diffusion/gaussian_diffusion.py:487-535, 667-716).
"""
from __future__ import annotations

import math

import torch

from . import _native as nv
from . import ops

SLOPE = 0.1          # LeakyReLU(0.1): models/diff_unet.py:34-35, pretrained/basic_unet.py:429
EPS = 1e-5           # nn.InstanceNorm3d default


class _Conv:
    """One Convolution block (Conv3d + InstanceNorm affine) bound to its buffers."""


class EmbeddingList(list):
    """What embed_model(image) returns: indexable like the reference's list of 5 NCDHW tensors
    (converted on first access), while the denoiser uses the channels-last device buffers."""

    def __init__(self, plan, token):
        super().__init__([None] * 5)
        self.plan, self.token = plan, token

    def __getitem__(self, i):
        v = super().__getitem__(i)
        if v is None:
            p = self.plan
            assert p.emb_token == self.token, "embeddings were overwritten by a later encoder pass"
            v = ops.from_channels_last(p.emb[i], p.f[i])
            super().__setitem__(i, v)
        return v

    def __iter__(self):
        return (self[i] for i in range(5))


class Plan:
    """Buffers + launch sequences for one (N, D, H, W, dtype)."""

    def __init__(self, net, N, D, H, W, dtype, device):
        f = tuple(net.features)
        assert len(f) == 6 and all(c % 8 == 0 for c in f), "feature sizes must be multiples of 8"
        assert D % 16 == 0 and H % 16 == 0 and W % 16 == 0, \
            "patch extents must be multiples of 16 (four 2x poolings; the reference recommends the same)"
        assert D >= 32 and H >= 32 and W >= 32, "InstanceNorm3d needs >1 voxel at the bottom level (SURVEY F7)"
        self.net, self.N, self.dims, self.dtype, self.dev, self.f = net, N, (D, H, W), dtype, device, f
        nv.prepare(device)            # function attributes of every kernel: before the first launch and before any capture
        self.C = net.num_classes
        self.cx = ops.state_stride(self.C)
        self.cin0 = -(-(self.C + 1) // 8) * 8           # [x_t (C) | image | zero pad]
        self.up = (f[1], f[2] // 2, f[3] // 2, f[4] // 2)   # deconv output channels landing at level 0..3
        S = [(D >> l, H >> l, W >> l) for l in range(5)]
        self.S = S
        z = lambda l, c, dt=dtype: torch.zeros((N, *S[l], c), dtype=dt, device=device)  # noqa: E731
        # encoder
        # The encoder's first TwoConv (two 96^3 convolutions, ONCE per patch) runs on the exact-fp32 MFMA path also in an
        # fp16 plan: embeddings[0] is added to x_0 in every one of the T steps, so its rounding error is the same in every
        # step and does not average out of the sum of predictions -- emulated on the oracle (tools/precision_sites.py) it was
        # 45 % of the 50-step Dice deviation of an all-fp16 pass, for ~2 ms per patch.
        self.enc_hi = dtype == torch.float16 and bool(getattr(net, "encoder_level0_fp32", True))
        # The LAST reverse steps of a DDPM loop (p_sample_loop, BASELINE config 2) run on a companion exact-fp32 plan: as t -> 0
        # the posterior mean turns into the clamped prediction itself (coef1 -> 1, coef2 -> 0), so the final sample carries the
        # rounding error of the last evaluations undamped.  Measured (tools/final_steps_precision.py, 1000 steps, 32^3): worst-
        # class 1 - Dice of the thresholded sample against the oracle 9.5e-4 all-fp16, 6.8e-4 / 4.7e-4 / 4.7e-4 / 4.1e-4 with the
        # last 1 / 2 / 5 / 10 steps in fp32 -- two steps cost 19 ms of a 1.5 s loop at 96^3 x 16.
        self.finish_fp32_steps = int(getattr(net, "ddpm_finish_fp32_steps", 2))
        if self.enc_hi:
            f32 = torch.float32
            self.img_in32 = z(0, 8, f32)
            self.rawA32, self.rawB32, self.emb32, self.pool32 = z(0, f[0], f32), z(0, f[0], f32), z(0, f[0], f32), z(1, f[0], f32)
        self.img_in = z(0, 8)
        self.rawA = [z(l, f[l]) for l in range(5)]
        self.rawB = [z(l, f[l]) for l in range(5)]
        self.emb = [z(l, f[l]) for l in range(5)]
        self.pool = [z(l + 1, f[l]) for l in range(4)]
        self.emb_token = 0
        # denoiser
        self.xin = z(0, self.cin0)
        self.cat = [z(l, f[l] + self.up[l]) for l in range(4)]
        self.x4 = z(4, f[4])
        dec_out = (f[5], f[1], f[2], f[3])
        self.uA = [z(l, dec_out[l]) for l in range(4)]
        self.uB = [z(l, dec_out[l]) for l in range(4)]
        self.dec_out = dec_out
        # InstanceNorm sums of every conv layer live in ONE arena of fixed-point words, zeroed by a single memset per pass
        self._stat_slices = []
        # sampler state
        self.x_state = torch.zeros((N, *S[0], self.cx), dtype=torch.float32, device=device)
        self.x_sum = torch.zeros((N, *S[0], self.cx), dtype=torch.float32, device=device)
        self.cur_coef = torch.zeros((N, 8), dtype=torch.float32, device=device)
        self.counter = torch.zeros(1, dtype=torch.int32, device=device)
        self.step_word = torch.zeros(1, dtype=torch.int32, device=device)
        self.err_word = torch.zeros(1, dtype=torch.int32, device=device)     # set by step_begin on an out-of-range index
        self.seed_word = torch.zeros(1, dtype=torch.int64, device=device)    # Philox key of the current sampling call
        self.calls = 0
        self.logits = torch.zeros((N, self.C, *S[0]), dtype=torch.float32, device=device)
        self._bind()
        self._alloc_stats()
        self.weights_version = None
        self.graphs = {}
        self.tables = {}

    # ---- parameter binding -------------------------------------------------------------------
    def _mk(self, name, block, cin_packed=None, perm=None, tap=None):
        c = _Conv()
        c.name = name
        # single-channel tap form of the first layers (fp16 only; ops.conv3d_k3): packed index of the lone image channel
        c.tap = tap if (tap in (0, 16) and self.dtype == torch.float16 and cin_packed == (tap or 0) + 8) else None
        c.w, c.b = block.conv.weight, block.conv.bias
        c.gamma, c.beta = block.adn.N.weight, block.adn.N.bias
        c.cout, c.cin = c.w.shape[0], c.w.shape[1]
        c.cin_packed, c.perm = cin_packed, perm
        c.wp = c.bp = None
        c.stats = c.norm = c.norm_add = None
        c.dt = self.dtype                 # operand type of this layer (enc_hi: fp32 for the encoder's first block)
        return c

    def _alloc_stats(self):
        convs = [c for pair in self.enc + self.den + self.dec for c in pair]
        sizes = [self.N * ops.STAT_REPLICAS * ops.STAT_WORDS * (-(-c.cout // 64) * 64) for c in convs]
        self.stat_arena = torch.zeros(sum(sizes), dtype=torch.int64, device=self.dev)
        o = 0
        for c, n in zip(convs, sizes):
            c.stats = self.stat_arena[o:o + n].view(self.N, ops.STAT_REPLICAS, ops.STAT_WORDS, -1)
            o += n
        n_enc = sum(sizes[:10])
        self.enc_stats, self.den_stats = self.stat_arena[:n_enc], self.stat_arena[n_enc:]
        # split-K scratch for the layers that cannot fill the chip on their own (<= 24^3)
        need = 0
        for pairs, levels in ((self.enc, range(5)), (self.den, range(5)), (self.dec, range(4))):
            for l, pair in zip(levels, pairs):
                for c in pair:
                    cin = c.cin_packed or c.cin
                    need = max(need, ops.conv3_workspace_bytes(c.dt, self.N, *self.S[l], -(-cin // 8) * 8, c.cout))
        self.splitk_ws = torch.empty(max(need, 16) // 4, dtype=torch.float32, device=self.dev)

    def _bind(self):
        net = self.net
        enc, den = net.embed_model, net.model
        self.enc = [(self._mk("e0a", enc.conv_0.conv_0, cin_packed=8, perm=[0] + [-1] * 7, tap=0), self._mk("e0b", enc.conv_0.conv_1))]
        if self.enc_hi:
            for c in self.enc[0]:
                c.dt, c.tap = torch.float32, None
        for i in range(4):
            tc = enc.down[i].convs
            self.enc.append((self._mk(f"e{i+1}a", tc.conv_0), self._mk(f"e{i+1}b", tc.conv_1)))
        C = self.C
        perm0 = list(range(1, C + 1)) + [0] + [-1] * (self.cin0 - C - 1)   # packed [x_t | image | pad] -> source [image | x_t]
        self.den = [(self._mk("d0a", den.conv_0.conv_0, cin_packed=self.cin0, perm=perm0, tap=C), self._mk("d0b", den.conv_0.conv_1))]
        for i, blk in enumerate((den.down_1, den.down_2, den.down_3, den.down_4)):
            self.den.append((self._mk(f"d{i+1}a", blk.convs.conv_0), self._mk(f"d{i+1}b", blk.convs.conv_1)))
        ups = (den.upcat_1, den.upcat_2, den.upcat_3, den.upcat_4)          # index = level the block outputs at
        self.dec = [(self._mk(f"u{l}a", ups[l].convs.conv_0), self._mk(f"u{l}b", ups[l].convs.conv_1)) for l in range(4)]
        self.deconv = [ups[l].upsample.deconv for l in range(4)]
        self.deconv_packed = [None] * 4
        self.upconv_packed = [None] * 4
        # timestep-embedding blocks in table order, with their offsets
        blocks = [den.conv_0, den.down_1.convs, den.down_2.convs, den.down_3.convs, den.down_4.convs,
                  den.upcat_4.convs, den.upcat_3.convs, den.upcat_2.convs, den.upcat_1.convs]
        self.temb_blocks = blocks
        offs, o = [], 0
        for b in blocks:
            offs.append(o)
            o += b.temb_proj.weight.shape[0]
        self.P = o
        self.temb_off = {"d0": offs[0], "d1": offs[1], "d2": offs[2], "d3": offs[3], "d4": offs[4],
                         "u3": offs[5], "u2": offs[6], "u1": offs[7], "u0": offs[8]}
        self.cur_add = torch.zeros((self.N, self.P), dtype=torch.float32, device=self.dev)
        self.temb_table = None

    def _params(self):
        return [p for p in self.net.parameters()]

    def refresh_weights(self):
        """Re-pack when any parameter changed (optimizer step, load_state_dict)."""
        ver = tuple((p.data_ptr(), p._version) for p in self._params())
        if ver == self.weights_version:
            return
        dt = self.dtype
        with torch.no_grad():
            for pair in self.enc + self.den + self.dec:
                for c in pair:
                    c.wp, c.bp = ops.pack_conv3_weights(c.w.detach().float().contiguous(), c.b.detach(), c.dt,
                                                        cin_packed=c.cin_packed, perm=c.perm, tap_channel=c.tap)
                    c.gamma_c = c.gamma.detach().float().contiguous()
                    c.beta_c = c.beta.detach().float().contiguous()
                    c.norm = c.norm_add = None       # rebuilt lazily against the new gamma/beta
            for l in range(4):
                d = self.deconv[l]
                if self._fold_level(l):
                    # UpCat's first convolution with the transposed convolution folded in (csrc/upconv.hip): composed weights
                    a = self.dec[l][0]
                    self.upconv_packed[l] = ops.pack_upconv_weights(a.w.detach().float().contiguous(), a.b.detach(),
                                                                    d.weight.detach().float().contiguous(), d.bias.detach(), self.f[l])
                    self.deconv_packed[l] = None
                else:
                    self.upconv_packed[l] = None
                    self.deconv_packed[l] = ops.pack_deconv_weights(d.weight.detach().float().contiguous(), d.bias.detach(), dt)
            den = self.net.model
            self.wf = den.final_conv.weight.detach().float().reshape(self.C, -1).contiguous()
            self.bf = den.final_conv.bias.detach().float().contiguous()
            # embedding table for every original timestep (depends on weights only)
            T = self.net.timesteps
            half = den.temb.embedding_dim // 2
            freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).to(self.dev)
            wcat = torch.cat([b.temb_proj.weight.detach().float() for b in self.temb_blocks], 0).contiguous()
            bcat = torch.cat([b.temb_proj.bias.detach().float() for b in self.temb_blocks], 0).contiguous()
            d0, d1 = den.temb.dense[0], den.temb.dense[1]
            ts = torch.arange(T, dtype=torch.int32, device=self.dev)
            self.temb_table = ops.temb_table(ts, freqs, d0.weight.detach().float().contiguous(),
                                             d0.bias.detach().float().contiguous(),
                                             d1.weight.detach().float().contiguous(),
                                             d1.bias.detach().float().contiguous(), wcat, bcat)
        self.weights_version = ver
        self.graphs.clear()

    # ---- building blocks ------------------------------------------------------------------------
    def _norm(self, c, level, add_key=None):
        """dua_in_norm of conv block ``c`` (the producer), optionally with a timestep-embedding bias slice."""
        count = self.S[level][0] * self.S[level][1] * self.S[level][2]
        if add_key is None:
            if c.norm is None:
                c.norm = ops.Norm(c.stats, c.gamma_c, c.beta_c, count, slope=SLOPE, eps=EPS)
            return c.norm
        if c.norm_add is None:
            c.norm_add = ops.Norm(c.stats, c.gamma_c, c.beta_c, count, add=self.cur_add.view(-1)[self.temb_off[add_key]:],
                                  add_stride=self.P, slope=SLOPE, eps=EPS)
        return c.norm_add

    def _conv(self, c, x, cin, y, level, xform_from=None, add_key=None, in_blocked=False, out_blocked=False):
        norm = None if xform_from is None else self._norm(xform_from, level, add_key)
        ops.conv3d_k3(x, cin, 0, c.wp, c.bp, c.cout, y, 0, c.stats, norm=norm, workspace=self.splitk_ws, tap_channel=c.tap,
                      in_blocked=in_blocked, out_blocked=out_blocked)

    # Decoder levels whose transposed convolution is folded into the convolution behind it (dua_upconv_k3_fwd: one launch for
    # upsample + cat + conv_0 of an UpCat block, 3.4x fewer multiply-adds on the upsampled half): fp16 plans, levels with at
    # least this many 8x8x8 output tiles (x batch x output-channel tiles): the launch must fill the chip.  Same-box A/B of the
    # whole 1000-step loop (profiles/r5_fold_threshold_ab.txt): no fold 1.469 ms/step, level 0 only 1.325, levels 0 and 1 (216 tiles
    # at 48^3) 1.316; the 24^3 level (27 tiles) stays on the split forms of the plain convolution.
    # ``net.fold_upconv = False`` keeps the two-launch form everywhere, ``net.upconv_min_tiles`` overrides the bound (A/B, tests).
    UPCONV_MIN_TILES = 200

    def _fold_level(self, l):
        if getattr(self, "_fold", None) is None:
            self._fold = [False] * 4
            if self.dtype == torch.float16 and bool(getattr(self.net, "fold_upconv", True)):
                for k in range(3):                       # level 3's source is the materialised bottom level (no producer descriptor)
                    D, H, W = self.S[k]
                    cat_c = self.f[k] + self.up[k]
                    if D % 8 or H % 8 or W % 8:
                        continue
                    tiles = (D // 8) * (H // 8) * (W // 8) * self.N * (-(-self.dec_out[k] // 64))
                    self._fold[k] = (tiles >= int(getattr(self.net, "upconv_min_tiles", self.UPCONV_MIN_TILES)) and
                                     ops.upconv_supported(self.dtype, self.N, D, H, W, self.f[k], cat_c, self.dec_out[k + 1],
                                                          self.dec_out[k + 1], self.dec_out[k], self.dec_out[k]))
        return self._fold[l]

    def _level0_layout(self):
        """Which level-0 buffers of the DENOISER are kept in 16-channel blocks (dua_conv3_desc.layout): those whose only
        consumer is the wide-tile convolution (it walks its input 16 channels at a time) and whose producer can write
        blocks -- asked of the launchers' own policy functions.  rawA[0]: first layer -> conv_0.conv_1; cat[0]: materialise +
        transposed convolution -> upcat_1.conv_0; uA[0]: upcat_1.conv_0 -> upcat_1.conv_1."""
        if getattr(self, "_l0", None) is None:
            N, f, dt = self.N, self.f, self.dtype
            D, H, W = self.S[0]
            a0, b0 = self.den[0]
            u0a, u0b = self.dec[0]
            kind = lambda cin, cs, cout, fused, tap=None: ops.conv3_kernel_kind(dt, N, D, H, W, cin, cs, cout, fused, tap)   # noqa: E731
            wide = ops.KIND_WIDE
            raw_a = (dt == torch.float16 and f[0] % 16 == 0 and kind(self.cin0, self.cin0, a0.cout, False, a0.tap) == ops.KIND_FIRST
                     and kind(a0.cout, f[0], b0.cout, True) == wide)
            if self._fold_level(0):
                # the folded up-convolution reads the skip half of cat[0] in blocks (written by materialise) and writes blocks
                cat = f[0] % 16 == 0 and (f[0] + self.up[0]) % 16 == 0
                u_a = self.dec_out[0] % 16 == 0 and kind(u0a.cout, self.dec_out[0], u0b.cout, True) == wide
            else:
                cat = (dt == torch.float16 and f[0] % 16 == 0 and (f[0] + self.up[0]) % 16 == 0
                       and kind(f[0] + self.up[0], f[0] + self.up[0], u0a.cout, False) == wide
                       and ops.deconv_kernel_kind(dt, N, *self.S[1], self.dec_out[1], self.up[0]) == ops.DECONV_ALLTAPS)
                u_a = (dt == torch.float16 and self.dec_out[0] % 16 == 0 and kind(f[0] + self.up[0], f[0] + self.up[0], u0a.cout, False) == wide
                       and kind(u0a.cout, self.dec_out[0], u0b.cout, True) == wide)
            self._l0 = (raw_a, cat, u_a)
        return self._l0

    def run_encoder(self, image):
        """BasicUNetEncoder.forward: fills self.emb[0..4] (channels-last)."""
        self.refresh_weights()
        N = self.N
        assert tuple(image.shape) == (N, 1, *self.dims), f"image shape {tuple(image.shape)} != plan {(N, 1, *self.dims)}"
        self._note_condition(image, None)
        img = image.detach().float().contiguous()
        ops.to_channels_last(img, self.img_in, 0, 8)
        ops.to_channels_last(img, self.xin, self.C, self.cin0 - self.C)     # conditioning channel of the denoiser input
        self.enc_stats.zero_()
        x, cin = self.img_in, 8
        for l in range(5):
            a, b = self.enc[l]
            if l == 0 and self.enc_hi:          # exact-fp32 operands for the first block, fp16 copies for its consumers
                ops.to_channels_last(img, self.img_in32, 0, 8)
                self._conv(a, self.img_in32, 8, self.rawA32, 0)
                self._conv(b, self.rawA32, a.cout, self.rawB32, 0, xform_from=a)
                ops.materialize(self.rawB32, b.cout, self._norm(b, 0), self.emb32, 0, pooled=self.pool32)
                self.emb[0].copy_(self.emb32)
                self.pool[0].copy_(self.pool32)
                x, cin = self.pool[0], b.cout
                continue
            self._conv(a, x, cin, self.rawA[l], l)
            self._conv(b, self.rawA[l], a.cout, self.rawB[l], l, xform_from=a)
            ops.materialize(self.rawB[l], b.cout, self._norm(b, l), self.emb[l], 0,
                            pooled=self.pool[l] if l < 4 else None)
            if l < 4:
                x, cin = self.pool[l], b.cout
        self.emb_token += 1
        return EmbeddingList(self, self.emb_token)

    def _hi_plan(self):
        """Companion exact-fp32 plan of an fp16 plan (same network, batch, patch), with the conditioning of the patch this plan
        holds: its own encoder pass on the fp32 image, or the caller's embeddings."""
        hi = getattr(self, "_hi", None)
        if hi is None:
            hi = self._hi = Plan(self.net, self.N, *self.dims, torch.float32, self.dev)
        hi.refresh_weights()
        image, emb = self._cond
        if getattr(hi, "_cond_token", None) != self._cond_token:
            if isinstance(emb, EmbeddingList) or emb is None:
                hi.run_encoder(image)
            else:
                hi.stage_condition(image, emb)
            hi._cond_token = self._cond_token
        return hi

    def _note_condition(self, image, embeddings):
        self._cond = (image.detach(), embeddings)
        self._cond_token = getattr(self, "_cond_token", 0) + 1

    def stage_condition(self, image, embeddings):
        """Make sure the denoiser's conditioning (image channel of xin, 5 embedding maps) is resident."""
        self._note_condition(image, embeddings)
        if isinstance(embeddings, EmbeddingList) and embeddings.plan is self and embeddings.token == self.emb_token:
            return          # run_encoder just staged both from this image
        img = image.detach().float().contiguous()
        assert tuple(img.shape) == (self.N, 1, *self.dims)
        ops.to_channels_last(img, self.xin, self.C, self.cin0 - self.C)
        self.load_embeddings(embeddings)

    def load_embeddings(self, embeddings):
        """Accept caller-supplied NCDHW embeddings (a plain list) instead of this plan's own."""
        if isinstance(embeddings, EmbeddingList) and embeddings.plan is self and embeddings.token == self.emb_token:
            return
        for l in range(5):
            e = embeddings[l]
            assert tuple(e.shape) == (self.N, self.f[l], *self.S[l])
            ops.to_channels_last(e.detach().float().contiguous(), self.emb[l], 0, self.f[l])
        self.emb_token += 1

    def denoiser_body(self, zero_stats=True):
        """BasicUNetRDenoiser.forward from the staged input (self.xin) up to the raw output of the
        last decoder block; self.cur_add must hold the embedding rows of this evaluation.  This method is the ONE
        place the launch sequence is written down: called directly it launches kernel by kernel (instrumented
        passes); under ops.recording it fills the op list dua_denoiser_step executes (native_step)."""
        f = self.f
        if zero_stats:
            self.den_stats.zero_()
        blk_raw, blk_cat, blk_u = self._level0_layout()
        x, cin = self.xin, self.cin0
        for l in range(5):
            a, b = self.den[l]
            self._conv(a, x, cin, self.rawA[l], l, out_blocked=blk_raw and l == 0)
            self._conv(b, self.rawA[l], a.cout, self.rawB[l], l, xform_from=a, add_key=f"d{l}", in_blocked=blk_raw and l == 0)
            if l < 4:
                ops.materialize(self.rawB[l], b.cout, self._norm(b, l), self.cat[l], 0, emb=self.emb[l],
                                pooled=self.pool[l], out_blocked=blk_cat and l == 0)
                x, cin = self.pool[l], b.cout
            else:
                ops.materialize(self.rawB[4], b.cout, self._norm(b, 4), self.x4, 0, emb=self.emb[4])
        src, src_c, src_conv = self.x4, f[4], None
        for l in (3, 2, 1, 0):
            norm = self._norm(src_conv, l + 1) if src_conv is not None else None
            a, b = self.dec[l]
            if self._fold_level(l):
                # upsample + cat + conv_0 in one launch: the skip half of cat[l] on the fine grid, the coarse raw tensor behind it
                w_skip, wu, btab = self.upconv_packed[l]
                ops.upconv_k3(self.cat[l], f[l], 0, src, src_c, 0, norm, w_skip, wu, btab, a.cout, self.uA[l], 0, a.stats,
                              in_blocked=blk_cat and l == 0, out_blocked=blk_u and l == 0)
            else:
                wp, bp = self.deconv_packed[l]
                ops.deconv_k2s2(src, src_c, 0, wp, bp, self.up[l], self.cat[l], f[l], norm=norm, out_blocked=blk_cat and l == 0)
                self._conv(a, self.cat[l], f[l] + self.up[l], self.uA[l], l, in_blocked=blk_cat and l == 0, out_blocked=blk_u and l == 0)
            self._conv(b, self.uA[l], a.cout, self.uB[l], l, xform_from=a, add_key=f"u{l}", in_blocked=blk_u and l == 0)
            src, src_c, src_conv = self.uB[l], b.cout, b
        return self.dec[0][1]

    def tail(self, mode, noise=None, logits=None, xstart=None, use_sum=False):
        """The Philox key of in-kernel noise is read from self.seed_word at run time (see new_seed)."""
        last = self.dec[0][1]
        ops.final_conv_sampler(self.uB[0], last.cout, self._norm(last, 0), self.wf, self.bf, self.C, mode,
                               coef=self.cur_coef, x_state=self.x_state, noise=noise, step_word=self.step_word,
                               xin=self.xin if mode != nv.MODE_LOGITS else None,
                               xstart_sum=self.x_sum if use_sum else None, logits=logits, xstart=xstart,
                               seed_dev=self.seed_word)

    def new_seed(self, seed=None):
        """Key of this call's in-kernel noise.  The reference draws a fresh th.randn_like every step of every call
        (gaussian_diffusion.py:430,576); the counter-based generator needs a fresh KEY per call for the same effect:
        one draw from torch's CPU generator (so torch.manual_seed governs it), mixed with the rank so that replicas
        do not share a noise field.  The key lives in a device word, not in the captured graph."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    seed ^= (dist.get_rank() + 1) * 0x9E3779B97F4A7C15 & (2 ** 62 - 1)
            except Exception:       # pragma: no cover - torch.distributed not built
                pass
        self.calls += 1
        self.seed_word.fill_(int(seed) & (2 ** 63 - 1))
        return seed

    # ---- the evaluation as one C-ABI call -------------------------------------------------------------------
    def native_step(self, mode, rows_per_sample=None, row_of_step=None, coef_table=None, noise=None, logits=None,
                    xstart=None, use_sum=False):
        """step begin + denoiser + tail through dua_denoiser_step (include/dua_hip.h).  The op list is recorded from
        denoiser_body once per weight packing; per call only the step's rows / noise / outputs are filled in."""
        if getattr(self, "_step_ops_version", None) != self.weights_version:
            rec = []
            with ops.recording(rec):
                last = self.denoiser_body(zero_stats=False)
            self._step_ops = (nv.StepOp * len(rec))(*rec)
            self._step_last = last
            self._step_ops_version = self.weights_version
        last = self._step_last
        tnorm = self._norm(last, 0)
        tn = tnorm.c
        N, vox = self.N, self.dims[0] * self.dims[1] * self.dims[2]
        ptr = lambda t: t.data_ptr() if t is not None else None          # noqa: E731
        sampling = mode != nv.MODE_LOGITS
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous() and noise.numel() == N * self.C * vox
        for t in (logits, xstart):
            if t is not None:
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == N * self.C * vox
        nsteps = 0
        if rows_per_sample is not None:
            assert rows_per_sample.dtype == torch.int32 and rows_per_sample.numel() == N and rows_per_sample.is_cuda
        else:
            assert row_of_step is not None and row_of_step.dtype == torch.int32 and coef_table is not None
            nsteps = row_of_step.numel()
            assert coef_table.numel() >= 8 * nsteps
        tail = nv.TailDesc(nv.dt_code(self.dtype), N, vox, last.cout, self.uB[0].shape[-1], self.C, self.cx, mode,
                           self.xin.shape[-1] if sampling else 0, 0, self.seed_word.data_ptr())
        p = nv.DenoiserPlan(
            N, self.P, self.temb_table.data_ptr(), self.temb_table.shape[0], ptr(rows_per_sample),
            ptr(row_of_step), nsteps, ptr(coef_table), self.counter.data_ptr(),
            self.cur_add.data_ptr(), self.cur_coef.data_ptr(), self.step_word.data_ptr(), self.err_word.data_ptr(),
            self.den_stats.data_ptr(), self.den_stats.numel() * 8,
            self._step_ops, len(self._step_ops), self.splitk_ws.data_ptr(), self.splitk_ws.numel() * 4,
            tail, self.uB[0].data_ptr(), nv.InNorm(tn.stats, tn.gamma, tn.beta, tn.add, tn.add_stride, tn.c_pad, tn.count, tn.eps, tn.slope),
            self.wf.data_ptr(), self.bf.data_ptr(), self.x_state.data_ptr(), ptr(noise),
            self.xin.data_ptr() if sampling else None, self.x_sum.data_ptr() if (use_sum and sampling) else None,
            ptr(logits), ptr(xstart))
        self._step_keep = (p, noise, logits, xstart, rows_per_sample, row_of_step, coef_table)
        ops.denoiser_step(p)

    # ---- public operations ------------------------------------------------------------------------
    def denoise(self, x, t):
        """logits = model(x, t, image, embeddings) for already-encoded image (denoiser.py:284-312)."""
        self.refresh_weights()
        N = self.N
        assert tuple(x.shape) == (N, self.C, *self.dims) and t.numel() == N
        ops.to_channels_last(x.detach().float().contiguous(), self.xin, 0, self.C)
        T = self.temb_table.shape[0]
        on_host = not t.is_cuda
        if on_host and not bool(((t >= 0) & (t < T)).all()):
            raise ValueError(f"timestep out of range: the model was built for 0 <= t < {T}, got {t.tolist()}")
        rows = t.detach().to(device=self.dev, dtype=torch.int32).contiguous()
        if not on_host:
            self.err_word.zero_()
        out = torch.empty((N, self.C, *self.dims), dtype=torch.float32, device=self.dev)
        self.native_step(nv.MODE_LOGITS, rows_per_sample=rows, logits=out)
        if not on_host and int(self.err_word.item()):     # device-resident t: the kernel clamped it, say so
            raise ValueError(f"timestep out of range: the model was built for 0 <= t < {T}")
        return out

    def sample_loop(self, diffusion, kind, noise=None, step_noise=None, eta=0.0, use_graph=True, seed=None,
                    want_final_xstart=False, snapshots=None, want_sum=None):
        """T reverse steps (T = diffusion.num_timesteps) starting from ``noise`` (x_T, NCDHW) or a fresh
        draw.  ``step_noise``: optional list of per-step NCDHW draws (parity runs); otherwise the
        tail kernel generates eps in-kernel (Philox, keyed per call: ``seed`` or a draw from torch's generator).
        ``snapshots``: optional dict {step count k: None}; after k steps the state x is stored there (NCDHW copy;
        eager mode) -- drift-versus-step measurements.  ``want_sum``: accumulate the sum of the per-step x0 predictions (what
        models/diffusion/diffusion.py:94-98 sums from ``all_samples``); default: DDIM loops only -- the reference's p_sample_loop
        (gaussian_diffusion.py:441-485) returns the final sample alone, and the sum is 113 MB of HBM traffic per step at 96^3 x 16.
        Returns dict(sample, sum_pred_xstart (None without the sum))."""
        want_sum = (kind == "ddim") if want_sum is None else bool(want_sum)
        self.refresh_weights()
        N, T = self.N, diffusion.num_timesteps
        shape = (N, self.C, *self.dims)
        if noise is None:
            noise = torch.randn(*shape, device=self.dev)
        assert tuple(noise.shape) == shape
        x_T = noise.detach().float().contiguous()
        ops.to_channels_last(x_T, self.x_state, 0, self.cx)
        ops.to_channels_last(x_T, self.xin, 0, self.C)
        self.x_sum.zero_()
        mode = nv.MODE_DDPM if kind == "ddpm" else nv.MODE_DDIM
        tkey = (diffusion, kind, float(eta))          # the object itself: the table keeps it alive, no id() reuse after GC
        gkey = tkey + (want_sum,)
        if tkey not in self.tables:
            order = list(range(T))[::-1]
            tt = torch.tensor(order)
            coef = diffusion.ddpm_coef(tt) if kind == "ddpm" else diffusion.ddim_coef(tt, eta)
            tmap = diffusion.model_timesteps()
            self.tables[tkey] = (coef.to(self.dev).contiguous(),
                                 torch.tensor([tmap[i] for i in order], dtype=torch.int32, device=self.dev))
        coef_table, row_of_step = self.tables[tkey]
        self.counter.zero_()
        self.new_seed(seed)
        if step_noise is not None:
            assert len(step_noise) == T
            use_graph = False
        if snapshots:
            use_graph = False

        def one_step(eps):
            self.native_step(mode, row_of_step=row_of_step, coef_table=coef_table, noise=eps, use_sum=want_sum)

        # DDPM in an fp16 plan: the LAST steps run on the exact-fp32 path (see finish_fp32_steps in __init__)
        finish = min(T, self.finish_fp32_steps) if (kind == "ddpm" and self.dtype == torch.float16) else 0
        lo_steps = T - finish

        def finish_hi(first_step):
            """steps [first_step, T) on the companion fp32 plan: hand the sampler state over, run, hand it back"""
            hi = self._hi_plan()
            hi.x_state.copy_(self.x_state)
            hi.xin[..., :self.C].copy_(self.x_state.view(N, *self.dims, self.cx)[..., :self.C])
            hi.x_sum.copy_(self.x_sum)
            hi.counter.copy_(self.counter)
            hi.seed_word.copy_(self.seed_word)
            for k in range(first_step, T):
                eps = None if step_noise is None else step_noise[k].detach().to(self.dev).float().contiguous()
                hi.native_step(mode, row_of_step=row_of_step, coef_table=coef_table, noise=eps, use_sum=want_sum)
                if snapshots and (k + 1) in snapshots:
                    snapshots[k + 1] = ops.from_channels_last(hi.x_state, self.C)
            self.x_state.copy_(hi.x_state)
            self.x_sum.copy_(hi.x_sum)
            self.counter.copy_(hi.counter)

        if not use_graph:
            for k in range(lo_steps):
                one_step(None if step_noise is None else step_noise[k].detach().to(self.dev).float().contiguous())
                if snapshots and (k + 1) in snapshots:
                    snapshots[k + 1] = ops.from_channels_last(self.x_state, self.C)
        else:
            key = gkey
            g = self.graphs.get(key)
            if g is None:
                # warm-up outside capture (sets kernel attributes, fills caches); state is reset below
                one_step(None)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    one_step(None)
                self.graphs[key] = g
                ops.to_channels_last(x_T, self.x_state, 0, self.cx)
                ops.to_channels_last(x_T, self.xin, 0, self.C)
                self.x_sum.zero_()
                self.counter.zero_()
            for _ in range(lo_steps):
                g.replay()
        if finish:
            finish_hi(lo_steps)
        out = {"sample": ops.from_channels_last(self.x_state, self.C),
               "sum_pred_xstart": ops.from_channels_last(self.x_sum, self.C) if want_sum else None}
        return out
