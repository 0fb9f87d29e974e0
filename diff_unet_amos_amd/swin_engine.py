"""Launch plan of the two Swin-UNETR networks (diff_swin_unetr variant, BASELINE config 5) on one MI355X.

Owns, per (batch, patch shape, dtype): packed weights, resident workspaces, the timestep table of all fifteen
``t_proj(swish(t_embedder(t)))`` projections, and the fixed kernel sequence of one encoder pass and one denoiser
evaluation.  Replaces SwinUNETREncoder.forward (models/swin_unetr/encoder.py:212-219) and SwinUNETRDenoiser.forward
(models/swin_unetr/denoiser.py:353-403).

Data layout
  * voxels / tokens are channels-last everywhere ([N, D, H, W, C]); a Swin token map IS a channels-last feature map, so
    the hidden states feed the convolutions without a transpose (the reference rearranges "b c d h w" <-> "b d h w c"
    around every stage, transformer.py:97,109).
  * the residual token stream of a stage is fp32; every GEMM / convolution operand is the compute dtype.
  * torch.cat((up, skip)) of UnetrUpBlock (blocks.py:90) is a buffer both producers write their half of.

What runs where
  * hand-written HIP: 3x3x3 convolutions + InstanceNorm statistics (conv3d_igemm.hip), the transposed convolutions
    (deconv.hip), windowed attention (window_attention.hip), norm1 + pad + roll + window partition and its inverse with
    the residual add and norm2 (swin_tokens.hip), patch embedding, patch-merging gather + LayerNorm, the stage adds,
    GELU, the UnetResBlock tail with the embedding / reverse-attention adds (swin_ops.hip), the 1x1x1 output head.
    The Linear layers (qkv / proj / MLP / patch-merging reduction) and the 1x1x1 conv3 of channel-changing UnetResBlocks run on
    the package's own MFMA GEMMs in fp16 plans (token_linear / swin_mlp at stage 0, token_gemm at stages 1-3:
    swin_gemm.hip, swin_gemm_wide.hip): no library GEMM is launched.
  * the fp32 PARITY plan (compute_dtype=torch.float32) runs the same layers on dua_linear_f32 (gemm_f32.hip: exact-fp32 MFMA
    32x32x2, plain 64 x 64 tiles, linear1's GELU in its epilogue): no library GEMM in either plan.
"""
from __future__ import annotations

import math

import torch

from . import _native as nv
from . import ops
from .swin_unetr import HEADS, WINDOW

SLOPE = 0.01         # get_act_layer(("leakyrelu", {"negative_slope": 0.01})), blocks.py:247
EPS = 1e-5


def clip_window(dims, window, shift):
    """attention.py:225-251 get_window_size."""
    ws, ss = list(window), list(shift)
    for i in range(3):
        if dims[i] <= window[i]:
            ws[i], ss[i] = dims[i], 0
    return tuple(ws), tuple(ss)


def region_ids(dims_padded, window, shift):
    """The image compute_mask (attention.py:123-160) partitions into windows: region id 0..26 per token,
    uint8 [windows, tokens]."""
    d, h, w = dims_padded
    img = torch.zeros((d, h, w), dtype=torch.uint8)
    cnt = 0
    for sd in (slice(-window[0]), slice(-window[0], -shift[0]), slice(-shift[0], None)):
        for sh in (slice(-window[1]), slice(-window[1], -shift[1]), slice(-shift[1], None)):
            for sw in (slice(-window[2]), slice(-window[2], -shift[2]), slice(-shift[2], None)):
                img[sd, sh, sw] = cnt
                cnt += 1
    wd, wh, ww = window
    x = img.view(d // wd, wd, h // wh, wh, w // ww, ww).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(-1, wd * wh * ww).contiguous()


class _Res:
    """One UnetResBlock bound to its packed weights and statistics."""


class SwinEmbeddings(list):
    """What embed_model(image) returns: [hidden_states_out (5 tensors), enc0, enc1, enc2, enc3] like the reference
    (NCDHW fp32, converted on first access) while the denoiser reads the channels-last device buffers."""

    def __init__(self, plan, token):
        super().__init__([None] * 5)
        self.plan, self.token = plan, token

    def __getitem__(self, i):
        v = super().__getitem__(i)
        if v is None:
            p = self.plan
            assert p.emb_token == self.token, "embeddings were overwritten by a later encoder pass"
            if i == 0:
                v = [ops.from_channels_last(p.e_hs[k], p.e_hs[k].shape[-1]) for k in range(5)]
            else:
                v = ops.from_channels_last(p.e_enc[i - 1], p.e_enc[i - 1].shape[-1])
            super().__setitem__(i, v)
        return v

    def __iter__(self):
        return (self[i] for i in range(5))


class _OneGraph:
    """The whole step as one HIP graph; the two-stream schedule is a fork / join captured inside it.  (Replaying the step as
    single-branch graphs per stream with host-side fork / join events was built and measured: tools/ubench_two_graphs.py shows a
    chain of 400 small launches losing 0.3 ms to a fork inside its graph and nothing to a second graph on another stream, but on
    the real step the segment boundaries cost more than that returns -- 2.61 against 2.55 ms, profiles/r2_swin_step_program_ab.txt.)"""

    def __init__(self, g):
        self.g = g

    def replay(self, times=1):
        for _ in range(times):
            self.g.replay()


class SwinPlan:
    """Buffers + launch sequences for one (N, D, H, W, dtype)."""

    def __init__(self, net, N, D, H, W, dtype, device):
        assert D % 32 == 0 and H % 32 == 0 and W % 32 == 0, \
            "input image size (image_size) should be divisible by stage-wise image resolution."      # denoiser.py:110-113
        assert (D // 32) * (H // 32) * (W // 32) > 1, "InstanceNorm3d needs more than one voxel at the 1/32 level"
        self.net, self.N, self.dims, self.dtype, self.dev = net, N, (D, H, W), dtype, device
        nv.prepare(device)            # function attributes of every kernel: before the first launch and before any capture
        self._gemm_ws, self._gemm_ws_retired = [None, None], []
        self.C = net.num_classes
        f = net.feature_size
        self.f = f
        self.cin0 = -(-(self.C + 1) // 8) * 8                 # [x_t (C) | image | zero pad]: the sampler tail writes x_{t-1}
        self.cx = ops.state_stride(self.C)                    # into channels [0, C) of the next evaluation's input
        self.perm0 = list(range(1, self.C + 1)) + [0]         # packed channel p of xin = channel perm0[p] of cat([image, x])
        S = [(D >> l, H >> l, W >> l) for l in range(6)]    # S[0] voxels, S[1..5] token maps x0..x4
        self.S = S
        self.tok_c = [f * 2 ** i for i in range(5)]           # channels of x0..x4
        z = lambda l, c, dt=dtype: torch.zeros((N, *S[l], c), dtype=dt, device=device)  # noqa: E731
        # ---- conditioning encoder: inputs, outputs
        self.img_in = z(0, 8)
        self.e_hs = [z(i + 1, self.tok_c[i]) for i in range(5)]
        self.e_enc = [z(0, f), z(1, f), z(2, 2 * f), z(3, 4 * f)]
        self.emb_token = 0
        # ---- denoiser
        self.xin = z(0, self.cin0)
        self.hs = [z(1, f), z(2, 2 * f), z(3, 4 * f), None, z(5, 16 * f)]      # hs[3] lives in cat[4]
        self.cat = [z(0, 2 * f), z(1, 2 * f), z(2, 4 * f), z(3, 8 * f), z(4, 16 * f)]   # (up | skip) of decoder1..5
        self.tail_k = -(-f // 32) * 32                      # decoder1's output keeps a 64-channel stride (48 real, the rest zero)
        # Decoder levels whose 3x3x3 convolution over torch.cat((up, skip)) takes the folded form (dua_upconv_k3_fwd, DESIGN 6: the
        # upsampled half contracted as 8 parents x the coarse channels instead of 27 taps): fp16, >= 200 tiles of 8x8x8 -- decoder1
        # (96^3) and decoder2 (48^3) at 96^3 patches.  The coarse tensor of such a level is kept at a 64-channel multiple (zero
        # padding behind its 48 / 96 channels); the transposed convolution itself still runs, for the block's 1x1x1 residual branch.
        self.dec_c = [self.tail_k, f, 2 * f, 4 * f, 8 * f, 16 * f]              # real channels of out, dec0..dec4
        self.fold_up = [False] * 5
        if dtype == torch.float16 and bool(getattr(net, "fold_upconv", True)):
            for k in range(2):
                Dk, Hk, Wk = S[k]
                cu_p = -(-self.dec_c[k + 1] // 64) * 64
                tiles = (Dk // 8) * (Hk // 8) * (Wk // 8) * N if (Dk % 8 == 0 and Hk % 8 == 0 and Wk % 8 == 0) else 0
                cout = self.cat[k].shape[-1] // 2
                self.fold_up[k] = (tiles >= int(getattr(net, "upconv_min_tiles", 200)) and cout % 16 == 0 and
                                   ops.upconv_supported(dtype, N, Dk, Hk, Wk, cout, 2 * cout, cu_p, cu_p, cout, cout))
        dec_stride = [c if not (1 <= i <= 2 and self.fold_up[i - 1]) else -(-c // 64) * 64 for i, c in enumerate(self.dec_c)]
        self.dec = [z(i if i else 0, dec_stride[i]) for i in range(6)]            # out, dec0..dec4 (levels 0, 1, 2, 3, 4, 5)
        # ... and whose 1x1x1 residual branch (conv3 over the same concat) takes its upsampled half from the coarse tensor too
        # (dua_deconv_k2s2_res_fwd: a transposed convolution with composed weights + a pointwise term on the skip half): then nobody
        # reads the upsampled tensor and the transposed convolution is not launched at all -- decoder1 (<= 128 coarse channels)
        self.res_up = [False] * 5
        if bool(getattr(net, "fold_residual", True)):
            for k in range(2):
                if self.fold_up[k]:
                    Dk, Hk, Wk = S[k + 1]
                    cout = self.cat[k].shape[-1] // 2
                    self.res_up[k] = ops.deconv_res_supported(dtype, N, Dk, Hk, Wk, self.dec_c[k + 1], dec_stride[k + 1], cout, cout, cout)
                                                            # so that the fused head + sampler tail runs its MFMA form
        # ---- shared scratch (the two networks never run concurrently)
        big = max(N * S[l][0] * S[l][1] * S[l][2] * c for l, c in ((0, f), (1, f), (2, 2 * f), (3, 4 * f), (4, 8 * f), (5, 16 * f)))
        self.raw1 = torch.zeros(big, dtype=dtype, device=device)
        self.raw2 = torch.zeros(big, dtype=dtype, device=device)
        self.res3 = torch.zeros(big, dtype=dtype, device=device)
        self.raw1b, self.raw2b, self.res3b = (torch.zeros(big, dtype=dtype, device=device) for _ in range(3))   # side stream
        self.side_stream = torch.cuda.Stream(device=device)
        self.two_streams = True
        # which encoder blocks the side stream takes once hidden_states_out[i] is enqueued (encoder_k reads hidden state k - 1;
        # encoder1, k = 0, reads the input only)
        self.side_plan = {2: (3, 2, 1, 0)}
        # the memory-bound tail of a side-stream block (conv3 + sums, residual_norm_act) also runs one workgroup per CU: at full
        # width encoder1's two 96^3 passes held the main stream's split-K finish kernels at 40-55 us instead of 6
        self.background_tails = True
        self.background_conv3 = 0           # workgroups per CU of a side-stream conv3 + sums launch (0: as many as fit)
        self.conv3_first = False            # conv3 ahead of the block's 3x3x3 convolutions on the side stream: +0.01 ms (A/B), off
        self.fused_tail = dtype == torch.float16 and self.cx == 16     # the tail assembles decoder1's output itself (MFMA tail kernel: fp16, 9..16 classes)
        self._tail_src = None
        self.background_convs = True        # side-stream 3x3x3 convolutions leave half of every CU to the main stream's chain
        self.enc_done = [torch.cuda.Event() for _ in range(4)]
        self.stream = [torch.zeros((N, *S[i + 1], self.tok_c[i]), dtype=torch.float32, device=device) for i in range(5)]
        self.geo = []
        tok_max = 0
        for i in range(4):
            dims = S[i + 1]
            ws, ss = clip_window(dims, WINDOW, tuple(w // 2 for w in WINDOW))
            pad = tuple(-(-dims[k] // ws[k]) * ws[k] for k in range(3))
            nw = (pad[0] // ws[0]) * (pad[1] // ws[1]) * (pad[2] // ws[2])
            n = ws[0] * ws[1] * ws[2]
            reg = region_ids(pad, ws, ss).to(device) if any(ss) else None
            self.geo.append(dict(dims=dims, ws=ws, ss=ss, nw=nw, n=n, region=reg,
                                 g0=ops.window_geom(N, dims, self.tok_c[i], ws, (0, 0, 0)),
                                 g1=ops.window_geom(N, dims, self.tok_c[i], ws, ss)))
            tok_max = max(tok_max, N * nw * n * self.tok_c[i])
        self.win = torch.zeros(tok_max, dtype=dtype, device=device)           # window-partitioned LN1(x)
        self.att = torch.zeros(tok_max, dtype=dtype, device=device)           # attention output
        self.ln2 = torch.zeros(tok_max, dtype=dtype, device=device)
        self.merged = torch.zeros(tok_max, dtype=dtype, device=device)        # gathered + normalised 8C tokens (= tokens * C)
        self.fused_linear = dtype == torch.float16                            # swin_gemm.hip is an fp16-operand kernel
        self.fused_mlp = True
        self.fused_reduction = False
        self.tl_qkv = self.tl_proj = self.tl_conv3 = True
        # stage 0 only (110 592 tokens x 48): at stage 1 (13 824 x 96) a launch has ~100 tiles and the library's 10 us GEMMs
        # beat the per-workgroup weight staging of the fused kernels (same-process A/B, tools/bench_swin_ab.py)
        self.fused_max_c = 48
        # Stages 1-3 and the wide 1x1x1 convolutions: the tiled MFMA GEMM of swin_gemm_wide.hip (fp16 operands) instead of the
        # ~60 hipBLASLt launches per step of round 2; the fp32 parity mode runs them on dua_linear_f32 (gemm_f32.hip).
        self.wide_gemm = dtype == torch.float16
        if self.fused_linear:
            self.qkv_buf = torch.zeros(3 * tok_max, dtype=dtype, device=device)
            self.hid_buf = torch.zeros(4 * tok_max, dtype=dtype, device=device)
            self.red_buf = torch.zeros(tok_max // 4 + 8, dtype=dtype, device=device)
            self.po_buf = torch.zeros(tok_max, dtype=dtype, device=device)
        # ---- sampler state
        self.x_state = torch.zeros((N, *S[0], self.cx), dtype=torch.float32, device=device)
        self.x_sum = torch.zeros((N, *S[0], self.cx), dtype=torch.float32, device=device)
        self.cur_coef = torch.zeros((N, 8), dtype=torch.float32, device=device)
        self.counter = torch.zeros(1, dtype=torch.int32, device=device)
        self.step_word = torch.zeros(1, dtype=torch.int32, device=device)
        self.err_word = torch.zeros(1, dtype=torch.int32, device=device)
        self.seed_word = torch.zeros(1, dtype=torch.int64, device=device)
        self._bind()
        self.weights_version = None
        self.graphs = {}
        self.tables = {}

    # ---- parameter binding -------------------------------------------------------------------
    def _res(self, name, block, level, cin_packed=None, perm=None, tap=None):
        r = _Res()
        r.name, r.block, r.level = name, block, level
        r.cout, r.cin = block.conv1.conv.weight.shape[:2]
        r.cin_packed = cin_packed or r.cin
        r.perm = perm
        # single-channel tap form of the denoiser's first convolution (fp16, 16 ordinary channels in front of the image channel:
        # ops.conv3d_k3 / the resident-weight kernel of conv3d_igemm.hip), as engine.py uses it for DiffUNet
        r.tap = tap if (tap == 16 and self.dtype == torch.float16 and r.cin_packed == tap + 8) else None
        r.has3 = hasattr(block, "conv3")
        r.t_off = None
        return r

    def _bind(self):
        enc, den = self.net.embed_model, self.net.model
        self.e_res = [self._res("e1", enc.encoder1.layer, 0, cin_packed=8), self._res("e2", enc.encoder2.layer, 1),
                      self._res("e3", enc.encoder3.layer, 2), self._res("e4", enc.encoder4.layer, 3)]
        self.d_res = [self._res("d1", den.encoder1.layer, 0, cin_packed=self.cin0, perm=self.perm0, tap=self.C),
                      self._res("d2", den.encoder2.layer, 1),
                      self._res("d3", den.encoder3.layer, 2), self._res("d4", den.encoder4.layer, 3),
                      self._res("d10", den.encoder10.layer, 5)]
        ups = [den.decoder1, den.decoder2, den.decoder3, den.decoder4, den.decoder5]
        self.u_res = [self._res(f"u{k + 1}", ups[k].conv_block, k) for k in range(5)]
        self.ups = ups
        # one arena for the InstanceNorm sums (fixed-point words, ops.stats_buffer), zeroed by one memset per pass
        sizes, blocks = [], self.e_res + self.d_res + self.u_res
        for r in blocks:
            sizes.append(self.N * ops.STAT_REPLICAS * ops.STAT_WORDS * (-(-r.cout // 64) * 64))
        self.stat_arena = torch.zeros(3 * sum(sizes), dtype=torch.int64, device=self.dev)
        o = 0
        for r, n in zip(blocks, sizes):
            r.st = [self.stat_arena[o + k * n:o + (k + 1) * n].view(self.N, ops.STAT_REPLICAS, ops.STAT_WORDS, -1) for k in range(3)]
            o += 3 * n
        n_enc = 3 * sum(sizes[:len(self.e_res)])
        self.enc_stats, self.den_stats = self.stat_arena[:n_enc], self.stat_arena[n_enc:]
        # timestep table columns: swinViT.t_proj[0..4], then every UnetResBlock's t_proj
        self.t_lin = [den.swinViT.t_proj[i] for i in range(5)] + [r.block.t_proj for r in self.d_res + self.u_res]
        offs, o = [], 0
        for lin in self.t_lin:
            offs.append(o)
            o += -(-lin.weight.shape[0] // 8) * 8
        self.P = o
        self.vit_t_off = offs[:5]
        for r, off in zip(self.d_res + self.u_res, offs[5:]):
            r.t_off = off
        self.cur_add = torch.zeros((self.N, self.P), dtype=torch.float32, device=self.dev)
        need = 0
        for r in blocks:
            dims = self.S[r.level]
            for cin in (-(-r.cin_packed // 8) * 8, r.cout):
                need = max(need, ops.conv3_workspace_bytes(self.dtype, self.N, *dims, cin, r.cout))
        self.splitk_ws = torch.empty(max(need, 16) // 4, dtype=torch.float32, device=self.dev)
        self.splitk_ws_b = torch.empty(max(need, 16) // 4, dtype=torch.float32, device=self.dev)

    def _pack_vit(self, vit, cin_packed, perm=None):
        dt = self.dtype
        f32 = lambda p: p.detach().float().contiguous()  # noqa: E731
        out = dict(pe_w=ops.pack_patch_embed_weights(vit.patch_embed.proj.weight.detach(), cin_packed, perm),
                   pe_b=f32(vit.patch_embed.proj.bias), stages=[])
        for i, layer in enumerate(vit.stages()):
            n = self.geo[i]["n"]
            blocks = []
            for blk in layer.blocks:
                a = blk.attn
                blocks.append(dict(g1=f32(blk.norm1.weight), b1=f32(blk.norm1.bias), g2=f32(blk.norm2.weight), b2=f32(blk.norm2.bias),
                                   table=a.relative_position_bias_table.detach().float().t().contiguous(),     # [head, 13^3]
                                   wqkv=a.qkv.weight.detach().to(dt).contiguous(), bqkv=a.qkv.bias.detach().to(dt).contiguous(),
                                   wproj=a.proj.weight.detach().to(dt).contiguous(), bproj=a.proj.bias.detach().to(dt).contiguous(),
                                   w1=blk.mlp.linear1.weight.detach().to(dt).contiguous(), bb1=blk.mlp.linear1.bias.detach().to(dt).contiguous(),
                                   w2=blk.mlp.linear2.weight.detach().to(dt).contiguous(), bb2=blk.mlp.linear2.bias.detach().to(dt).contiguous(),
                                   fqkv=f32(a.qkv.bias), fproj=f32(a.proj.bias), f1=f32(blk.mlp.linear1.bias), f2=f32(blk.mlp.linear2.bias)))
            out["stages"].append(dict(blocks=blocks, gm=f32(layer.downsample.norm.weight), bm=f32(layer.downsample.norm.bias),
                                      wred=layer.downsample.reduction.weight.detach().to(dt).contiguous()))
        return out

    def refresh_weights(self):
        """Re-pack when any parameter changed (load_state_dict, optimizer step)."""
        ver = tuple((p.data_ptr(), p._version) for p in self.net.parameters())
        if ver == self.weights_version:
            return
        dt, dev = self.dtype, self.dev
        with torch.no_grad():
            for r in self.e_res + self.d_res + self.u_res:
                b = r.block
                r.w1, r.b1 = ops.pack_conv3_weights(b.conv1.conv.weight.detach().float().contiguous(), None, dt,
                                                    cin_packed=r.cin_packed if r.cin_packed != r.cin else None, perm=r.perm,
                                                    tap_channel=r.tap)
                r.w2, r.b2 = ops.pack_conv3_weights(b.conv2.conv.weight.detach().float().contiguous(), None, dt)
                if r.has3:
                    w3 = torch.zeros((r.cout, r.cin_packed), dtype=dt, device=dev)
                    w3[:, :r.cin] = b.conv3.conv.weight.detach().reshape(r.cout, r.cin)[:, r.perm or list(range(r.cin))].to(dt)
                    r.w3 = w3
                r.ones = torch.ones(r.cout, dtype=torch.float32, device=dev)
                r.zeros = torch.zeros(r.cout, dtype=torch.float32, device=dev)
                r.norms = None
            self.up_packed = [ops.pack_deconv_weights(u.transp_conv.conv.weight.detach().float().contiguous(), None, dt)
                              for u in self.ups]
            for k in range(5):
                r = self.u_res[k]
                r.fold = None
                if self.fold_up[k]:
                    wd = self.ups[k].transp_conv.conv.weight.detach().float().contiguous()
                    r.fold = ops.pack_upconv_weights(r.block.conv1.conv.weight.detach().float().contiguous(), None, wd, None,
                                                     r.cout, up_first=True, cu_packed=self.dec[k + 1].shape[-1])
                    r.res_fold = None
                    if self.res_up[k]:
                        r.res_fold = ops.pack_deconv_res_weights(r.block.conv3.conv.weight.detach().reshape(r.cout, r.cin), wd, r.cout,
                                                                 dt, up_first=True)
            enc, den = self.net.embed_model, self.net.model
            self.e_vit = self._pack_vit(enc.swinViT, 8)
            self.d_vit = self._pack_vit(den.swinViT, self.cin0, self.perm0)
            self.wf = torch.zeros((self.C, self.tail_k), dtype=torch.float32, device=dev)
            self.wf[:, :self.f] = den.out.conv.conv.weight.detach().float().reshape(self.C, -1)
            self.bf = den.out.conv.conv.bias.detach().float().contiguous()
            # t_proj(swish(t_embedder(t))) of every projection, for every original timestep (depends on weights only)
            T = self.net.timesteps
            te = den.t_embedder
            half = te.embedding_dim // 2
            freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).to(dev)
            wcat = torch.zeros((self.P, te.dense[1].weight.shape[0]), dtype=torch.float32, device=dev)
            bcat = torch.zeros(self.P, dtype=torch.float32, device=dev)
            o = 0
            for lin in self.t_lin:
                c = lin.weight.shape[0]
                wcat[o:o + c] = lin.weight.detach().float()
                bcat[o:o + c] = lin.bias.detach().float()
                o += -(-c // 8) * 8
            ts = torch.arange(T, dtype=torch.int32, device=dev)
            f32 = lambda p: p.detach().float().contiguous()  # noqa: E731
            self.temb_table = ops.temb_table(ts, freqs, f32(te.dense[0].weight), f32(te.dense[0].bias), f32(te.dense[1].weight),
                                             f32(te.dense[1].bias), wcat, bcat)
        self.weights_version = ver
        self.graphs.clear()

    # ---- building blocks ------------------------------------------------------------------------
    def _view(self, flat, level, c):
        n = self.N * self.S[level][0] * self.S[level][1] * self.S[level][2] * c
        return flat[:n].view(self.N, *self.S[level], c)

    def _tadd(self, off, c):
        return None if off is None else self.cur_add[:, off:off + c]

    def _res_block(self, r, x, cin, out, out_off=0, post_add=None, ra=None, ra_off=0, side=False, defer=False, up_src=None):
        """UnetResBlock.forward (blocks.py:298-316) on channels [0, cin) of ``x`` -> channels [out_off, ...) of ``out``.
        ``side``: use the second set of scratch buffers (blocks running on the side stream, see denoiser_body).
        ``defer``: do not materialise the output; return (raw2, norm2, res, norm3) for a consumer that assembles it (tail)."""
        l, N = r.level, self.N
        count = self.S[l][0] * self.S[l][1] * self.S[l][2]
        b1, b2, b3, ws = (self.raw1b, self.raw2b, self.res3b, self.splitk_ws_b) if side else (self.raw1, self.raw2, self.res3, self.splitk_ws)
        raw1, raw2 = self._view(b1, l, r.cout), self._view(b2, l, r.cout)
        if r.norms is None:
            add = None if r.t_off is None else self.cur_add.view(-1)[r.t_off:]
            r.norms = (ops.Norm(r.st[0], r.ones, r.zeros, count, add=add, add_stride=self.P if add is not None else 0,
                                slope=SLOPE, eps=EPS),
                       ops.Norm(r.st[1], r.ones, r.zeros, count, slope=SLOPE, eps=EPS),
                       ops.Norm(r.st[2], r.ones, r.zeros, count, slope=SLOPE, eps=EPS))
        n1, n2, n3 = r.norms
        bg = side and self.two_streams and self.background_convs
        assert r.has3 or not defer

        def convs():
            if up_src is not None:
                # conv1 over cat((up, skip)) with the transposed convolution folded in: the skip half of x, the coarse activation
                w_skip, wu, btab = r.fold
                ops.upconv_k3(x, cin - r.cout, r.cout, up_src, up_src.shape[-1], 0, None, w_skip, wu, btab, r.cout, raw1, 0, r.st[0])
            else:
                ops.conv3d_k3(x, cin, 0, r.w1, r.b1, r.cout, raw1, 0, r.st[0], workspace=ws, background=bg, tap_channel=r.tap)
            ops.conv3d_k3(raw1, r.cout, 0, r.w2, r.b2, r.cout, raw2, 0, r.st[1], norm=n1, workspace=ws, background=bg)

        # conv3 reads the block's input only, so on the side stream it could go first and land under the Swin chain's small GEMMs
        # instead of under the decoder's split-K finish kernels (which read 95 MB and crawl next to another streaming kernel);
        # measured: +0.01 ms (tools/bench_swin_ab.py), so the order of the reference stays
        first3 = side and self.conv3_first
        if not first3:
            convs()
        if r.has3:
            res = self._view(b3, l, r.cout)
            x2 = x.view(-1, x.shape[-1])[:, :cin] if cin != x.shape[-1] else x.view(-1, cin)
            if up_src is not None and getattr(r, "res_fold", None) is not None:
                wp3, ws3 = r.res_fold                # conv3 over cat((up, skip)) from the coarse tensor and the skip half: no `up`
                ops.deconv_res(up_src, self.dec_c[l + 1], 0, wp3, x, cin - r.cout, r.cout, ws3, r.cout, res, 0, r.st[2])
            elif self.fused_linear and self.tl_conv3 and r.cout <= 64 and cin <= 384:
                ops.token_linear(x2, r.w3, None, "stats", out=res.view(-1, r.cout), stats=r.st[2], samples=N,
                                 background=self.background_conv3 if bg else 0)      # conv3 + norm3 sums
            elif self.wide_gemm:
                ops.token_gemm(x2, r.w3, None, "plain", out=res.view(-1, r.cout),
                               workspace=self._gemm_scratch_b if side else self._gemm_scratch)    # 1x1x1 conv3 on the tiled MFMA GEMM
                ops.instnorm_stats(res, r.cout, r.st[2])
            else:
                ops.linear_f32(x2, r.w3, out=res.view(-1, r.cout))                   # fp32 parity mode: the exact-fp32 MFMA kernel
                ops.instnorm_stats(res, r.cout, r.st[2])
            if first3:
                convs()
            if defer:
                return raw2, n2, res, n3
            ops.residual_norm_act(raw2, n2, res, n3, slope=SLOPE, out=out, out_off=out_off, post_add=post_add, ra_src=ra,
                                  ra_off=ra_off, background=bg and self.background_tails)
        else:
            if first3:
                convs()
            ops.residual_norm_act(raw2, n2, x, None, slope=SLOPE, out=out, out_off=out_off, post_add=post_add, ra_src=ra,
                                  ra_off=ra_off, background=bg and self.background_tails)

    def _swin(self, vit, xin, cin_packed, t_offs, emb, outs, ready=None):
        """SwinTransformer.forward (transformer.py:270-316): patch embedding, four stages, the adds between them.
        ``outs[i]`` = (buffer, channel offset) receiving hidden_states_out[i] (+ emb[i]); ``ready(i)`` is called once the
        launch producing it has been enqueued."""
        N, dt = self.N, self.dtype
        tadd = lambda i: None if t_offs is None else self._tadd(t_offs[i], self.tok_c[i])  # noqa: E731
        ops.patch_embed(xin, cin_packed, vit["pe_w"], vit["pe_b"], outs[0][0], outs[0][1], tadd=tadd(0),
                        emb=None if emb is None else emb[0], x=self.stream[0])
        if ready is not None:
            ready(0)
        for i in range(4):
            g, st, C_ = self.geo[i], vit["stages"][i], self.tok_c[i]
            x = self.stream[i]
            ntok_w = N * g["nw"] * g["n"]
            ntok = x.numel() // C_
            win = self.win[:ntok_w * C_].view(N * g["nw"], g["n"], C_)
            att = self.att[:ntok_w * C_].view(N * g["nw"], g["n"], C_)
            ln2 = self.ln2[:ntok * C_].view(ntok, C_)
            y = None
            fused = self.fused_linear and C_ <= self.fused_max_c     # tall token GEMMs with fused epilogues (swin_gemm.hip)
            wide = self.wide_gemm and not fused
            if fused or wide:
                qkv_buf = self.qkv_buf[:ntok_w * 3 * C_].view(N * g["nw"], g["n"], 3 * C_)
                hid = self.hid_buf[:ntok * 4 * C_].view(ntok, 4 * C_)
            for k, b in enumerate(st["blocks"]):
                shifted = k % 2 == 1 and any(g["ss"])
                geom = g["g1"] if shifted else g["g0"]
                ops.window_gather_norm(x, geom, b["g1"], b["b1"], win, y=y)
                if fused and self.tl_qkv:
                    qkv = ops.token_linear(win.view(-1, C_), b["wqkv"], b["fqkv"], "plain", out=qkv_buf)
                elif wide:
                    qkv = ops.token_gemm(win.view(-1, C_), b["wqkv"], b["fqkv"], "plain", out=qkv_buf, workspace=self._gemm_scratch)
                else:
                    qkv = ops.linear_f32(win, b["wqkv"], b["bqkv"])
                ops.window_attention(qkv, HEADS[i], None, region_ids=g["region"] if shifted else None,
                                     windows_per_image=g["nw"], out=att, bias_table=b["table"], table_grid=WINDOW)
                if fused:
                    if self.tl_proj:
                        ops.token_linear(att.view(-1, C_), b["wproj"], b["fproj"], "scatter", x=x, geom=geom, gamma=b["g2"],
                                         beta=b["b2"], ln_out=ln2)
                    else:
                        po = self.po_buf[:ntok_w * C_].view(-1, C_)
                        ops.token_linear(att.view(-1, C_), b["wproj"], b["fproj"], "plain", out=po)
                        ops.window_scatter_add_norm(x, geom, po.view(N * g["nw"], g["n"], C_), b["g2"], b["b2"], ln2)
                    if self.fused_mlp:                                             # linear1 + GELU + linear2 + residual, one launch
                        ops.swin_mlp(ln2, b["w1"], b["f1"], b["w2"], b["f2"], x)
                    else:
                        ops.token_linear(ln2, b["w1"], b["f1"], "gelu", out=hid)
                        ops.token_linear(hid, b["w2"], b["f2"], "residual", x=x)   # x + mlp(norm2(x)) lands in the stream
                elif wide:
                    po = self.po_buf[:ntok_w * C_].view(N * g["nw"], g["n"], C_)
                    ops.token_gemm(att.view(-1, C_), b["wproj"], b["fproj"], "plain", out=po, workspace=self._gemm_scratch)
                    ops.window_scatter_add_norm(x, geom, po, b["g2"], b["b2"], ln2)
                    ops.token_gemm(ln2, b["w1"], b["f1"], "gelu", out=hid, workspace=self._gemm_scratch)                       # linear1 + GELU
                    ops.token_gemm(hid, b["w2"], b["f2"], "residual", x=x.view(-1, C_), workspace=self._gemm_scratch)         # x + mlp(norm2(x)) on the stream
                else:
                    po = ops.linear_f32(att, b["wproj"], b["bproj"])
                    ops.window_scatter_add_norm(x, geom, po, b["g2"], b["b2"], ln2)
                    h = ops.linear_f32(ln2, b["w1"], b["bb1"], gelu=True)          # linear1 + exact GELU
                    y = ops.linear_f32(h, b["w2"], b["bb2"])
            dims = g["dims"]
            mshape = (N, (dims[0] + 1) // 2, (dims[1] + 1) // 2, (dims[2] + 1) // 2, 8 * C_)
            merged = self.merged[:mshape[0] * mshape[1] * mshape[2] * mshape[3] * mshape[4]].view(mshape)
            ops.patch_merge_norm(x, st["gm"], st["bm"], legacy=True, y=y, dtype=dt, out=merged)
            if fused and 8 * C_ <= 384 and self.fused_reduction:
                red = ops.token_linear(merged.view(-1, 8 * C_), st["wred"], None, "plain",
                                       out=self.red_buf[:merged.numel() // 4].view(-1, 2 * C_))
            elif self.wide_gemm:
                red = ops.token_gemm(merged.view(-1, 8 * C_), st["wred"], None, "plain",
                                     out=self.red_buf[:merged.numel() // 4].view(-1, 2 * C_), workspace=self._gemm_scratch)
            else:
                red = ops.linear_f32(merged.view(-1, 8 * C_), st["wred"])
            ops.stage_out(red, N, 2 * C_, outs[i + 1][0], outs[i + 1][1], tadd=tadd(i + 1),
                          emb=None if emb is None else emb[i + 1], x=self.stream[i + 1] if i < 3 else None)
            if ready is not None:
                ready(i + 1)

    # ---- the two networks ---------------------------------------------------------------------------
    def run_encoder(self, image):
        """SwinUNETREncoder.forward (encoder.py:212-219): fills e_hs[0..4], e_enc[0..3] (channels-last)."""
        self.refresh_weights()
        N = self.N
        assert tuple(image.shape) == (N, 1, *self.dims), f"image shape {tuple(image.shape)} != plan {(N, 1, *self.dims)}"
        img = image.detach().float().contiguous()
        ops.to_channels_last(img, self.img_in, 0, 8)
        ops.to_channels_last(img, self.xin, self.C, self.cin0 - self.C)     # conditioning channel of the denoiser input
        self.enc_stats.zero_()
        self._swin(self.e_vit, self.img_in, 8, None, None, [(self.e_hs[i], 0) for i in range(5)])
        srcs = [(self.img_in, 8), (self.e_hs[0], self.f), (self.e_hs[1], 2 * self.f), (self.e_hs[2], 4 * self.f)]
        for r, (x, cin), out in zip(self.e_res, srcs, self.e_enc):
            self._res_block(r, x, cin, out)
        self.emb_token += 1
        return SwinEmbeddings(self, self.emb_token)

    def stage_condition(self, image, embeddings):
        """Make sure the denoiser's conditioning (image channel of xin, the nine encoder maps) is resident."""
        if isinstance(embeddings, SwinEmbeddings) and embeddings.plan is self and embeddings.token == self.emb_token:
            return
        img = image.detach().float().contiguous()
        assert tuple(img.shape) == (self.N, 1, *self.dims)
        ops.to_channels_last(img, self.xin, self.C, self.cin0 - self.C)
        for k in range(5):
            e = embeddings[0][k]
            ops.to_channels_last(e.detach().float().contiguous(), self.e_hs[k], 0, e.shape[1])
        for k in range(4):
            e = embeddings[k + 1]
            ops.to_channels_last(e.detach().float().contiguous(), self.e_enc[k], 0, e.shape[1])
        self.emb_token += 1

    def denoiser_body(self, zero_stats=True):
        """SwinUNETRDenoiser.forward from the staged input (self.xin) to channels-last logits; self.cur_add holds the
        t_proj rows of this evaluation.  ``zero_stats=False``: the caller's step_begin(clear=self.den_stats) already
        zeroed the statistics arena in its own launch."""
        f = self.f
        if zero_stats:
            self.den_stats.zero_()
        cat, dec, hs = self.cat, self.dec, self.hs
        outs = [(hs[0], 0), (hs[1], 0), (hs[2], 0), (cat[4], 8 * f), (hs[4], 0)]
        # Two streams.  The coarse Swin stages (12^3 and 6^3 tokens), encoder10 and decoder5..3 are a chain of ~150 launches of
        # 5-20 us that leave most of the chip idle; encoder1..4 (denoiser.py:370-383) depend only on the input and on
        # hidden_states_out[0..2] and are needed late (by decoder1..4, denoiser.py:388-397) -- with the two 96^3
        # convolutions of encoder1 needed last.  They go to a side stream once stage 1 has run (stages 0-1 fill the chip by
        # themselves: overlapping those only slows both), smallest first, and each decoder waits for its skip only.
        # enc_k = encoder_k(...) + embeddings[k + 1] lands in the skip half of its decoder's concat buffer.
        main, side = torch.cuda.current_stream(), self.side_stream
        two = self.two_streams
        srcs = [(self.xin, self.cin0), (hs[0], f), (hs[1], 2 * f), (hs[2], 4 * f)]

        def enc(k):
            r = self.d_res[k]
            self._res_block(r, srcs[k][0], srcs[k][1], cat[k], r.cout, post_add=self.e_enc[k], side=two)

        def ready(i):                                   # hidden_states_out[i] has been enqueued on the main stream
            if two and self.side_plan.get(i):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    for k in self.side_plan[i]:
                        enc(k)
                        self.enc_done[k].record(side)

        self._swin(self.d_vit, self.xin, self.cin0, self.vit_t_off, self.e_hs, outs, ready)
        if not two:
            for k in range(4):
                enc(k)
        self._res_block(self.d_res[4], hs[4], 16 * f, dec[5])                     # dec4 = encoder10(hs[4])
        src = dec[5]
        for k in (4, 3, 2, 1, 0):                                                  # decoder5 .. decoder1
            cout = cat[k].shape[-1] // 2
            if not self.res_up[k]:                                                  # (else: nobody reads the upsampled tensor)
                wp, bp = self.up_packed[k]
                ops.deconv_k2s2(src, self.dec_c[k + 1], 0, wp, bp, cout, cat[k], 0)
            if two and k < 4:
                main.wait_event(self.enc_done[k])                                  # the skip half of cat[k]
            ra = cat[k] if k < 4 else None                                         # + r_k (not for decoder5: skip = hs[3])
            if k == 0 and self.fused_tail:
                # decoder1's output is consumed by the `out` convolution only: the tail assembles it from the block's two
                # branches instead of reading it back (saves writing and re-reading 2 x 113 MB at 96^3)
                self._tail_src = self._res_block(self.u_res[0], cat[0], 2 * cout, dec[0], 0, defer=True,
                                                 up_src=src if self.fold_up[0] else None) + (cat[0], cout, cout)
                break
            self._res_block(self.u_res[k], cat[k], 2 * cout, dec[k], 0, ra=ra, ra_off=cout, up_src=src if self.fold_up[k] else None)
            src = dec[k]

    def _gemm_scratch(self, nbytes, which=0):
        """K-split scratch of the token GEMMs, owned by this plan (its graphs bake the address in); ``which`` = 1: the second
        buffer, for launches on the side stream (reuse is ordered by the stream).  Grown by the warm-up pass that precedes
        every capture; growing it inside a capture would put it into that graph's private memory pool."""
        buf = self._gemm_ws[which]
        if buf is None or buf.numel() * 4 < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("SwinPlan: the token-GEMM scratch must be sized by the warm-up pass, not inside a capture")
            if buf is not None:
                self._gemm_ws_retired.append(buf)          # an earlier graph of this plan may still hold its address
            buf = self._gemm_ws[which] = torch.empty(max(int(nbytes), 16 << 20) // 4, dtype=torch.float32, device=self.dev)
        return buf

    def _gemm_scratch_b(self, nbytes):
        return self._gemm_scratch(nbytes, 1)

    def capture_step(self, step_fn):
        """Record ``step_fn`` (one sampler step: step_begin + denoiser_body + tail on the current stream) into a HIP graph;
        the result has ``replay(times=1)``."""
        step_fn()                                   # warm-up outside capture (kernel attributes, GEMM workspaces)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step_fn()
        return _OneGraph(g)

    def tail(self, mode, noise=None, logits=None, use_sum=False):
        """UnetOutBlock (1x1x1, denoiser.py:399-400) fused with the sampler update (engine.Plan.tail's kernel, fed the
        materialised decoder1 output)."""
        raw, norm, residual = self.dec[0], None, None
        if self.fused_tail:
            raw, norm, res, n3, ra, ra_off, ch = self._tail_src
            residual = (res, n3, ra, ra_off, ch)
        ops.final_conv_sampler(raw, self.tail_k, norm, self.wf, self.bf, self.C, mode, coef=self.cur_coef,
                               x_state=self.x_state, noise=noise, step_word=self.step_word,
                               xin=self.xin if mode != nv.MODE_LOGITS else None,
                               xstart_sum=self.x_sum if use_sum else None, logits=logits, seed_dev=self.seed_word,
                               residual=residual)

    def new_seed(self, seed=None):
        """Philox key of this call's in-kernel noise (engine.Plan.new_seed)."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    seed ^= (dist.get_rank() + 1) * 0x9E3779B97F4A7C15 & (2 ** 62 - 1)
            except Exception:       # pragma: no cover - torch.distributed not built
                pass
        self.seed_word.fill_(int(seed) & (2 ** 63 - 1))
        return seed

    def denoise(self, x, t):
        """logits = model(x, t, image, embeddings) for an already-staged image (denoiser.py:353-403)."""
        self.refresh_weights()
        N = self.N
        assert tuple(x.shape) == (N, self.C, *self.dims) and t.numel() == N
        T = self.temb_table.shape[0]
        on_host = not t.is_cuda
        if on_host and not bool(((t >= 0) & (t < T)).all()):
            raise ValueError(f"timestep out of range: the model was built for 0 <= t < {T}, got {t.tolist()}")
        rows = t.detach().to(device=self.dev, dtype=torch.int32).contiguous()
        if not on_host:
            self.err_word.zero_()
        ops.to_channels_last(x.detach().float().contiguous(), self.xin, 0, self.C)
        ops.step_begin(N, self.temb_table, self.cur_add, rows_per_sample=rows, err_word=self.err_word, clear=self.den_stats)
        self.denoiser_body(zero_stats=False)
        out = torch.empty((N, self.C, *self.dims), dtype=torch.float32, device=self.dev)
        self.tail(nv.MODE_LOGITS, logits=out)
        if not on_host and int(self.err_word.item()):
            raise ValueError(f"timestep out of range: the model was built for 0 <= t < {T}")
        return out

    def sample_loop(self, diffusion, kind, noise=None, step_noise=None, eta=0.0, use_graph=True, seed=None, want_sum=None):
        """T reverse steps from ``noise`` (x_T, NCDHW) or a fresh draw: the loop bodies of p_sample_loop_progressive /
        ddim_sample_loop_progressive (gaussian_diffusion.py:487-535, 667-716) around SwinUNETRDenoiser.forward, one
        captured HIP graph replayed per step.  ``want_sum`` (default: DDIM loops only, see engine.Plan.sample_loop): keep the sum of
        the x0 predictions.  Returns dict(sample, sum_pred_xstart (None without the sum))."""
        want_sum = (kind == "ddim") if want_sum is None else bool(want_sum)
        self.refresh_weights()
        N, T = self.N, diffusion.num_timesteps
        shape = (N, self.C, *self.dims)
        if noise is None:
            noise = torch.randn(*shape, device=self.dev)
        assert tuple(noise.shape) == shape
        x_T = noise.detach().float().contiguous()

        def reset():
            ops.to_channels_last(x_T, self.x_state, 0, self.cx)
            ops.to_channels_last(x_T, self.xin, 0, self.C)
            self.x_sum.zero_()
            self.counter.zero_()

        reset()
        mode = nv.MODE_DDPM if kind == "ddpm" else nv.MODE_DDIM
        tkey = (diffusion, kind, float(eta))          # the object itself: the table keeps it alive, no id() reuse after GC
        if tkey not in self.tables:
            order = list(range(T))[::-1]
            tt = torch.tensor(order)
            coef = diffusion.ddpm_coef(tt) if kind == "ddpm" else diffusion.ddim_coef(tt, eta)
            tmap = diffusion.model_timesteps()
            self.tables[tkey] = (coef.to(self.dev).contiguous(),
                                 torch.tensor([tmap[i] for i in order], dtype=torch.int32, device=self.dev))
        coef_table, row_of_step = self.tables[tkey]
        self.new_seed(seed)
        if step_noise is not None:
            assert len(step_noise) == T
            use_graph = False

        def one_step(eps):
            ops.step_begin(N, self.temb_table, self.cur_add, row_of_step=row_of_step, counter=self.counter,
                           coef_table=coef_table, cur_coef=self.cur_coef, step_word=self.step_word, err_word=self.err_word,
                           clear=self.den_stats)
            self.denoiser_body(zero_stats=False)
            self.tail(mode, noise=eps, use_sum=want_sum)

        if not use_graph:
            for k in range(T):
                one_step(None if step_noise is None else step_noise[k].detach().to(self.dev).float().contiguous())
        else:
            g = self.graphs.get(tkey + (want_sum,))
            if g is None:
                g = self.capture_step(lambda: one_step(None))
                self.graphs[tkey + (want_sum,)] = g
                reset()
            g.replay(T)
        return {"sample": ops.from_channels_last(self.x_state, self.C),
                "sum_pred_xstart": ops.from_channels_last(self.x_sum, self.C) if want_sum else None}
