"""STrajNet model -- MI355X-native host side, mirroring the reference call surface.

    STrajNet(cfg, model_name='STrajNet', use_pyramid=True, actor_only=True, sep_actors=False, fg_msa=False,
             use_last_ref=False, fg=False, large_ogm=True)                       # reference modules.py:778-779
    model(ogm, map_img, training=True, obs=None, occ=None, mapt=None, flow=None, dense_vec=None, dense_map=None)
        -> [B, Hg, Hg, 32] float32                                              # reference modules.py:815-839

Tensors are torch CUDA (ROCm) tensors, NHWC like the reference.  All math runs in the HIP kernels of
libstrajnet_hip.so (strajnet_amd/csrc); torch supplies memory, streams and the autograd tape.
Parameters live in ONE flat f32 buffer (+ a flat f32 gradient buffer that doubles as the data-parallel
all-reduce bucket, + a bf16 shadow in bf16 mode); names/shapes/layouts follow the reference's Keras variables
 so a name->array dict (state_dict / load_weights) exchanges weights with the oracle.

training=True draws the reference's stochastic regularisers -- DropPath linspace(0, .1, 6) over the Swin blocks
(modules.py:507,258-260), dropout .1 on every tfa-MHA's attention coefficients and after FFN1/FFN2 of the
cross-attention blocks (trajNet.py:33,71-77,195-211) -- from a counter-based RNG inside the kernels (csrc/rng.hip);
training=False is the deterministic graph.

Not built (documented gaps): actor_only=False (MapEncoder), sep_actors=True, use_last_ref=True.
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .ops import Param, ACT_NONE, ACT_ELU


def _param_spec(cfg, hb, fg_msa, fg):
    """name -> (shape, init).  Keras layouts (Dense [in,out]; Conv HWIO; Conv3D DHWIO; Conv1D [k,in,out];
    tfa-MHA [H,in,hs]/[H,hs,out]).  init in {'glorot','glorot_mha','zeros','ones','rpb','fg_rpe'}."""
    C, ws, heads, depths = cfg['embed_dim'], cfg['window_size'], cfg['num_heads'], cfg['depths']
    sp = OrderedDict()

    def add(name, shape, init):
        assert name not in sp
        sp[name] = (tuple(shape), init)

    def dense(n, i, o, bias=True):
        add(n + '/kernel', (i, o), 'glorot')
        if bias:
            add(n + '/bias', (o,), 'zeros')

    def conv(n, kh, kw, i, o, bias=True):
        add(n + '/kernel', (kh, kw, i, o), 'glorot')
        if bias:
            add(n + '/bias', (o,), 'zeros')

    def ln(n, c):
        add(n + '/gamma', (c,), 'ones')
        add(n + '/beta', (c,), 'zeros')

    def mha(n, h, i, hs, o):
        for k in ('query', 'key', 'value'):
            add(f'{n}/{k}_kernel', (h, i, hs), 'glorot_mha')
        add(f'{n}/projection_kernel', (h, hs, o), 'glorot_mha')
        add(f'{n}/projection_bias', (o,), 'zeros')

    for nm, cin in (('patch_embed_vecicle', 11), ('patch_embed_map', 3), ('patch_embed_flow', 2)):   # modules.py:490-537
        conv(nm + '/proj', 4, 4, cin, C)
        ln(nm + '/norm', C)
    ln('flow_norm', C)                                                                                # modules.py:517
    ln('all_patch_norm', C)                                                                           # modules.py:557

    def block(pre, c, h):                                                                             # modules.py:179-187,76-86
        ln(pre + '/norm1', c)
        dense(pre + '/attn/qkv', c, 3 * c)
        add(pre + '/attn/relative_position_bias_table', ((2 * ws - 1) ** 2, h), 'rpb')
        dense(pre + '/attn/proj', c, c)
        ln(pre + '/norm2', c)
        dense(pre + '/mlp/fc1', c, 4 * c)
        dense(pre + '/mlp/fc2', 4 * c, c)

    def merge(pre, c):                                                                                # modules.py:270-272
        ln(pre + '/downsample/norm', 4 * c)
        dense(pre + '/downsample/reduction', 4 * c, 2 * c, bias=False)

    for i in range(depths[0]):
        block(f'flow_layers0/blocks{i}', C, heads[0])
    merge('flow_layers0', C)
    for L in range(3):
        for i in range(depths[L]):
            block(f'layers{L}/blocks{i}', C * 2 ** L, heads[L])
        if L < 2:
            merge(f'layers{L}', C * 2 ** L)
    Cb = 4 * C
    if fg_msa:                                                                                        # FG_MSA.py:51-73
        gc = Cb // 8
        for p in ('proj_q', 'proj_k', 'proj_v', 'proj_out'):
            conv('fg_msa/' + p, 1, 1, Cb, Cb)
        conv('fg_msa/conv_offset_0', 3, 3, gc, Cb)
        ln('fg_msa/conv_norm', Cb)
        conv('fg_msa/conv_offset_proj', 1, 1, gc, 2, bias=False)
        if fg:
            conv('fg_msa/conv_offset_proj2', 1, 1, 2, Cb)
        add('fg_msa/warp_attn_rel_table', (2 * hb - 1, 2 * hb - 1, 8), 'fg_rpe')
    add('traj_net/traj_encoder/node_feature/kernel', (1, 5, 64), 'glorot')                           # trajNet.py:32-36
    add('traj_net/traj_encoder/node_feature/bias', (64,), 'zeros')
    mha('traj_net/traj_encoder/node_attention', 4, 64, 64, 320)
    dense('traj_net/traj_encoder/vector_feature', 3, 64, bias=False)
    dense('traj_net/traj_encoder/sublayer', 384, Cb)
    mha('traj_net/cross_attention/mha', 6, Cb, Cb // 6, Cb)                                           # trajNet.py:71-76
    ln('traj_net/cross_attention/norm1', Cb)
    ln('traj_net/cross_attention/norm2', Cb)
    dense('traj_net/cross_attention/FFN1', Cb, 4 * Cb)
    dense('traj_net/cross_attention/FFN2', 4 * Cb, Cb)
    ln('traj_net/obs_norm', Cb)
    ln('traj_net/occ_norm', Cb)
    dense('traj_net/seg_embed', 2, Cb, bias=False)
    for i in range(8):                                                                                # trajNet.py:195-210,257
        p = f'cross_attn_obs{i}'
        mha(p + '/mha', 3, Cb, 128 // 3, 128)
        ln(p + '/norm1', 128)
        dense(p + '/FFN1', 128, 512)
        dense(p + '/FFN2', 512, Cb)
        ln(p + '/norm2', Cb)
    ch = [48, 96, 128, 192, 384]                                                                      # modules.py:635-730
    cin = Cb
    for i in (3, 2, 1, 0):
        conv(f'decoder/upconv_{i}_0', 3, 3, cin, ch[i])
        cin = ch[i]
    for nm, ci, co in (('resconv_3', 2 * C, ch[3]), ('resconv_2', C, ch[2]), ('resconv_f', C, 128)):
        add(f'decoder/{nm}/kernel', (8, 1, 1, ci, co), 'glorot')
        add(f'decoder/{nm}/bias', (co,), 'zeros')
    conv('decoder/upconvf_1_0', 3, 3, 128, ch[1])
    conv('decoder/upconvf_0_0', 3, 3, ch[1], ch[0])
    conv('decoder/outconv', 3, 3, ch[0], 2)
    conv('decoder/outconv_f', 3, 3, ch[0], 2)
    return sp


def _init_tensor(shape, kind, gen):
    if kind in ('glorot', 'glorot_mha'):
        if kind == 'glorot':
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fi, fo = shape[-2] * rf, shape[-1] * rf
        else:
            fi, fo = shape[1] * shape[0], shape[2] * shape[0]
        lim = math.sqrt(6.0 / (fi + fo))
        return (torch.rand(shape, generator=gen) * 2 - 1) * lim
    if kind == 'rpb':        # reference zero-initialises (modules.py:86); N(0,0.02) keeps the bias path live
        return torch.randn(shape, generator=gen) * 0.02
    if kind == 'fg_rpe':     # TruncatedNormal(0, 0.01) (FG_MSA.py:72)
        return (torch.randn(shape, generator=gen) * 0.01).clamp_(-0.02, 0.02)
    if kind == 'zeros':
        return torch.zeros(shape)
    if kind == 'ones':
        return torch.ones(shape)
    raise ValueError(kind)


class STrajNet:
    PARTS = 32
    def __init__(self, cfg, model_name='STrajNet', use_pyramid=True, actor_only=True, sep_actors=False,
                 fg_msa=False, use_last_ref=False, fg=False, large_ogm=True,
                 device='cuda', dtype=torch.float32, seed=0, dropout_seed=None):
        if not use_pyramid or not actor_only or sep_actors or use_last_ref:
            raise NotImplementedError('only use_pyramid=True, actor_only=True, sep_actors=False, use_last_ref=False '
                                      '(the configuration train.py:194 / modules.py:851 uses) is built')
        if fg and not fg_msa:
            raise ValueError('fg=True requires fg_msa=True (modules.py:828-831 reads the FG-MSA output)')
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError('dtype must be torch.float32 (parity mode), torch.bfloat16 (throughput mode) or torch.float16 '
                             '(inference mode: no loss scaling is applied to gradients)')
        H, W = cfg['input_size']
        if H != W:
            raise ValueError('square inputs only (reference modules.py:583-585)')
        if len(cfg['depths']) != 3 or len(cfg['num_heads']) != 3:
            raise ValueError('depths/num_heads must have length 3 (reference modules.py:792-801,822)')
        if cfg['window_size'] != 8 or any(cfg['embed_dim'] * 2 ** i // h != 32 for i, h in enumerate(cfg['num_heads'])):
            raise NotImplementedError('window kernels are specialised for window_size=8, head_dim=32')
        if cfg['embed_dim'] != 96:
            raise NotImplementedError('embed_dim must be 96 (decoder / trajNet hard-code 384, reference modules.py:794,822)')
        self.cfg = dict(cfg)
        self.name = model_name
        self.fg_msa, self.fg, self.large_ogm = fg_msa, fg, large_ogm
        self.device = torch.device(device)
        self.dtype = dtype
        C = cfg['embed_dim']
        self.P = H // 4
        self.stage_res = [self.P, self.P // 2, self.P // 4]
        self.stage_dim = [C, 2 * C, 4 * C]
        crop = 2 if large_ogm else 1
        self.skip_res = [r // crop for r in self.stage_res]
        self.hb = self.skip_res[2]
        self.map_size = H // 2 if large_ogm else H
        if self.stage_res[2] % 8 != 0:
            raise ValueError('input_size must be a multiple of 128')

        spec = _param_spec(cfg, self.hb, fg_msa, fg)
        total = sum(int(np.prod(s)) for s, _ in spec.values())
        # 16-byte aligned slots so vector loads of every tensor are legal
        offs, off = {}, 0
        for n, (s, _) in spec.items():
            offs[n] = off
            off += (int(np.prod(s)) + 7) // 8 * 8
        self.n_params = total
        self._offs = offs
        self._zstride = offs['cross_attn_obs1/norm1/gamma'] - offs['cross_attn_obs0/norm1/gamma']
        for n in spec:
            if n.startswith('cross_attn_obs0/'):
                for z in range(1, 8):
                    assert offs[n.replace('obs0', f'obs{z}')] - offs[n] == z * self._zstride
        self._flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        # the flat gradient buffer and the step's zeroed scratch (ops._ZeroArena) share ONE allocation: zero_grad() is one fill over the
        # gradients + the used prefix of the arena instead of two launches at the head of every step
        self._goff8 = (off + ops._ZeroArena.ALIGN - 1) // ops._ZeroArena.ALIGN * ops._ZeroArena.ALIGN
        self._gbuf = torch.zeros(self._goff8 + ops._ZeroArena.FLOATS, dtype=torch.float32, device=self.device)
        self._gflat = self._gbuf[:off]
        self._cflat = self._flat if dtype == torch.float32 else torch.zeros(off, dtype=dtype, device=self.device)
        gen = torch.Generator().manual_seed(seed)
        # DropPath schedule (modules.py:507,527,548): rates linspace(0, .1, sum(depths)); the flow stage reuses the first depths[0]
        dpr = np.linspace(0.0, 0.1, sum(cfg['depths']))
        self.drop_path_rate = {}
        for i, d in enumerate(cfg['depths']):
            for j in range(d):
                self.drop_path_rate[f'layers{i}/blocks{j}'] = float(dpr[sum(cfg['depths'][:i]) + j])
        for j in range(cfg['depths'][0]):
            self.drop_path_rate[f'flow_layers0/blocks{j}'] = float(dpr[j])
        # constants of the graph, built once: the obs/occ segment one-hots (trajNet.py:119-120)
        self._seg_onehot = {}
        # weights are seeded by `seed` (identical on every data-parallel replica); the Dropout / DropPath stream by `dropout_seed`
        # (per replica: seed + rank, like MirroredStrategy's independent per-replica draws)
        self.dropctx = ops.DropCtx(self.device, seed if dropout_seed is None else dropout_seed)
        self._arena = ops._ZeroArena()    # this model's zeroed scratch (one fill per step)
        self._arena.adopt(self._gbuf[self._goff8:])
        self._dctx = None
        self._side = ops.role_stream(self.device, 'side') if (self.device.type == 'cuda' and os.environ.get('STJ_NO_SIDE_STREAM') != '1') else None
        self._side2 = ops.role_stream(self.device, 'side2') if (self._side is not None and os.environ.get('STJ_NO_SIDE_STREAM') != '2') else None
        self._streams = (self._side, self._side2)
        self.serial = False              # True: everything on the current stream (per-kernel timing, debugging)
        self.taps = None                 # a dict: call() stores detached float copies of the stage boundaries in it (tools/bf16_attribution.py)
        # fused Swin-block kernels (csrc/swin_fused.hip); STJ_FUSED_SWIN=0 selects the layer-by-layer path (the one the f32 mode's C = 384 stage takes)
        # where the decoder's three skip GEMMs are issued on the side stream: 0 behind the encoder, 1 behind FG-MSA, 2 behind the cross-attention (round 6, the B = 32
        # inference timeline has `fgattn_fwd` at 290 us beside them against 64 alone: 0 / 1 / 2 = 5803-5962 / 5854-5889 / 5741-5771 scenes/s, training 1407 / 1389-1418 / 1210:
        # moving them moves the contention, profiles/r06_zk_skips_issue.txt)
        self.skips_issue = 0
        self.fused_mlp = self.fused_attn = os.environ.get('STJ_FUSED_SWIN', '1') != '0'
        # C = 384 (the 16x16 stage: 2048 rows = 32 row blocks / 32 windows at B = 8) runs the SPLIT variants of the fused kernels --
        # (row block | window) x (slice of the hidden dimension | of the heads) workgroups + a finishing launch.  The f32 parity mode runs the
        # MLP half of that stage through the same split kernel (round 5: the oracle gate then covers its code too) and keeps only the
        # attention half layer by layer (an f32 weight slice of one head does not fit LDS next to the window's q|k|v tile)
        self.fused_attn_dims = (96, 192, 384)      # (f32 at C = 384 since round 6: the Wqkv slice of a head staged in two halves, csrc/swin_fused.hip AttnCfg::KH)
        self.fused_mlp_dims = (96, 192, 384)
        # the 8 time-separated cross-attentions as one kernel per direction (csrc/xattn_fused.hip); False = the layer-by-layer chain
        # (the parity tests run both and compare)
        self.fused_xattn = True
        self.fused_stem = True         # PatchEmbed + the stem's sums / norms as one launch per raster (csrc/patch_embed.hip); False = im2col + dense + LayerNorm launches
        self.agent_issue_mode = 2
        self.agent_override = None            # (key, mask) from agent_encode(): call() then skips the agent branch
        self.mid_forward_hook = None          # callable run once per forward pass behind the encoder's first stage (GraphedTrainStep: loss.prepare on its side stream)
        self.fused_fgattn = True       # FG-MSA attention core as one kernel per direction (8 x 8 / 16 x 16 maps; all three storage types)
        self.kv_in_agent_branch = False # (measured neutral: 1333-1338 either way) the cross-attention's key / value projections of the agent encoding issued on the agent branch's stream
        self._xattn_kv_pre = None
        self.fused_agent = True        # TrajEncoder of all agents as one kernel per direction (csrc/agent_fused.hip); False = the layer-by-layer chain
        self.fused_agent_int = True    # ... and the 64-agent interaction block (16-bit storage types)
        # stage 0: the last block's weight gradients leave on the side stream under the first block's backward, so that the pass's final launch (122 us with
        # nothing left to run beside it) halves.  Measured 1362 / 1365 / 1363 / 1365 with against 1366 / 1366 / 1368 / 1363 without: off
        self.wg_mid_flush = False
        self.fused_fgoff = True        # FG-MSA's offset head (conv_offset -> tanh * range) as one kernel per direction (csrc/fgoff_fused.hip)
        self._agent_pack = None
        self._agent_pack_event = None
        self._agent_pack_stale = True
        self._fgoff_pack = None
        self._fgoff_pack_event = None
        self._fgoff_pack_stale = True
        self._xattn_pack = None
        self.params = OrderedDict()
        for n, (s, kind) in spec.items():
            k = int(np.prod(s))
            sl = slice(offs[n], offs[n] + k)
            self._flat[sl] = _init_tensor(s, kind, gen).reshape(-1).to(self.device)
            master = self._flat[sl].view(s).requires_grad_(True)
            grad = self._gflat[sl].view(s)
            master.grad = grad
            self.params[n] = Param(n, s, master, self._cflat[sl].view(s), grad)
        # Partial-gradient copies.  Some backward kernels end with hundreds of workgroups adding into the same few addresses
        # (LayerNorm gamma / beta, the window-attention bias tables, the up-conv biases); same-address atomics serialise at ~35 ns
        # each, so these parameters get PARTS copies (contiguous per parameter) that the workgroups rotate over, and ONE
        # index_add folds all of them into the flat gradient buffer at the end of backward (_fold_partials).  The 8-set batched
        # norms of the time-separated attention keep the direct path (their parameter stride is that of the flat buffer).
        def wants_parts(n):
            leaf = n.rsplit('/', 1)[-1]
            if n.startswith('cross_attn_obs'):
                return False
            return leaf in ('gamma', 'beta', 'relative_position_bias_table') or (leaf == 'bias' and '/upconv' in n)
        names = [n for n in spec if wants_parts(n)]
        P = self.PARTS
        self._parts = torch.zeros(P * sum(int(np.prod(spec[n][0])) for n in names), dtype=torch.float32, device=self.device)
        idx, o = [], 0
        for n in names:
            k = int(np.prod(spec[n][0]))
            self.params[n].part = (self._parts[o:o + k], P, k)
            idx.append(torch.arange(offs[n], offs[n] + k).repeat(P))
            o += P * k
        self._fold_index = torch.cat(idx).to(self.device)
        # Two gradient buckets for data parallelism (dp.BucketedStep): the raster ENCODER's parameters come first in the flat buffer
        # (patch embeds, Swin stages); everything downstream of the encoder outputs (FG-MSA, trajNet, cross-attentions, decoder)
        # is the tail.  Backward produces the tail first, so its all-reduce runs under the encoder's backward.
        enc = ('patch_embed', 'flow_norm', 'all_patch_norm', 'flow_layers', 'layers')
        first_tail = next(n for n in spec if not n.startswith(enc))
        assert all(not n.startswith(enc) for n in list(spec)[list(spec).index(first_tail):])
        self.bucket_split = offs[first_tail]                                  # elements of the flat buffer in the encoder bucket
        self._parts_split = P * sum(int(np.prod(spec[n][0])) for n in names if n.startswith(enc))
        self.cut_encoder = False         # True: autograd is cut at the encoder outputs (backward() stops there; backward_encoder() finishes)
        self._cut_src = self._cut_leaf = None
        self._upconv_prep = {}
        self._xattn_pack_stale = True
        self._prep_event = None
        self.weights_version = 0          # bumped by load_weights (callers that cache results of the weights compare it)
        self._sync_compute_weights()

    # ------------------------------------------------------------------ weights
    def _sync_compute_weights(self):
        if self.dtype != torch.float32:
            ops.call('stj_cast', ops._p(self._flat), 0, ops._p(self._cflat), ops.DTYPE_CODE[self.dtype], self._flat.numel(), ops._st())

    def state_dict(self):
        return OrderedDict((n, p.master.detach().cpu().numpy().copy()) for n, p in self.params.items())

    def save_weights(self, path):
        """`model.save_weights(path)` (train.py:358,366): a TF-format checkpoint (`path.index` + `path.data-00000-of-00001`)
        keyed by the reference's object-graph paths -- see strajnet_amd.checkpoint."""
        from . import checkpoint
        checkpoint.save_tf_checkpoint(path, self.state_dict(), self.cfg, self.fg_msa, self.fg)

    def load_weights(self, weights):
        """`model.load_weights(path)` (train.py:372, inference.py:283): `weights` is the prefix of a TF-format checkpoint
        written by the reference (or by save_weights), or a dict name -> array (Keras layouts, App. B names)."""
        if isinstance(weights, (str, os.PathLike)):
            from . import checkpoint
            weights = checkpoint.load_tf_checkpoint(os.fspath(weights), self.cfg, self.fg_msa, self.fg)
        missing = [n for n in self.params if n not in weights]
        extra = [n for n in weights if n not in self.params]
        if missing or extra:
            raise KeyError(f'load_weights: missing {missing[:5]} extra {extra[:5]}')
        with torch.no_grad():
            for n, p in self.params.items():
                w = torch.from_numpy(np.array(weights[n], dtype=np.float32))       # (copy: checkpoint arrays are read-only maps)
                if tuple(w.shape) != p.shape:
                    raise ValueError(f'{n}: shape {tuple(w.shape)} != {p.shape}')
                p.master.copy_(w.to(self.device))
        self.weights_version += 1
        self._sync_compute_weights()

    def trainable_weights(self):
        return [p.master for p in self.params.values()]

    def parameters(self):
        return self.trainable_weights()

    def flat_grads(self):
        """The flat f32 gradient buffer (also the data-parallel all-reduce bucket)."""
        return self._gflat

    def flat_weights(self):
        return self._flat

    def zero_grad(self):
        self._gbuf[:self._goff8 + self._arena.off].zero_()      # the gradients and what the last step took from the arena: one fill
        self._arena.rearm()                  # the step's zeroed scratch comes out of this model's arena
        ops.use_arena(self._arena)

    def _fold_partials(self, which=None):
        """Partial-gradient copies -> flat gradient buffer (runs on the caller's stream when backward() has been enqueued).
        which: None = all, 'tail' / 'encoder' = the copies of that gradient bucket only (cut_encoder mode)."""
        k = self._parts_split
        lo, hi = {None: (0, self._parts.numel()), 'encoder': (0, k), 'tail': (k, self._parts.numel())}[which]
        if hi > lo:
            ops.call('stj_fold_parts', ops._p(self._gflat), ops._p(self._fold_index[lo:hi]), ops._p(self._parts[lo:hi]), hi - lo, ops._st())

    def backward_encoder(self):
        """cut_encoder mode: the second half of backward -- from the gradients that backward() left on the encoder outputs down to
        the rasters' patch embeddings.  Ends with the side streams joined and the encoder bucket's partial copies folded, so the
        caller can all-reduce flat_grads()[:bucket_split] next."""
        if self._cut_src is None:
            raise RuntimeError('backward_encoder(): no cut forward pending (set model.cut_encoder = True before the call)')
        src, leaf = self._cut_src, self._cut_leaf
        self._cut_src = self._cut_leaf = None
        pairs = [(s_, l.grad) for s_, l in zip(src, leaf) if l.grad is not None]
        ops.wgrad_queue_begin(self.device)    # the encoder's dense weight gradients: queued, flushed at the stage boundaries and here
        try:
            torch.autograd.backward([a for a, _ in pairs], [g for _, g in pairs])
            main = torch.cuda.current_stream(self.device)
            for st in self._streams:
                if st is not None:
                    main.wait_stream(st)
        finally:
            with torch.no_grad():
                ops.wgrad_queue_end(self.device)
        ops.wgrad_join_now(main)
        with torch.no_grad():
            self._fold_partials('encoder')

    def grads(self):
        return OrderedDict((n, p.grad) for n, p in self.params.items())

    # ------------------------------------------------------------------ blocks
    def _p(self, name):
        return self.params[name]

    def _ln(self, x, name, eps, gather_res=0, res=None):
        return ops.layernorm(x, self._p(name + '/gamma'), self._p(name + '/beta'), eps, gather_res, res=res)

    def _ln_skip(self, x, name, eps):
        """(LayerNorm(x), x) -- x for the residual that bypasses the norm (gradient accumulation fused into the LN backward)."""
        return ops.layernorm_skip(x, self._p(name + '/gamma'), self._p(name + '/beta'), eps)

    def _dense(self, x, name, act=ACT_NONE, res=None, bias=True):
        return ops.linear(x, self._p(name + '/kernel'), self._p(name + '/bias') if bias else None, act, res)

    def _swin_block(self, x, pre, B, res, heads, shift):
        """SwinTransformerBlock.call (modules.py:220-262); the roll/partition/reverse plumbing lives in the kernel."""
        if res <= 8:
            shift = 0                                             # modules.py:173-175
        dpr = self.drop_path_rate.get(pre, 0.0) if self._dctx is not None else 0.0
        fused_mlp = self.fused_mlp and x.shape[-1] in self.fused_mlp_dims
        if self.fused_attn and x.shape[-1] in self.fused_attn_dims:
            # LN1 -> qkv -> (S)W-MSA -> proj -> DropPath -> + shortcut in ONE kernel, one workgroup per window (csrc/swin_fused.hip)
            x = ops.swin_attn_half(x, self._p(pre + '/norm1/gamma'), self._p(pre + '/norm1/beta'), self._p(pre + '/attn/qkv/kernel'),
                                   self._p(pre + '/attn/qkv/bias'), self._p(pre + '/attn/relative_position_bias_table'),
                                   self._p(pre + '/attn/proj/kernel'), self._p(pre + '/attn/proj/bias'), B, res, shift, 1e-5,
                                   self._dctx, pre + '/drop_path_attn', dpr)
            if fused_mlp:
                return ops.swin_mlp(x, self._p(pre + '/norm2/gamma'), self._p(pre + '/norm2/beta'), self._p(pre + '/mlp/fc1/kernel'),
                                    self._p(pre + '/mlp/fc1/bias'), self._p(pre + '/mlp/fc2/kernel'), self._p(pre + '/mlp/fc2/bias'), 1e-5,
                                    self._dctx, pre + '/drop_path_mlp', dpr, rows_per_sample=res * res)
            h, x = self._ln_skip(x, pre + '/norm2', 1e-5)
            h = ops.gelu(self._dense(h, pre + '/mlp/fc1'))
            if dpr == 0.0:
                return self._dense(h, pre + '/mlp/fc2', res=x)
            h = self._dense(h, pre + '/mlp/fc2').view(B, -1)
            return ops.dropout(h, dpr, self._dctx, pre + '/drop_path_mlp', res=x.view(B, -1), per_sample=True).view(x.shape)
        h, x = self._ln_skip(x, pre + '/norm1', 1e-5)
        qkv = self._dense(h, pre + '/attn/qkv')
        a = ops.win_attn(qkv, self._p(pre + '/attn/relative_position_bias_table'), B, res, heads, shift)

        def mlp_fused(x):            # LN2 -> fc1 -> GELU -> fc2 -> DropPath -> + shortcut in ONE kernel (csrc/swin_fused.hip)
            return ops.swin_mlp(x, self._p(pre + '/norm2/gamma'), self._p(pre + '/norm2/beta'), self._p(pre + '/mlp/fc1/kernel'),
                                self._p(pre + '/mlp/fc1/bias'), self._p(pre + '/mlp/fc2/kernel'), self._p(pre + '/mlp/fc2/bias'), 1e-5,
                                self._dctx, pre + '/drop_path_mlp', dpr, rows_per_sample=res * res)
        if dpr == 0.0:
            x = self._dense(a, pre + '/attn/proj', res=x)         # shortcut + attn
            if fused_mlp:
                return mlp_fused(x)
            h, x = self._ln_skip(x, pre + '/norm2', 1e-5)
            h = ops.gelu(self._dense(h, pre + '/mlp/fc1'))
            return self._dense(h, pre + '/mlp/fc2', res=x)
        # training: shortcut + DropPath(branch), one Bernoulli(keep) draw per sample and branch (modules.py:137-151,258,260)
        a = self._dense(a, pre + '/attn/proj').view(B, -1)
        x = ops.dropout(a, dpr, self._dctx, pre + '/drop_path_attn', res=x.view(B, -1), per_sample=True).view(x.shape)
        if fused_mlp:
            return mlp_fused(x)
        h, x = self._ln_skip(x, pre + '/norm2', 1e-5)
        h = ops.gelu(self._dense(h, pre + '/mlp/fc1'))
        h = self._dense(h, pre + '/mlp/fc2').view(B, -1)
        return ops.dropout(h, dpr, self._dctx, pre + '/drop_path_mlp', res=x.view(B, -1), per_sample=True).view(x.shape)

    def _tap(self, name, t):
        if self.taps is not None:
            self.taps[name] = t.detach().float().clone()

    def _basic_layer(self, x, pre, B, res, depth, heads, downsample, add=None, mid_flush=False):
        """BasicLayer.call (modules.py:351-364) -> (downsampled, pre-merge tokens)."""
        for i in range(depth):
            if mid_flush and i == depth - 1:
                # backward: the LAST block's weight gradients (and whatever else is queued) leave on the side stream while the blocks in front
                # of it still run backward -- the final launch of the pass, which nothing can hide, is then half as long
                x = ops.wgrad_queue_flush_point(x, side=True)
            x = self._swin_block(x, f'{pre}/blocks{i}', B, res, heads, 0 if i % 2 == 0 else 4)
            self._tap(f'{pre}/block{i}', x)
        if not downsample:
            return x, x
        m = self._ln(x, pre + '/downsample/norm', 1e-5, gather_res=res)      # PatchMerging (modules.py:274-292)
        if callable(add):
            add = add()                                                      # join point of a branch computed on a side stream
        return self._dense(m, pre + '/downsample/reduction', bias=False, res=add), x

    def _patch_embed(self, src, name, Cin, ch_stride, pix_stride, add=None, norm2=None):
        """PatchEmbed.call (modules.py:437-446): conv4x4/s4 + LN(1e-5) (+ `add`, the other embedding the caller sums it with)
        (+ `norm2`, the LayerNorm the caller applies to that sum: all_patch_norm modules.py:590 / flow_norm :578) -- one launch
        (csrc/patch_embed.hip); widths the fused kernel is not built for take im2col + dense + the LayerNorm kernels."""
        B, H = src.shape[0], src.shape[1]
        pw = self._p(name + '/proj/kernel')
        if self.fused_stem and ops.patch_embed_ok(Cin, pw.c.shape[-1], self.dtype):
            y = ops.patch_embed(src, pw, self._p(name + '/proj/bias'), self._p(name + '/norm/gamma'), self._p(name + '/norm/beta'),
                                Cin, ch_stride, pix_stride, self.dtype, 1e-5, add,
                                self._p(norm2 + '/gamma') if norm2 else None, self._p(norm2 + '/beta') if norm2 else None)
            return y.view(B, (H // 4) ** 2, -1)
        cols = ops.patch_im2col(src, Cin, ch_stride, pix_stride, self.dtype)
        y = self._dense(cols, name + '/proj')
        y = self._ln(y, name + '/norm', 1e-5, res=add.view(y.shape) if add is not None else None)
        if norm2:
            y = self._ln(y, norm2, 1e-5)
        return y.view(B, (H // 4) ** 2, -1)

    def _encoder(self, ogm, map_img, flow, hook=None):
        """SwinTransformerEncoder.forward_features (modules.py:570-624), sep_encode/flow_sep/use_flow branch."""
        B = ogm.shape[0]
        C, P = self.stage_dim[0], self.P
        depths, heads = self.cfg['depths'], self.cfg['num_heads']
        def flow_branch():
            fl = self._patch_embed(flow, 'patch_embed_flow', 2, 1, 2, norm2='flow_norm')
            return self._basic_layer(fl, 'flow_layers0', B, P, depths[0], heads[0], True)
        # the flow stage (2 Swin blocks at 64x64 tokens) and the vehicle/map stage 0 are independent until stage 0's PatchMerging
        # adds flow_x (modules.py:576-578,596-605): side stream, joined right before that GEMM
        side = self._side2
        if side is not None:
            main = torch.cuda.current_stream(self.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                flow_x, flow_res = flow_branch()

            def joined_flow_x():
                main.wait_stream(side)
                flow_x.record_stream(main)
                flow_res.record_stream(main)
                return flow_x
        else:
            flow_x, flow_res = flow_branch()
            joined_flow_x = flow_x
        vec = self._patch_embed(ogm, 'patch_embed_vecicle', 11, 2, 22)          # ogm[...,0]: stride-2 channel pick (:572)
        if self.large_ogm:                                                      # modules.py:582-587
            maps = self._patch_embed(map_img, 'patch_embed_map', 3, 1, 3)
            Pm = self.map_size // 4
            pad = (P - Pm) // 2
            maps = torch.nn.functional.pad(maps.view(B, Pm, Pm, C), (0, 0, pad, pad, pad, pad)).reshape(B, P * P, C)
            x = self._ln(vec + maps, 'all_patch_norm', 1e-5)
        else:
            x = self._patch_embed(map_img, 'patch_embed_map', 3, 1, 3, add=vec, norm2='all_patch_norm')   # vec + maps (modules.py:589-590)
        self._tap('stem', x)
        res_list = []

        def crop(t, r, c):
            q = r // 4
            return t.view(B, r, r, c)[:, q:q + r // 2, q:q + r // 2].reshape(B, (r // 2) ** 2, c)
        for i in range(3):
            r, c = self.stage_res[i], self.stage_dim[i]
            x, res = self._basic_layer(x, f'layers{i}', B, r, depths[i], heads[i], i < 2, add=joined_flow_x if i == 0 else None,
                                       mid_flush=self.wg_mid_flush and i == 0)
            if i < 2:       # in backward: stage i + 1 is through -> its weight gradients (and whatever else is queued) leave as one launch
                x = ops.wgrad_queue_flush_point(x)
            if i == 0:
                res_list.append(crop(flow_res, r, c) if self.large_ogm else flow_res)
                if hook is not None:
                    hook()
            res_list.append(crop(res, r, c) if self.large_ogm else res)
        return res_list

    def _fgmsa(self, x):
        """FGMSA.call (FG_MSA.py:106-183), eval semantics, plus what the caller does with its two results (modules.py:825-831):
        x [B,hb,hb,384] -> (x + y, decoder query [8,B,HW,384] = (x + y) broadcast over the waypoints + flow_hidden)."""
        B, Hh, Ww, C = x.shape
        G = 8
        gc = C // G
        HW = Hh * Ww
        with ops.gemm_group():           # three independent 1x1 convs of x (k, v use the unsampled x: App. D-3), one launch
            q = self._dense(x, 'fg_msa/proj_q')
            k = self._dense(x, 'fg_msa/proj_k')
            v = self._dense(x, 'fg_msa/proj_v')
        if self.fused_fgoff and ops.fgoff_ok(self.dtype, Hh, Ww, C, G):
            # conv_offset (grouped 3x3 conv -> LayerNorm -> gelu -> 1x1 conv gc->2) and tanh * (H/2) as ONE launch per direction (csrc/fgoff_fused.hip)
            pw = self._p('fg_msa/conv_offset_0/kernel')
            if self._fgoff_pack_stale:                   # (no second side stream: packed here)
                self._fgoff_pack = ops.fgoff_pack(pw, self.dtype, out=self._fgoff_pack)
                self._fgoff_pack_stale = False
            elif self._fgoff_pack_event is not None:     # packed with the other per-step preparations on the second side stream
                torch.cuda.current_stream(self.device).wait_event(self._fgoff_pack_event)
                self._fgoff_pack_event = None
            off = ops.fgoff_chain(q, pw, self._p('fg_msa/conv_offset_0/bias'), self._p('fg_msa/conv_norm/gamma'), self._p('fg_msa/conv_norm/beta'),
                                  self._p('fg_msa/conv_offset_proj/kernel'), self._fgoff_pack, Hh / 2.0, 1e-3)
        else:
            o = ops.grouped_conv3(q, self._p('fg_msa/conv_offset_0/kernel'), self._p('fg_msa/conv_offset_0/bias'), G)
            o = ops.gelu(self._ln(o, 'fg_msa/conv_norm', 1e-3))
            # per group: 1x1 conv gc->2 (no bias), tanh * (H/2); the kernel reads o in place ([B,H,W,G,gc]) and writes [B,G,HW,2]
            off = ops.fg_offset(o, self._p('fg_msa/conv_offset_proj/kernel'), Hh / 2.0, G)
        # the sampled relative-position bias is built from `off` inside the attention op
        if self.fused_fgattn and ops.fg_attn_ok(self.dtype, Hh, Ww, gc):      # one kernel per direction (csrc/fgattn.hip)
            a = ops.fg_attn(q.view(B, HW, C), k.view(B, HW, C), v.view(B, HW, C), off, self._p('fg_msa/warp_attn_rel_table'), Hh, Ww, gc ** -0.5)
        else:
            a = ops.mha_core(q.view(B, HW, C), k.view(B, HW, C), v.view(B, HW, C), G, gc, gc ** -0.5,
                             fg_off=off, fg=(self._p('fg_msa/warp_attn_rel_table'), Hh, Ww))
        if self.taps is not None:
            y = self._dense(a.view(B, Hh, Ww, C), 'fg_msa/proj_out')
            self._tap('fg_msa_out', y)
            xy = x + y
        else:
            xy = self._dense(a.view(B, Hh, Ww, C), 'fg_msa/proj_out', res=x)                          # modules.py:825
        if self.fg:      # flow_hidden = 1x1 conv 2 -> C of the offsets, added per waypoint (= per group) to the broadcast query
            query = ops.fg_query(off, self._p('fg_msa/conv_offset_proj2/kernel'), self._p('fg_msa/conv_offset_proj2/bias'),
                                 qres=xy.view(B, HW, C))
        else:
            query = xy.reshape(1, B, HW, C).expand(8, B, HW, C).contiguous()
        return xy, query

    def _tfa_mha(self, pre, query, key, H, qvalid, kvalid):
        """tensorflow_addons MultiHeadAttention with inputs=[query, key] (value=key), eval (trajNet.py:42,80,225)."""
        pq, pk, pv = self._p(pre + '/query_kernel'), self._p(pre + '/key_kernel'), self._p(pre + '/value_kernel')
        hs = pq.shape[-1]
        with ops.gemm_group():           # three independent projections, one launch
            q = ops.linear_heads_in(query, pq)
            k = ops.linear_heads_in(key, pk)
            v = ops.linear_heads_in(key, pv)
        o = ops.mha_core(q, k, v, H, hs, 1.0 / math.sqrt(hs), qvalid=qvalid, kvalid=kvalid,
                         drop=self._attn_drop(pre + '/dropout', (q.shape[0], H, q.shape[1], k.shape[1])))
        return ops.linear_heads_out(o, self._p(pre + '/projection_kernel'), self._p(pre + '/projection_bias'))

    def _attn_drop(self, name, shape, p=0.1):
        """tfa-MHA dropout=0.1 on the attention coefficients (trajNet.py:33,71,195) when training."""
        if self._dctx is None:
            return None
        return (p, self._dctx.snap, self._dctx.site(name, shape, p))

    def _drop(self, x, name, p=0.1):
        """tf.keras.layers.Dropout(0.1) (trajNet.py:75,77,209,211) when training."""
        return x if self._dctx is None else ops.dropout(x, p, self._dctx, name)

    def _cross_attention(self, pre, query, key, H, qvalid, kvalid):
        """Cross_Attention / Cross_AttentionT .call (trajNet.py:79-87,224-234), eval, sep_actors off."""
        v = self._tfa_mha(pre + '/mha', query, key, H, qvalid, kvalid)
        v = self._ln(v, pre + '/norm1', 1e-3)
        v = self._drop(self._dense(v, pre + '/FFN1', act=ACT_ELU), pre + '/dropout1')
        v = self._drop(self._dense(v, pre + '/FFN2'), pre + '/dropout2')
        return self._ln(v, pre + '/norm2', 1e-3)

    def agent_encode(self, obs, occ, cast=True):
        """The agent branch alone (TrajNet.call, trajNet.py:125-187), eval semantics: obs [B,48,11,8], occ [B,16,11,8] -> (key [B,64,384],
        mask [B,64]) as call() computes them.  For callers that run the branch apart from the raster path (set `agent_override` to the
        result before call()): it depends on the agents' tracks and the weights only."""
        ops.set_serial(self.serial)
        ops.use_arena(self._arena)
        if cast:                       # (cast=False: the caller knows the compute copy is current -- graph.GraphedForward's pipeline)
            self._sync_compute_weights()
        self._agent_pack_stale = True
        self._dctx = None
        return tuple(self._traj_net(obs, occ))

    def _agent_ws(self):
        """the agent branch's Params under the names ops.agent_pack / agent_enc use"""
        e, c = 'traj_net/traj_encoder', 'traj_net/cross_attention'
        p = self._p
        return {'wn': p(e + '/node_feature/kernel'), 'bn': p(e + '/node_feature/bias'), 'wv3': p(e + '/vector_feature/kernel'),
                'e_wq': p(e + '/node_attention/query_kernel'), 'e_wk': p(e + '/node_attention/key_kernel'),
                'e_wv': p(e + '/node_attention/value_kernel'), 'e_wo': p(e + '/node_attention/projection_kernel'),
                'e_bo': p(e + '/node_attention/projection_bias'), 'e_ws': p(e + '/sublayer/kernel'), 'e_bs': p(e + '/sublayer/bias'),
                'i_wq': p(c + '/mha/query_kernel'), 'i_wk': p(c + '/mha/key_kernel'), 'i_wv': p(c + '/mha/value_kernel'),
                'i_wo': p(c + '/mha/projection_kernel'), 'i_w1': p(c + '/FFN1/kernel'), 'i_w2': p(c + '/FFN2/kernel'),
                'i_bo': p(c + '/mha/projection_bias'), 'g1': p(c + '/norm1/gamma'), 'be1': p(c + '/norm1/beta'), 'b1': p(c + '/FFN1/bias'),
                'b2': p(c + '/FFN2/bias'), 'g2': p(c + '/norm2/gamma'), 'be2': p(c + '/norm2/beta'), 'seg': p('traj_net/seg_embed/kernel'),
                'g_obs': p('traj_net/obs_norm/gamma'), 'b_obs': p('traj_net/obs_norm/beta'), 'g_occ': p('traj_net/occ_norm/gamma'),
                'b_occ': p('traj_net/occ_norm/beta')}

    def _traj_net(self, obs, occ):
        """TrajNet.call (trajNet.py:125-187) with the 64-way TrajEncoder loop batched.  -> key [B,64,384], mask [B,64]."""
        pre = 'traj_net/traj_encoder'
        B, n_obs, T = obs.shape[0], obs.shape[1], obs.shape[2]
        A = n_obs + occ.shape[1]
        # tr = cat(obs, occ) [B,64,11,8]; step valid = tr[...,0] != 0 (trajNet.py:127,131); agent valid = any step (trajNet.py:138):
        # one launch writes the two feature slices in the activation dtype and the masks
        if self.fused_agent and ops.agent_enc_ok(n_obs, occ.shape[1], T, self.dtype):
            # the whole TrajEncoder (trajNet.py:38-48) for every agent: ONE launch (csrc/agent_fused.hip), straight from the raw tracks
            if self._agent_pack_stale:
                self._agent_pack = ops.agent_pack(self._agent_ws(), self.dtype, out=self._agent_pack)
                self._agent_pack_stale = False
            drop_e = self._attn_drop(pre + '/node_attention/dropout', (B * A, 4, T, T))
            if self.fused_agent_int and ops.agent_int_ok(n_obs, occ.shape[1], self.dtype):
                # ... followed by the 64-agent interaction block (trajNet.py:135-187) as three more launches: 16-bit storage types
                drop = None
                if self._dctx is not None:
                    c, d = 'traj_net/cross_attention', self._dctx
                    drop = (0.1, d.snap, (drop_e[2], d.site(c + '/mha/dropout', (B, 6, A, A), 0.1), d.site(c + '/dropout1', (B, A, 1536), 0.1),
                                          d.site(c + '/dropout2', (B, A, 384), 0.1)))
                return ops.agent_branch(obs, occ, self._agent_ws(), self._agent_pack, self.dtype, drop)
            enc, cmi = ops.agent_enc(obs, occ, self._agent_ws(), self._agent_pack, self.dtype, drop_e)
            cmf = cmi.to(self.dtype)
        else:
            x5, v3, vt, cmi, cmf = ops.agent_prep(obs, occ, self.dtype)
            nodes = ops.linear(x5, self._p(pre + '/node_feature/kernel'), self._p(pre + '/node_feature/bias'), act=ACT_ELU)   # Conv1D(64,1)+ELU
            nodes = nodes.view(B * A, T, 64)
            nodes = self._tfa_mha(pre + '/node_attention', nodes, nodes, 4, vt, vt)           # [B*A,T,320]
            nodes = ops.maxpool_time(nodes).view(B, A, 320)
            vec = ops.linear(v3.view(B, A, 3), self._p(pre + '/vector_feature/kernel'))
            enc = ops.linear(torch.cat([nodes, vec], -1), self._p(pre + '/sublayer/kernel'), self._p(pre + '/sublayer/bias'), act=ACT_ELU)
        onehot = self._seg_onehot.get((A, n_obs))
        if onehot is None:
            onehot = torch.zeros((A, 2), dtype=self.dtype, device=self.device)
            onehot[:n_obs, 0] = 1
            onehot[n_obs:, 1] = 1
            self._seg_onehot[(A, n_obs)] = onehot
        embed = ops.linear(onehot, self._p('traj_net/seg_embed/kernel'))                  # [64,384]
        concat, q_in = ops.agent_mix(enc, embed, cmf)                                     # enc * mask ; + embed
        value = self._cross_attention('traj_net/cross_attention', q_in, concat, 6, cmi, cmi)
        if enc.shape[-1] == 384:          # sum + obs_norm | occ_norm in one launch (the cross-attention waits on the end of this chain)
            key = ops.agent_out(enc, value, embed, self._p('traj_net/obs_norm/gamma'), self._p('traj_net/obs_norm/beta'),
                                self._p('traj_net/occ_norm/gamma'), self._p('traj_net/occ_norm/beta'), n_obs, 1e-3)
            return key, cmi
        out = ops.agent_sum(enc, value, embed)
        o1 = self._ln(out[:, :n_obs].contiguous(), 'traj_net/obs_norm', 1e-3)
        o2 = self._ln(out[:, n_obs:].contiguous(), 'traj_net/occ_norm', 1e-3)
        return torch.cat([o1, o2], 1), cmi

    def _resconv(self, skip, name):
        """ELU(Conv3D(8,1,1) SAME (tf.repeat(skip, 8))) collapsed exactly to 8 per-time 1x1 GEMMs with summed
        time taps (modules.py:750-765): W_t = sum_{j=max(0,3-t)}^{min(7,10-t)} W[j]."""
        pw, pb = self._p(name + '/kernel'), self._p(name + '/bias')
        Ci, Co = pw.shape[3], pw.shape[4]
        wz = torch.empty((8, Ci, Co), dtype=self.dtype, device=self.device)
        ops.call('stj_time_collapse', ops._p(pw.master), ops._p(wz), Ci * Co, ops.DTYPE_CODE[self.dtype], ops._st())
        gwz = ops.zeros_f32((8, Ci, Co), self.device)

        def fold():
            ops.call('stj_time_fold', ops._p(gwz), ops._p(pw.grad), Ci * Co, ops._st())
        # (16-bit training: the decoder level that adds this skip returns its gradient times ELU' -- ops.upconv_add(skips_pre=True))
        return ops.linear_z(skip, pw.master, wz[0], Ci * Co, pb.master.detach(), 0, gwz[0], Ci * Co, pb.grad, 8,
                            act=ACT_ELU, shared_x=True, fold=fold, grad_is_pre=ops.skips_pre_ok(self.dtype))    # [8, B*HW, Co]  (time-major)

    # ---- the 8 time-separated cross-attentions, batched over the waypoint axis z (trajNet.py:305-314) ----
    def _zp(self, suffix):
        return self._p('cross_attn_obs0/' + suffix)

    def _xattn_params(self):
        z = self._zp
        return {'wq': z('mha/query_kernel'), 'wo': z('mha/projection_kernel'), 'bo': z('mha/projection_bias'), 'g1': z('norm1/gamma'),
                'be1': z('norm1/beta'), 'w1': z('FFN1/kernel'), 'b1': z('FFN1/bias'), 'w2': z('FFN2/kernel'), 'b2': z('FFN2/bias'),
                'g2': z('norm2/gamma'), 'be2': z('norm2/beta')}

    def _pack_xattn(self):
        """The 8 sets' weights as the LDS-image stream the fused kernels stage (depends on the weights only: once per step)."""
        self._xattn_pack = ops.xattn_pack(self._xattn_params(), self._zstride, 8, self.dtype, out=self._xattn_pack)
        self._xattn_pack_stale = False

    def _xattn_kv(self, key):
        """key [B,64,Cb] -> (k, v) [8, B*64, 126]: the 8 sets' tfa key / value projections (kernels [3,384,42], addressed in place), one grouped launch."""
        zs = self._zstride
        with ops.gemm_group():
            pk, pv = self._zp('mha/key_kernel'), self._zp('mha/value_kernel')
            k = ops.linear_heads_in_z(key, pk.master, pk.c, pk.grad, zs, 8, True)
            v = ops.linear_heads_in_z(key, pv.master, pv.c, pv.grad, zs, 8, True)
        return k, v

    def _cross_attention_z(self, query, key, tmask):
        """8 x Cross_AttentionT (trajNet.py:224-234) + query residual in one batched pass, waypoint-major:
        query [8,B,HW,Cb], key [B,64,Cb] -> [8,B,HW,Cb]."""
        Z, B, HW, Cb = query.shape
        zs = self._zstride
        hs = 128 // 3
        A = key.shape[1]

        def proj_in(x, suffix, shared):           # tfa kernels [3, 384, 42] of the 8 sets, addressed in place (set stride zs)
            p0 = self._zp(suffix)
            return ops.linear_heads_in_z(x, p0.master, p0.c, p0.grad, zs, 8, shared)
        if self.fused_xattn and A == 64 and HW % 64 == 0 and Cb == 384:
            # ONE kernel: q projection, masked softmax attention, out projection, LN, FFN, LN, + query (csrc/xattn_fused.hip);
            # only the projections of the 64 agent keys / values stay GEMMs (one grouped launch)
            pre, self._xattn_kv_pre = self._xattn_kv_pre, None
            k, v = pre if pre is not None else self._xattn_kv(key)
            ps = self._xattn_params()
            if self._xattn_pack_stale:
                self._pack_xattn()
            return ops.xattn(query, k, v, tmask, self._xattn_pack, ps, zs, self._dctx,
                             ('cross_attn_obs/mha/dropout', 'cross_attn_obs/dropout1', 'cross_attn_obs/dropout2'), defer_wg=pre is not None)
        with ops.gemm_group():
            q = proj_in(query, 'mha/query_kernel', False)                    # [8, B*HW, 126]
            k = proj_in(key, 'mha/key_kernel', True)                         # [8, B*64, 126]
            v = proj_in(key, 'mha/value_kernel', True)
        kvalid = tmask[None].expand(Z, B, A).reshape(Z * B, A).contiguous()
        o = ops.mha_core(q.view(Z * B, HW, 3 * hs), k.view(Z * B, A, 3 * hs), v.view(Z * B, A, 3 * hs), 3, hs,
                         1.0 / math.sqrt(hs), kvalid=kvalid, drop=self._attn_drop('cross_attn_obs/mha/dropout', (Z, B, 3, HW, A)))
        pw, pb = self._zp('mha/projection_kernel'), self._zp('mha/projection_bias')
        H_, hs_, O_ = pw.shape
        v1 = ops.linear_z(o.view(Z, B * HW, 3 * hs), pw.master, pw.c.view(H_ * hs_, O_), zs, pb.master.detach(), zs,
                          pw.grad.view(H_ * hs_, O_), zs, pb.grad, 8)
        v1 = ops.layernorm(v1, self._zp('norm1/gamma'), self._zp('norm1/beta'), 1e-3, group_rows=B * HW, ngroups=8, gstride=zs)
        pw, pb = self._zp('FFN1/kernel'), self._zp('FFN1/bias')
        v1 = self._drop(ops.linear_z(v1, pw.master, pw.c, zs, pb.master.detach(), zs, pw.grad, zs, pb.grad, 8, act=ACT_ELU),
                        'cross_attn_obs/dropout1')                                   # draws laid out [8, B*HW, 512]
        pw, pb = self._zp('FFN2/kernel'), self._zp('FFN2/bias')
        v1 = self._drop(ops.linear_z(v1, pw.master, pw.c, zs, pb.master.detach(), zs, pw.grad, zs, pb.grad, 8),
                        'cross_attn_obs/dropout2')
        v1 = ops.layernorm(v1, self._zp('norm2/gamma'), self._zp('norm2/beta'), 1e-3, group_rows=B * HW, ngroups=8, gstride=zs,
                           res=query.reshape(v1.shape))                             # + query (trajNet.py:317)
        return v1.view(Z, B, HW, Cb)

    def _decoder(self, x, res_list, B, skips=None):
        """Pyramid3DDecoder.call (modules.py:739-772): shallow_decode=1, flow_sep_decode, use_pyramid, rep_res."""
        hb = self.hb
        flow_res, r0, r1 = res_list[0], res_list[1], res_list[2]

        def up(t, name, grad_is_pre=False, x_is_elu_out=False):
            return ops.upconv(t, self._p(name + '/kernel'), self._p(name + '/bias'), grad_is_pre, x_is_elu_out,
                              prep=self._upconv_prep.get(name))
        x = x.view(8 * B, hb, hb, -1)                                                # frames are TIME-major: f = t*B + b
        if self._prep_event is not None:                                             # folded kernels come from the side stream
            torch.cuda.current_stream(self.device).wait_event(self._prep_event)
        if skips is not None:                                                        # computed on the side stream: join
            main = torch.cuda.current_stream(self.device)
            main.wait_stream(self._side2)
            for t in skips:
                t.record_stream(main)
            s3, s2, sf = skips
        else:
            s3, s2, sf = (self._resconv(r1, 'decoder/resconv_3'), self._resconv(r0, 'decoder/resconv_2'),
                          self._resconv(flow_res, 'decoder/resconv_f'))

        def up_add(t, name, ra, rb=None):        # up-conv with the skip sum(s) in its epilogue (modules.py:750-765)
            return ops.upconv_add(t, self._p(name + '/kernel'), self._p(name + '/bias'), ra, rb, prep=self._upconv_prep.get(name),
                                  skips_pre=ops.skips_pre_ok(self.dtype))
        x = up_add(x, 'decoder/upconv_3_0', s3)                                      # [F,2hb,2hb,192]
        self._tap('decoder/level3', x)
        x, fx = up_add(x, 'decoder/upconv_2_0', s2, sf)                              # [F,4hb,4hb,128] x 2
        self._tap('decoder/level2', x)
        self._tap('decoder/level2_flow', fx)
        oc = ('decoder/outconv/kernel', 'decoder/outconv/bias', 'decoder/outconv_f/kernel', 'decoder/outconv_f/bias')
        if ops.upconv_head_ok(2 * x.shape[1], 2 * x.shape[2], self._p('decoder/upconv_0_0/kernel'), x.dtype, 8) and self.taps is None:
            # inference: the last level's output never exists -- its epilogue projects it onto the heads' weights, the heads are a 9-neighbour sum
            def last(t, n1, n0, head):
                t = up(t, n1)
                return ops.upconv_head(t, self._p(n0 + '/kernel'), self._p(n0 + '/bias'), self._p(head), prep=self._upconv_prep.get(n0))
            # both branches on ONE stream: their persistent one-workgroup-per-CU kernels time-slice the CUs anyway, and the two cross-stream
            # joins of the two-stream form cost more than its overlap gave (B = 32 fp16, alternating same-box runs in both orders:
            # 5555 / 5539 / 5341 / 5422 vs 5482 / 5317 / 5320 / 5337 and 5338 / 5452 / 5459 vs 5274 / 5378 / 5444 scenes/s)
            zo = last(x, 'decoder/upconv_1_0', 'decoder/upconv_0_0', oc[0])
            zf = last(fx, 'decoder/upconvf_1_0', 'decoder/upconvf_0_0', oc[2])
            return ops.heads_gather(zo, zf, self._p(oc[1]), self._p(oc[3]), B, 8, t_major=True)
        # the last two levels of each branch have a single consumer, whose backward folds ELU' into the gradient it returns
        if self._side2 is not None:      # the observed-occupancy and flow branches of the last two levels are independent
            main = torch.cuda.current_stream(self.device)
            self._side2.wait_stream(main)
            fx.record_stream(self._side2)
            with torch.cuda.stream(self._side2):
                fx = up(up(fx, 'decoder/upconvf_1_0', grad_is_pre=True), 'decoder/upconvf_0_0', grad_is_pre=True, x_is_elu_out=True)
            x = up(up(x, 'decoder/upconv_1_0', grad_is_pre=True), 'decoder/upconv_0_0', grad_is_pre=True, x_is_elu_out=True)
            main.wait_stream(self._side2)
            fx.record_stream(main)
        else:
            x = up(up(x, 'decoder/upconv_1_0', grad_is_pre=True), 'decoder/upconv_0_0', grad_is_pre=True, x_is_elu_out=True)
            fx = up(up(fx, 'decoder/upconvf_1_0', grad_is_pre=True), 'decoder/upconvf_0_0', grad_is_pre=True, x_is_elu_out=True)
        self._tap('decoder/level0', x)
        self._tap('decoder/level0_flow', fx)
        return ops.outconv_pair(x, fx, self._p('decoder/outconv/kernel'), self._p('decoder/outconv/bias'),
                                self._p('decoder/outconv_f/kernel'), self._p('decoder/outconv_f/bias'), B, 8, t_major=True, x_is_elu_out=True,
                                side=self._side2)

    # ------------------------------------------------------------------ call
    def __call__(self, ogm, map_img, training=True, obs=None, occ=None, mapt=None, flow=None, dense_vec=None, dense_map=None):
        return self.call(ogm, map_img, training, obs, occ, mapt, flow, dense_vec, dense_map)

    def call(self, ogm, map_img, training=True, obs=None, occ=None, mapt=None, flow=None, dense_vec=None, dense_map=None):
        """STrajNet.call (modules.py:815-839).  mapt is ignored (actor_only=True), as in the reference."""
        if obs is None or occ is None or flow is None:
            raise ValueError('obs, occ and flow are required (reference modules.py:818,834)')
        H = self.cfg['input_size'][0]
        if ogm.dim() != 5 or tuple(ogm.shape[1:]) != (H, H, 11, 2):
            raise ValueError(f'ogm must be [B,{H},{H},11,2], got {tuple(ogm.shape)}')
        B = ogm.shape[0]
        if tuple(map_img.shape) != (B, self.map_size, self.map_size, 3):
            raise ValueError(f'map_img must be [B,{self.map_size},{self.map_size},3], got {tuple(map_img.shape)}')
        if tuple(flow.shape) != (B, H, H, 2):
            raise ValueError(f'flow must be [B,{H},{H},2], got {tuple(flow.shape)}')
        if tuple(obs.shape[2:]) != (11, 8) or tuple(occ.shape[2:]) != (11, 8):
            raise ValueError('obs/occ must be [B,n,11,8]')
        for t in (ogm, map_img, flow, obs, occ):
            if not t.is_cuda:
                raise RuntimeError('inputs must be CUDA (ROCm) tensors: the HIP path has no CPU fallback')
        self._side, self._side2 = (None, None) if self.serial else self._streams
        ops.set_serial(self.serial)
        ops.wgrad_queue_reset(self.device)
        ops.use_arena(self._arena)
        self._sync_compute_weights()
        self._dctx = None
        if training:                     # Dropout / DropPath draws of this step (reference: training=True, train.py:218)
            self.dropctx.begin()
            self._dctx = self.dropctx
        ogm, map_img, flow = ogm.float().contiguous(), map_img.float().contiguous(), flow.float().contiguous()
        hb, Cb = self.hb, self.stage_dim[2]
        # fold the six decoder kernels into their 2x2-tap phase matrices NOW (they depend on the weights only): issued where they are
        # consumed, each of these tiny launches sat on the critical path of a decoder branch behind whatever shared the GPU with it (up
        # to 0.8 ms apiece in the batch-32 inference trace).  They go to the second side stream: as the first launches of the main
        # stream, the chain of six delayed the first encoder kernel by ~50 us.
        names = ('decoder/upconv_3_0', 'decoder/upconv_2_0', 'decoder/upconv_1_0', 'decoder/upconv_0_0', 'decoder/upconvf_1_0',
                 'decoder/upconvf_0_0')
        self._prep_event = None
        self._xattn_pack_stale = True
        self._agent_pack_stale = True
        self._agent_pack_event = None
        self._fgoff_pack_stale = True
        self._fgoff_pack_event = None

        def issue_prep():
            if self._side2 is not None:
                self._side2.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._side2):
                    if self.fused_fgoff and self.fg_msa and ops.fgoff_ok(self.dtype, hb, hb, Cb, 8):
                        self._fgoff_pack = ops.fgoff_pack(self._p('fg_msa/conv_offset_0/kernel'), self.dtype, out=self._fgoff_pack)
                        self._fgoff_pack_stale = False
                        self._fgoff_pack_event = torch.cuda.Event()
                        self._fgoff_pack_event.record(self._side2)
                    if self.fused_agent and self._agent_pack_stale:      # the agent branch's weight pack: with the other preparations, not on its chain
                        self._agent_pack = ops.agent_pack(self._agent_ws(), self.dtype, out=self._agent_pack)
                        self._agent_pack_stale = False
                        self._agent_pack_event = torch.cuda.Event()
                        self._agent_pack_event.record(self._side2)
                    if self.fused_xattn:
                        self._pack_xattn()
                    self._upconv_prep = {n: ops.upconv_prep(self._p(n + '/kernel'), self.dtype) for n in names}
                    self._prep_event = torch.cuda.Event()
                    self._prep_event.record(self._side2)
            else:
                if self.fused_xattn:
                    self._pack_xattn()
                self._upconv_prep = {n: ops.upconv_prep(self._p(n + '/kernel'), self.dtype) for n in names}
        # The agent branch (trajNet: a dependent chain of ~45 small launches that occupy a few CUs each, ~0.4 ms end to end) is
        # independent of the raster encoder up to the cross-attention: it runs on a side stream, forked HERE.  Its launches are ISSUED
        # after the encoder's first stage though: a replayed hipGraph starts branches roughly in node-creation order, and issued first
        # the chain ran alone on an idle GPU for 0.5 ms before the first Swin kernel started (profiles/r02_c_timeline_concurrent.txt).
        mode = self.agent_issue_mode if self._side is not None else -1       # (0: issued at the head of the step, 1: after the encoder -- both measured equal or worse)
        main_pos = None
        if mode in (3, 4, 5):           # on the MAIN stream: 4 = at the head of the step, 3 = behind the encoder, 5 = behind FG-MSA (in front of the cross-attention)
            main_pos, mode = mode, -3
        agent = []
        if self.agent_override is not None:     # the caller ran agent_encode() itself (graph.GraphedForward: a graph of its own, a batch ahead)
            agent.extend(self.agent_override)
            mode = -2

        self._xattn_kv_pre = None

        def issue_agent():
            self._side.wait_event(fork)
            if self._agent_pack_event is not None:
                self._side.wait_event(self._agent_pack_event)
            with torch.cuda.stream(self._side):
                agent.extend(self._traj_net(obs, occ))
                if self.kv_in_agent_branch and self.fused_xattn and agent[0].shape[1] == 64 and Cb == 384:
                    # the 8 sets' key / value projections of the agent encoding belong to the cross-attention (trajNet.py:225) but depend on
                    # the agent branch only: issued HERE, on its stream, their backward (two grouped launches, ~80 us in the step) runs beside
                    # the FG-MSA backward instead of in front of it on the main stream -- and autograd, which runs the most recently
                    # created nodes first, reaches FG-MSA's nodes before these
                    self._xattn_kv_pre = self._xattn_kv(agent[0])
        if mode >= 0:
            main = torch.cuda.current_stream(self.device)
            fork = torch.cuda.Event()
            fork.record(main)
        if mode == 0:
            issue_agent()
        elif mode == -1 or main_pos == 4:
            agent.extend(self._traj_net(obs, occ))
        # Side work that is not needed before the cross-attention / the decoder / the loss is ISSUED behind the encoder's first stage
        # too: a replayed hipGraph starts its first ~20 nodes one after the other whatever their stream, so the seven packing /
        # folding launches and the caller's mid_forward_hook (graph.py: the ground-truth half of the loss) at the head of the step kept
        # the first encoder kernel waiting until 142 us (profiles/r04_b_timeline_concurrent.txt)
        late = mode == 2 and self._side2 is not None
        if not late:
            issue_prep()

        def hook():
            if late:
                issue_prep()
            if mode == 2:
                issue_agent()
            if self.mid_forward_hook is not None:
                self.mid_forward_hook()
        res_list = self._encoder(ogm, map_img, flow, hook=hook)
        if mode == 1:
            issue_agent()
        if main_pos == 3:
            agent.extend(self._traj_net(obs, occ))
        fold = self._fold_partials
        if self.cut_encoder and torch.is_grad_enabled():
            # data-parallel overlap: detach here; backward() then ends at these leaves (the tail bucket is complete and can be
            # all-reduced) and backward_encoder() resumes from their .grad
            self._cut_src = res_list
            res_list = [t.detach().requires_grad_(True) for t in res_list]
            self._cut_leaf = res_list
            fold = lambda: self._fold_partials('tail')
        # the three time-kernel skips (Conv3D collapsed to per-waypoint 1x1 GEMMs) only need the encoder outputs: side stream,
        # overlapping FG-MSA / the cross-attentions / the first up-convs; joined in the decoder where they are added
        skips = None

        def issue_skips():
            main2 = torch.cuda.current_stream(self.device)
            self._side2.wait_stream(main2)
            for t in res_list[:3]:
                t.record_stream(self._side2)
            with torch.cuda.stream(self._side2):
                return (self._resconv(res_list[2], 'decoder/resconv_3'), self._resconv(res_list[1], 'decoder/resconv_2'),
                        self._resconv(res_list[0], 'decoder/resconv_f'))
        if self._side2 is not None and self.skips_issue == 0:
            skips = issue_skips()
        q = ops.wgrad_queue_flush_point(res_list[-1]).reshape(B, hb, hb, Cb)     # backward: FG-MSA / cross-attention / agent branch are through
        # waypoint-major [8,B,HW,Cb] (the reference's [B,8,...] transposed): every per-waypoint product downstream is then a
        # plain batched GEMM and the decoder frames are t-major; the output kernel undoes it when writing [B,H,W,32]
        if self.fg_msa:
            q, query = self._fgmsa(q)                                              # modules.py:825-831
        else:
            query = q.reshape(1, B, hb * hb, Cb).expand(8, B, hb * hb, Cb).contiguous()   # modules.py:827
        if self._side2 is not None and self.skips_issue == 1:       # behind FG-MSA: beside the cross-attention
            skips = issue_skips()
        if main_pos == 5:
            agent.extend(self._traj_net(obs, occ))
        key, tmask = agent
        if self._side is not None and mode >= 0:       # join the agent branch
            main.wait_stream(self._side)
            key.record_stream(main)
            tmask.record_stream(main)
            if self._xattn_kv_pre is not None:
                for t in self._xattn_kv_pre:
                    t.record_stream(main)
        self._tap('agent_key', key)
        self._tap('query', query)
        if self._prep_event is not None:                 # the packed cross-attention weights come from the side stream
            torch.cuda.current_stream(self.device).wait_event(self._prep_event)
        x = self._cross_attention_z(query, key, tmask)               # [8,B,hb*hb,Cb]  (trajNet.py:305-317)
        self._tap('cross_attention_out', x)
        # the decoder's weight gradients are launched when ITS backward is through (ops.py).  (Flushing them only after the cross-attention
        # backward as well -- so that kernel has the GPU to itself -- measured 5 % SLOWER, 7.26 vs 6.88 ms: the 1.5 ms of half-GPU
        # weight-gradient launches then reach into the encoder's backward.)
        x = ops.wgrad_flush_point(x)
        if self._side2 is not None and self.skips_issue == 2:       # behind the cross-attention: beside the first up-conv
            skips = issue_skips()
        try:
            out = self._decoder(x, res_list, B, skips)
        finally:
            ops.wgrad_defer_end()        # (also when the decoder raises: a stale "defer" flag would swallow the next stand-alone up-conv's weight gradient)
        self._tap('output', out)
        return ops.join_after_backward(out, (self._side, self._side2), fold)
