"""Known-answer tests pinning the oracle's restatement of third-party semantics (SURVEY.md App. C KATs i-vi).
The reference ships no tests; these are hand-derivable facts about the ops it calls."""
import numpy as np
import torch

from oracle import np_ref as R
from oracle import torch_ref as T


def test_relative_position_index_ws2():
    """(i) modules.py:88-98 for a 2x2 window, by hand: tokens (0,0),(0,1),(1,0),(1,1); index = (dy+1)*3 + (dx+1)."""
    idx = R.relative_position_index(2)
    exp = np.array([[4, 3, 1, 0], [5, 4, 2, 1], [7, 6, 4, 3], [8, 7, 5, 4]])
    assert (idx == exp).all()


def test_shift_mask_region_counts():
    """(ii) H=W=16, ws=8, shift=4 (modules.py:189-216): 4 windows; window 0 is unmasked, windows on the wrapped edge
    split 32/32, the corner window into four 16-token regions."""
    m = R.shift_attn_mask(16, 16, 8, 4)
    assert m.shape == (4, 64, 64)
    assert set(np.unique(m)) == {0.0, -100.0}
    allowed = (m == 0).sum(-1)                # per token: how many keys it may attend to
    assert (allowed[0] == 64).all()
    assert (allowed[1] == 32).all() and (allowed[2] == 32).all()
    assert (allowed[3] == 16).all()
    assert (m == m.transpose(0, 2, 1)).all()


def test_bilinear_sampler_cases():
    """(iii) occu_metric.sample / tfa interpolate_bilinear with zero border: integer coords return the pixel, -0.5
    returns half the border pixel, beyond -1 returns 0, far positive clamps into the zero border."""
    img = np.arange(1, 13, dtype=np.float64).reshape(1, 3, 4, 1)       # H=3, W=4
    def s(x, y):
        return float(R.sample(img, np.array([[[x, y]]], np.float64))[0, 0, 0])
    assert s(2, 1) == img[0, 1, 2, 0]
    assert s(0, 0) == 1.0
    assert s(-0.5, 0) == 0.5 * img[0, 0, 0, 0]
    assert s(0, -0.5) == 0.5 * img[0, 0, 0, 0]
    assert s(-1.0, 0) == 0.0 and s(-3.0, 1) == 0.0
    assert s(100.0, 1) == 0.0 and s(1, 100.0) == 0.0
    assert s(3.5, 2) == 0.5 * img[0, 2, 3, 0]
    assert abs(s(1.25, 0.5) - (0.5 * (0.75 * 2 + 0.25 * 3) + 0.5 * (0.75 * 6 + 0.25 * 7))) < 1e-12
    # the independent formulation (grid_sample) agrees on random coordinates, including far outside
    rng = np.random.default_rng(0)
    im = rng.normal(size=(2, 7, 9, 3))
    wp = rng.uniform(-4, 13, size=(2, 50, 2))
    a = R.sample(im, wp)
    b = T._sample(torch.as_tensor(im), torch.as_tensor(wp)).numpy()
    assert np.abs(a - b).max() < 1e-12


def test_conv3d_time_collapse():
    """(iv) Conv3D(8,1,1) SAME on a time-repeated input == 8 per-t 1x1 GEMMs with summed taps (App. C-4/5)."""
    rng = np.random.default_rng(1)
    skip = rng.normal(size=(2, 5, 5, 6))
    W = rng.normal(size=(8, 1, 1, 6, 4))
    b = rng.normal(size=4)
    direct = R.conv3d_time_same(np.repeat(skip[:, None], 8, 1), W, b)
    for t in range(8):
        Wt = W[max(0, 3 - t):min(7, 10 - t) + 1, 0, 0].sum(0)
        assert np.abs(direct[:, t] - (skip @ Wt + b)).max() < 1e-12
    # pad asymmetry: out[0] must not see W[0..2], out[7] must not see W[4..7]... (3 before, 4 after)
    assert max(0, 3 - 0) == 3 and min(7, 10 - 7) == 3


def test_keras_auc_toys():
    """(v) Keras AUC(PR, 100 thresholds, interpolation): perfectly separated -> 1; no positives -> 0; closed-form toy."""
    yt = np.array([0, 0, 1, 1.0])
    assert abs(R.keras_auc_pr(yt, np.array([0.0, 0.0, 1.0, 1.0])) - 1.0) < 1e-6
    assert R.keras_auc_pr(np.zeros(4), np.array([0.1, 0.5, 0.2, 0.9])) == 0.0
    # all predictions identical 0.5: tp=2, fp=2 up to thr<0.5 then 0 -> one step where precision is interpolated
    v = R.keras_auc_pr(yt, np.full(4, 0.5))
    # thresholds below 0.5: precision 0.5 recall 1 ; above: tp=fp=0.  Interpolation on the single drop:
    # slope = dtp/dp = 2/4, intercept = 0, increment = slope*(dtp + 0*log(ratio=1))/ (tp+fn) = 0.5*2/2 = 0.5
    assert abs(v - 0.5) < 1e-9
    # the bucketised restatement agrees on random data
    rng = np.random.default_rng(2)
    t = (rng.random(5000) < 0.1).astype(np.float64)
    p = np.clip(rng.random(5000) * 0.7 + 0.3 * t, 0, 1)
    assert abs(R.keras_auc_pr(t, p) - T.auc_pr_bucketised(torch.as_tensor(t), torch.as_tensor(p))) < 1e-12


def test_upsample_fold_identity():
    """(vi) nearest-2x then 3x3 SAME == four 2x2 phase convs on the low-res input with summed taps (the algebra the
    HIP decoder kernels rely on, strajnet_amd/csrc/conv.hip)."""
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 5, 6, 3))
    W = rng.normal(size=(3, 3, 3, 2))
    ref = R.conv2d_same(R.upsample2(x), W)
    R_ = {(0, 0): [0], (0, 1): [1, 2], (1, 0): [0, 1], (1, 1): [2]}
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    out = np.zeros_like(ref)
    for a in range(2):
        for b in range(2):
            for r in range(2):
                for s in range(2):
                    Weff = sum(W[y, xx] for y in R_[(a, r)] for xx in R_[(b, s)])
                    # X[i+a-1+r, j+b-1+s] -> padded index i+a+r, j+b+s
                    out[:, a::2, b::2] += xp[:, a + r:a + r + 5, b + s:b + s + 6] @ Weff
    assert np.abs(out - ref).max() < 1e-12


def test_tfa_mask_semantics():
    """tfa MHA additive mask -10e9*(1-mask) in f32: fully masked rows become uniform, partially masked keys vanish."""
    rng = np.random.default_rng(4)
    q = rng.normal(size=(1, 3, 8))
    Wq = rng.normal(size=(2, 8, 4)); Wo = rng.normal(size=(2, 4, 5)); bo = np.zeros(5)
    mask = np.array([[[1, 1, 0], [0, 0, 0], [1, 0, 0]]])
    out = R.tfa_mha(q, q, q, Wq, Wq, Wq, Wo, bo, mask=mask)
    v = np.einsum('...mi,hio->...mho', q, Wq)
    uniform = np.einsum('mhi,hio->o', v[0], Wo) / 3.0
    assert np.abs(out[0, 1] - uniform).max() < 1e-12              # fully masked row -> mean of all values
    only0 = np.einsum('hi,hio->o', v[0, 0], Wo)
    assert np.abs(out[0, 2] - only0).max() < 1e-12                # single allowed key


def test_param_count_matches_survey():
    cfg = dict(input_size=(256, 256), window_size=8, embed_dim=96, depths=[2, 2, 2], num_heads=[3, 6, 12])
    g = R.geometry(cfg)
    n = 0
    for name, (shape, kind) in R.param_spec(cfg).items():
        if kind == 'fg_rpe':
            shape = (2 * g['hb'] - 1, 2 * g['hb'] - 1, 8)
        n += int(np.prod(shape))
    assert n == 13277788
    n2 = sum(int(np.prod(s)) for s, k in R.param_spec(cfg, fg_msa=False, fg=False).values())
    assert n2 == 12510452


def test_metrics_kats():
    """occu_metric.py: soft IoU of identical binary maps is 1 and 0 for empty truth (divide_no_nan); EPE counts only cells with
    ground-truth flow; the packed metric function reproduces the hand computation on a 2x2 toy and obeys no_warp."""
    from oracle import np_ref as R
    a = np.array([[1.0, 0.0], [0.0, 1.0]])
    assert abs(R.soft_iou(a, a) - 1.0) < 1e-12
    assert R.soft_iou(np.zeros((2, 2)), np.zeros((2, 2))) == 0.0
    assert abs(R.soft_iou(a, 0.5 * np.ones((2, 2))) - (0.25 / (0.5 + 0.5 - 0.25))) < 1e-12
    tf_ = np.zeros((1, 2, 2, 2)); tf_[0, 0, 0] = (3.0, 4.0)
    pf = np.zeros((1, 2, 2, 2)); pf[0, 1, 1] = (100.0, 100.0)          # error where no GT flow exists does not count
    assert abs(R.flow_epe(tf_, pf) - 5.0) < 1e-12
    assert R.flow_epe(np.zeros((1, 2, 2, 2)), pf) == 0.0
    rng = np.random.default_rng(0)
    y = rng.normal(size=(1, 8, 8, 32))
    go = (rng.random((1, 8, 8, 8, 1)) < 0.3).astype(np.float64); gc = (rng.random((1, 8, 8, 8, 1)) < 0.1).astype(np.float64)
    gf = rng.normal(size=(1, 8, 8, 8, 2)) * go; org = rng.random((1, 8, 8, 8, 1))
    m = R.occupancy_flow_metrics(y, go, gc, gf, org)
    m2 = R.occupancy_flow_metrics(y, go, gc, gf, org, no_warp=True)
    assert len(m) == 7 and all(np.isfinite(m)) and m2[5] == 0.0 and m2[6] == 0.0 and m[:5] == m2[:5]
    sig = 1 / (1 + np.exp(-y))
    ious = [R.soft_iou(go[:, k], sig[..., 4 * k:4 * k + 1]) for k in range(8)]
    assert abs(m[2] - np.mean(ious)) < 1e-12
    # probabilities in, same answer
    yp = y.copy(); yp[..., 0::4] = sig[..., 0::4]; yp[..., 1::4] = sig[..., 1::4]
    m3 = R.occupancy_flow_metrics(yp, go, gc, gf, org, pred_is_logits=False)
    assert np.allclose(m, m3, atol=1e-12)


def test_focal_and_keras_bce_known_answers():
    """tfa sigmoid_focal_crossentropy (alpha .25, gamma 2) and Keras backend BCE on probabilities, by hand."""
    from oracle import np_ref as R
    ln2 = np.log(2.0)
    # y=1, logit 0: ce = ln 2, p_t = .5, alpha_t = .25, (1-p_t)^2 = .25
    assert abs(float(R.focal_logits(np.array(1.0), np.array(0.0))) - 0.25 * 0.25 * ln2) < 1e-12
    # y=0, logit 0: alpha_t = .75
    assert abs(float(R.focal_logits(np.array(0.0), np.array(0.0))) - 0.75 * 0.25 * ln2) < 1e-12
    # confident and right -> modulating factor kills the term; confident and wrong -> ~alpha_t * ce
    assert float(R.focal_logits(np.array(1.0), np.array(20.0))) < 1e-20
    assert abs(float(R.focal_logits(np.array(1.0), np.array(-20.0))) - 0.25 * 20.0) < 1e-6
    eps = float(np.float32(1e-7))
    assert abs(float(R.bce_prob(np.array(1.0), np.array(0.5))) + np.log(0.5 + eps)) < 1e-15
    assert abs(float(R.bce_prob(np.array(1.0), np.array(0.0))) + np.log(2 * eps)) < 1e-12          # clipped to eps, + eps
    assert abs(float(R.bce_prob(np.array(0.0), np.array(0.0))) + np.log(1 - eps + eps)) < 1e-12    # == 0 up to rounding
    # probability form of the focal term: y=1, q=.5 -> .25 * .25 * bce
    assert abs(float(R.focal_prob(np.array(1.0), np.array(0.5))) - 0.0625 * float(R.bce_prob(np.array(1.0), np.array(0.5)))) < 1e-15


def test_published_vectors():
    """PUBLISHED third-party vectors (docstring examples of the libraries the reference calls) -- the only outside pins available in
    an image without TensorFlow; everything else in this file is hand-derived.  Each is quoted with its source."""
    from oracle import np_ref as R
    # (1) tensorflow_addons.losses.SigmoidFocalCrossEntropy docstring (alpha .25, gamma 2, probabilities in, reduction NONE):
    #     y_true [[1],[1],[0]], y_pred [[0.97],[0.91],[0.03]] -> [6.8532745e-06, 1.9097870e-04, 2.0559824e-05]   (loss.py:32-38 constructs it)
    y = np.array([1.0, 1.0, 0.0])
    q = np.array([0.97, 0.91, 0.03])
    pub = np.array([6.8532745e-06, 1.9097870e-04, 2.0559824e-05])
    got = R.focal_prob(y, q)
    assert np.abs(got / pub - 1).max() < 5e-6, got            # float64 evaluation vs figures published from a float32 evaluation ...
    got32 = R.focal_prob(y.astype(np.float32), q.astype(np.float32))
    assert got32.dtype == np.float32 and np.abs(got32 / pub - 1).max() < 2e-7, got32      # ... which the same formula in float32 reproduces digit for digit
    # (2) tf.keras.metrics.AUC docstring worked example (num_thresholds=3, y_true [0,0,1,1], y_pred [0,0.5,0.3,0.9]): thresholds
    #     [0 - 1e-7, 0.5, 1 + 1e-7], tp = [2,1,0], fp = [2,0,0], fn = [0,1,2], ROC AUC 0.75 -- pins the threshold construction and the
    #     strict `>` comparison that keras_auc_pr (loss.py:41,134-136) shares with it
    thr, tp, fp, fn = R.keras_auc_counts([0, 0, 1, 1], [0, 0.5, 0.3, 0.9], 3)
    assert np.allclose(thr, [0 - 1e-7, 0.5, 1 + 1e-7], atol=1e-12 + 1e-7) and thr[0] < 0 and thr[2] > 1
    assert tp.tolist() == [2, 1, 0] and fp.tolist() == [2, 0, 0] and fn.tolist() == [0, 1, 2]
    tn = 2 - fp
    recall, fpr = tp / (tp + fn), fp / (fp + tn)
    roc = sum((recall[i] + recall[i + 1]) / 2 * (fpr[i] - fpr[i + 1]) for i in range(2))
    assert abs(roc - 0.75) < 1e-12
    # (3) tf.keras.activations.gelu docstring, approximate=True (the tanh form the reference's own Gelu class restates, modules.py:18-29):
    #     x = [-3, -1, 0, 1, 3] -> [-0.00363752, -0.15880796, 0., 0.841192, 2.9963627]
    pubg = np.array([-0.00363752, -0.15880796, 0.0, 0.841192, 2.9963627])
    assert np.abs(R.gelu(np.array([-3.0, -1.0, 0.0, 1.0, 3.0])) - pubg).max() < 5e-7
    # (4) tf.keras.layers.LayerNormalization docstring: data = arange(10).reshape(5, 2) * 10, axis = 1 -> every row prints [-1., 1.]
    #     (biased variance over the last axis, eps 1e-3 inside the root: (0 - 5) / sqrt(25 + 1e-3) = -0.99998)
    d = np.arange(10, dtype=np.float64).reshape(5, 2) * 10
    ln = R.layer_norm(d, np.ones(2), np.zeros(2), 1e-3)
    assert np.abs(ln - np.array([[-1.0, 1.0]] * 5)).max() < 3e-5
    assert np.abs(ln[0, 0] + 5 / np.sqrt(25 + 1e-3)) < 1e-12
    # (5) tf.nn.sigmoid_cross_entropy_with_logits docstring: logits [1., -1., 0., 1., -1., 0., 0.], labels [0., 0., 0., 1., 1., 1., 0.5]
    #     -> [1.3132616, 0.3132617, 0.6931472, 0.3132617, 1.3132616, 0.6931472, 0.6931472]
    lg = np.array([1.0, -1.0, 0.0, 1.0, -1.0, 0.0, 0.0])
    lb = np.array([0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 0.5])
    pubx = np.array([1.3132616, 0.3132617, 0.6931472, 0.3132617, 1.3132616, 0.6931472, 0.6931472])
    assert np.abs(R.sigmoid_xe(lb, lg) - pubx).max() < 1e-7
