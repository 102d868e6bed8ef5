"""GPU parity of every HIP operator (forward AND backward) against an fp32 PyTorch reference of the same op, evaluated on the same
bf16-rounded inputs.  Tolerances: outputs are bf16 (8 significant bits) with fp32 accumulation, so max|err| <= 2e-2 * max|ref| for
activations and 3e-2 for gradients; index-like results (max-pool routing) are exact.  All calls go through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACT_TOL = 2e-2
GRAD_TOL = 3e-2


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import __graft_entry__ as g
    g.build()
    from multitask_hydranet_amd import ops
    torch.manual_seed(0)
    return ops


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0):
    """bf16-representable fp32 tensor on the GPU."""
    return (torch.randn(*shape, device=dev()) * scale).bfloat16().float()


def nhwc(x):
    """NCHW fp32 -> NHWC bf16 leaf."""
    return x.permute(0, 2, 3, 1).contiguous().bfloat16()


def nchw(x):
    return x.float().permute(0, 3, 1, 2)


def close(a, b, tol, name=""):
    a, b = a.float(), b.float()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = float((a - b).abs().max())
    ref = float(b.abs().max())
    assert err <= tol * ref + 1e-6, f"{name}: max err {err:.4e} vs ref max {ref:.4e} (tol {tol})"


def bfr(z):
    """round to bf16 with a straight-through gradient: the HIP path stores conv outputs in bf16 BEFORE BatchNorm, so the
    reference must see the same values (otherwise a ReLU mask flips wherever |BN(z)| is within a bf16 ulp of zero)."""
    return z + (z.bfloat16().float() - z).detach()


def swish(x):
    return x * torch.sigmoid(x)


ACTS = {0: lambda x: x, 1: F.relu, 2: swish, 3: F.elu, 4: torch.sigmoid}


def bn_tuple(c):
    g = (torch.rand(c, device=dev()) + 0.5).requires_grad_(True)
    b = (torch.randn(c, device=dev()) * 0.1).requires_grad_(True)
    rm = torch.randn(c, device=dev()) * 0.1
    rv = torch.rand(c, device=dev()) + 0.5
    nbt = torch.tensor(0, device=dev())
    return g, b, rm, rv, nbt


@pytest.mark.parametrize("cin,cout,stride,act,res,bias,n,h,w", [
    (24, 40, 1, 1, False, False, 2, 10, 6),
    (152, 152, 1, 1, True, False, 2, 8, 8),
    (32, 8, 1, 0, False, True, 1, 6, 10),
    (64, 376, 2, 0, False, False, 2, 8, 12),
    (376, 112, 1, 2, False, True, 3, 5, 7),
    (936, 936, 1, 1, True, False, 2, 4, 4),
    (64, 152, 1, 1, True, False, 4, 64, 64),        # many row blocks, ragged last 64-channel chunk, 256 partial statistic rows
    (32, 24, 1, 1, False, False, 4, 128, 256),      # > 512 partial rows: folded before the fused apply
    (376, 376, 1, 2, False, False, 16, 16, 32),     # deep-stage shape (stage 3 at N = 16), Swish
])
def test_conv1x1_bn_act(K, cin, cout, stride, act, res, bias, n, h, w):
    x = rnd(n, cin, h, w)
    wt = rnd(cout, cin, 1, 1, scale=cin ** -0.5)
    cb = rnd(cout, scale=0.1) if bias else None
    g, b, rm, rv, nbt = bn_tuple(cout)
    ho, wo = h // stride, w // stride
    r = rnd(n, cout, ho, wo) if res else None
    up = rnd(n, cout, ho, wo)
    # HIP
    xk = nhwc(x).requires_grad_(True)
    wk = wt.clone().requires_grad_(True)
    cbk = cb.clone().requires_grad_(True) if bias else None
    rk = nhwc(r).requires_grad_(True) if res else None
    rm_k, rv_k = rm.clone(), rv.clone()
    out = K.conv_bn_act(xk, wk, cbk, (g, b, rm_k, rv_k, nbt), res=rk, kind="1x1", stride=stride, act=act, eps=1e-3, momentum=0.01)
    out.backward(nhwc(up))
    gk, bk = g.grad.clone(), b.grad.clone()
    g.grad = b.grad = None
    # reference
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    z = bfr(F.conv2d(xr, wr, cb, stride))
    rm_r, rv_r = rm.clone(), rv.clone()
    y = F.batch_norm(z, rm_r, rv_r, g, b, True, 0.01, 1e-3)
    rr = r.clone().requires_grad_(True) if res else None
    if res:
        y = y + rr
    y = ACTS[act](y)
    y.backward(up)
    close(nchw(out), y, ACT_TOL, "out")
    close(nchw(xk.grad), xr.grad, GRAD_TOL, "dx")
    close(wk.grad, wr.grad, GRAD_TOL, "dw")
    close(gk, g.grad, GRAD_TOL, "dgamma")
    close(bk, b.grad, GRAD_TOL, "dbeta")
    if res:
        close(nchw(rk.grad), rr.grad, GRAD_TOL, "dres")
    close(rm_k, rm_r, 1e-2, "running_mean")
    close(rv_k, rv_r, 1e-2, "running_var")
    assert int(nbt) == 1
    g.grad = b.grad = None


@pytest.mark.parametrize("c,stride,n,h,w", [(24, 2, 2, 12, 16), (64, 1, 2, 9, 10), (152, 1, 1, 6, 6), (8, 2, 2, 8, 8), (936, 2, 1, 4, 8),
                                            (24, 1, 2, 20, 36), (376, 1, 2, 16, 32), (936, 1, 3, 8, 16), (72, 1, 1, 17, 5)])
def test_grouped_conv_bn_relu(K, c, stride, n, h, w):
    x = rnd(n, c, h, w)
    wt = rnd(c, 8, 3, 3, scale=0.15)
    g, b, rm, rv, nbt = bn_tuple(c)
    ho, wo = h // stride, w // stride
    up = rnd(n, c, ho, wo)
    xk = nhwc(x).requires_grad_(True)
    wk = wt.clone().requires_grad_(True)
    out = K.conv_bn_act(xk, wk, None, (g, b, rm.clone(), rv.clone(), nbt), kind="g3x3", stride=stride, act=1)
    out.backward(nhwc(up))
    gk, bk = g.grad.clone(), b.grad.clone()
    g.grad = b.grad = None
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    y = F.relu(F.batch_norm(bfr(F.conv2d(xr, wr, None, stride, 1, 1, c // 8)), rm.clone(), rv.clone(), g, b, True, 0.1, 1e-5))
    y.backward(up)
    close(nchw(out), y, ACT_TOL, "out")
    close(nchw(xk.grad), xr.grad, GRAD_TOL, "dx")
    close(wk.grad, wr.grad, GRAD_TOL, "dw")
    close(gk, g.grad, GRAD_TOL, "dgamma")
    close(bk, b.grad, GRAD_TOL, "dbeta")
    g.grad = b.grad = None


@pytest.mark.parametrize("n,h,w", [(2, 16, 24), (2, 64, 128), (1, 34, 68), (3, 18, 22)])
def test_stem(K, n, h, w):
    """(2, 64, 128): whole 256-thread blocks only (rows staged through LDS, 1 KB per store instruction); (1, 34, 68): one whole block + a
    ragged one (thread-owned rows); (3, 18, 22): odd output width (pixel pairs straddle nothing, the last thread of a row owns one pixel)"""
    x = torch.randn(n, 3, h, w, device=dev())
    wt = torch.randn(32, 3, 3, 3, device=dev()) * 0.2
    g, b, rm, rv, nbt = bn_tuple(32)
    up = rnd(n, 32, h // 2, w // 2)
    wk = wt.clone().requires_grad_(True)
    out = K.conv_bn_act(x, wk, None, (g, b, rm.clone(), rv.clone(), nbt), kind="stem", act=1)
    out.backward(nhwc(up))
    gk = g.grad.clone()
    g.grad = b.grad = None
    wr = wt.clone().requires_grad_(True)
    y = F.relu(F.batch_norm(bfr(F.conv2d(x, wr, None, 2, 1)), rm.clone(), rv.clone(), g, b, True, 0.1, 1e-5))
    y.backward(up)
    close(nchw(out), y, ACT_TOL, "out")
    close(wk.grad, wr.grad, GRAD_TOL, "dw")
    close(gk, g.grad, GRAD_TOL, "dgamma")
    g.grad = b.grad = None


@pytest.mark.parametrize("c,n,h,w", [(112, 2, 8, 12), (16, 1, 5, 5), (112, 2, 4, 8)])
def test_depthwise(K, c, n, h, w):
    x = rnd(n, c, h, w)
    wt = rnd(c, 1, 3, 3, scale=0.3)
    up = rnd(n, c, h, w)
    xk = nhwc(x).requires_grad_(True)
    wk = wt.clone().requires_grad_(True)
    out = K.DwConv.apply(xk, wk)
    out.backward(nhwc(up))
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    y = F.conv2d(F.pad(xr, [1, 1, 1, 1]), wr, None, 1, 0, 1, c)
    y.backward(up)
    close(nchw(out), y, ACT_TOL, "out")
    close(nchw(xk.grad), xr.grad, GRAD_TOL, "dx")
    close(wk.grad, wr.grad, GRAD_TOL, "dw")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("c,n,h,w", [(112, 2, 8, 12), (16, 1, 4, 4), (8, 2, 2, 2)])
def test_maxpool(K, mode, c, n, h, w):
    x = rnd(n, c, h, w)
    if mode == 0:
        x = (x - 0.3).bfloat16().float()            # negative regions so that the ZERO padding wins some windows
    up = rnd(n, c, h // 2, w // 2)
    xk = nhwc(x).requires_grad_(True)
    out = K.MaxPool.apply(xk, mode)
    out.backward(nhwc(up))
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(F.pad(xr, [0, 1, 0, 1]), 3, 2) if mode == 0 else F.max_pool2d(xr, 3, 2, 1)
    y.backward(up)
    assert torch.equal(nchw(out), y), "max-pool values are exact"
    close(nchw(xk.grad), xr.grad, 1e-2, "dx")          # only bf16 rounding of summed routed gradients


@pytest.mark.parametrize("order", [0, 1])
def test_shared_gradient_slots(K, order):
    """ops.share / GradSlot: a map consumed by three fusion nodes (as the identity input of one, up-sampled into a second, max-pooled into
    a third) plus one consumer that knows nothing about slots gets the SAME gradient as when the autograd engine sums the four
    contributions itself (in-place fp32 accumulation rounds once per consumer, the engine's bf16 adds as well: bf16-level tolerance)."""
    n, c, h, w = 2, 16, 8, 12
    x = rnd(n, c, h, w)                      # the shared map
    a_hi, b_hi = rnd(n, c, 2 * h, 2 * w), rnd(n, c, 2 * h, 2 * w)      # double-resolution node: x enters up-sampled (mode 2)
    a_lo = rnd(n, c, h // 2, w // 2)                                  # half-resolution node: x enters max-pooled (mode 3)
    b_same = rnd(n, c, h, w)
    ps = [torch.rand(2, device=dev()) + 0.2, torch.rand(3, device=dev()) + 0.2, torch.rand(2, device=dev()) + 0.2]
    ups = [rnd(n, c, h, w), rnd(n, c, 2 * h, 2 * w), rnd(n, c, h // 2, w // 2), rnd(n, c, h, w)]

    def run(shared):
        xk = nhwc(x).requires_grad_(True)
        if shared:
            (x0, x1, x2, x3, x4, x5), slot = K.share(xk, 6)
        else:
            (x0, x1, x2, x3, x4, x5), slot = (xk,) * 6, None
        pk = [p.clone().requires_grad_(True) for p in ps]
        nodes = [lambda: K.Fuse.apply(pk[0], 1, 1, 0, x0, nhwc(b_same), None, (slot, None, None)),
                 lambda: K.Fuse.apply(pk[1], 1, 1, 2, nhwc(a_hi), nhwc(b_hi), x1, (None, None, slot)),
                 lambda: K.Fuse.apply(pk[2], 1, 3, 0, nhwc(a_lo), x2, None, (None, slot, None))]
        outs = [None] * 3
        for i in ([0, 1, 2] if order == 0 else [2, 0, 1]):      # backward visits the nodes in reverse creation order
            outs[i] = nodes[i]()
        # consumers without slot support: their gradients are added by Share.backward (hn_add_n: up to three per launch)
        plain = x3 * 0.5, x4 * 0.25, x5 * -1.5
        loss = sum((o.float() * nhwc(u).float()).sum() for o, u in zip(outs, ups[:3])) + sum((q.float() * nhwc(ups[3]).float()).sum() for q in plain)
        loss.backward()
        assert slot is None or slot.buf is None, "Share.backward hands the buffer over and resets the slot"
        return xk.grad.float(), [p.grad.clone() for p in pk]

    g_ref, pg_ref = run(False)
    g, pg = run(True)
    close(g, g_ref, 1e-2, "shared dx")
    for a, b in zip(pg, pg_ref):
        close(a, b, 1e-3, "fusion parameter gradients")


@pytest.mark.parametrize("order", [0, 1])
def test_shared_gradient_slots_of_top_down_nodes(K, order):
    """fuse_bwd_quads_kernel with accumulating destinations: a map that is the identity input of one top-down node and the up-sampled input
    of another (both through its GradSlot, either backward order) gets the gradient the autograd engine sums itself"""
    n, c, h, w = 2, 24, 8, 12
    x = rnd(n, c, h, w)
    a_hi, lo = rnd(n, c, 2 * h, 2 * w), rnd(n, c, h // 2, w // 2)
    ps = [torch.rand(2, device=dev()) + 0.2, torch.rand(2, device=dev()) + 0.2]
    ups = [rnd(n, c, 2 * h, 2 * w), rnd(n, c, h, w)]

    def run(shared):
        xk = nhwc(x).requires_grad_(True)
        (x0, x1), slot = K.share(xk, 2) if shared else ((xk, xk), None)
        pk = [p.clone().requires_grad_(True) for p in ps]
        nodes = [lambda: K.Fuse.apply(pk[0], 1, 2, 0, nhwc(a_hi), x0, None, (None, slot, None)),
                 lambda: K.Fuse.apply(pk[1], 1, 2, 0, x1, nhwc(lo), None, (slot, None, None))]
        outs = [None] * 2
        for i in ([0, 1] if order == 0 else [1, 0]):
            outs[i] = nodes[i]()
        sum((o.float() * nhwc(u).float()).sum() for o, u in zip(outs, ups)).backward()
        return xk.grad.float(), [p.grad.clone() for p in pk]

    g_ref, pg_ref = run(False)
    g, pg = run(True)
    close(g, g_ref, 1e-2, "shared dx")
    for a, b in zip(pg, pg_ref):
        close(a, b, 1e-3, "fusion parameter gradients")


@pytest.mark.parametrize("modes,sum_inside", [((1, 2, 0), True), ((1, 2, 0), False), ((1, 0, 2), True), ((1, 2, 1), True),
                                              ((1, 1, 3), True), ((1, 3, 0), True)])
def test_bifpn_fuse(K, modes, sum_inside):
    """sum_inside (ops.FUSE_SUM2X2): the 2 x 2 gradient sums of the up-sampled input inside hn_fuse_bwd -- one identity + one up-sampled
    input runs fuse_bwd_quads_kernel (the up-sampled one second or third), two identity inputs the generic quad walk; False: hn_sum2x2 launches"""
    n, c, h, w = 2, 16, 8, 12
    K.FUSE_SUM2X2 = sum_inside
    shapes = {1: (h, w), 2: (h // 2, w // 2), 3: (2 * h, 2 * w)}
    ins = [rnd(n, c, *shapes[m]) if m else None for m in modes]
    nw = 3 if modes[2] else 2
    p = torch.rand(nw, device=dev()) + 0.2
    p[1] = -0.3 if nw == 3 else p[1]          # one negative parameter exercises the relu gate
    up = rnd(n, c, h, w)
    pk = p.clone().requires_grad_(True)
    ik = [nhwc(t).requires_grad_(True) if t is not None else None for t in ins]
    out = K.Fuse.apply(pk, modes[0], modes[1], modes[2], ik[0], ik[1], ik[2])
    out.backward(nhwc(up))
    pr = p.clone().requires_grad_(True)
    wr = torch.relu(pr)
    wr = wr / (wr.sum() + 1e-4)
    ir = [t.clone().requires_grad_(True) if t is not None else None for t in ins]
    acc = 0
    for i, m in enumerate(modes):
        if m == 1:
            acc = acc + wr[i] * ir[i]
        elif m == 2:
            acc = acc + wr[i] * F.interpolate(ir[i], scale_factor=2, mode="nearest")
        elif m == 3:
            acc = acc + wr[i] * F.max_pool2d(F.pad(ir[i], [0, 1, 0, 1]), 3, 2)
    y = swish(acc)
    y.backward(up)
    close(nchw(out), y, ACT_TOL, "out")
    for i, m in enumerate(modes):
        if m:
            close(nchw(ik[i].grad), ir[i].grad, GRAD_TOL, f"din{i}")
    close(pk.grad, pr.grad, GRAD_TOL, "dweights")
    K.FUSE_SUM2X2 = True


def test_se_gate(K):
    n, c, cs, h, w = 3, 64, 8, 6, 10
    x = rnd(n, c, h, w)
    w1, b1 = rnd(cs, c, 1, 1, scale=0.2), rnd(cs, scale=0.1)
    w2, b2 = rnd(c, cs, 1, 1, scale=0.3), rnd(c, scale=0.1)
    up = rnd(n, c, h, w)
    xk = nhwc(x).requires_grad_(True)
    pk = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    out = K.SEGate.apply(xk, *pk)
    out.backward(nhwc(up))
    xr = x.clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    gate = torch.sigmoid(F.conv2d(F.relu(F.conv2d(F.adaptive_avg_pool2d(xr, 1), pr[0], pr[1])), pr[2], pr[3]))
    y = xr * gate
    y.backward(up)
    close(nchw(out), y, ACT_TOL, "out")
    close(nchw(xk.grad), xr.grad, GRAD_TOL, "dx")
    for a, b, nm in zip(pk, pr, ("dw1", "db1", "dw2", "db2")):
        close(a.grad, b.grad, GRAD_TOL, nm)


@pytest.mark.parametrize("c0,c1,cout,up,act,f32,n,h,w", [
    (16, 0, 64, 0, 3, False, 2, 4, 4),
    (64, 16, 64, 1, 3, False, 2, 8, 8),
    (112, 0, 512, 0, 3, False, 1, 6, 10),
    (128, 24, 128, 1, 3, False, 1, 12, 8),
    (64, 0, 5, 1, 0, True, 2, 8, 12),
    # many-workgroup launches (hundreds of patches x several cout tiles, more than one round of workgroups per CU): forward with
    # up-sampling + concat and a K that is not a multiple of 64, ragged patches, data gradient on the padded grid
    (128, 24, 512, 1, 3, False, 8, 50, 100),
    # ... and the folding data-gradient epilogue (interior + ring) at the size where ops.seg switches it on (>= 2^24 elements)
    (256, 0, 256, 0, 3, False, 8, 64, 128),
])
def test_seg_conv(K, c0, c1, cout, up, act, f32, n, h, w):
    """h, w = OUTPUT resolution; x0 lives at (h>>up, w>>up)."""
    x0 = rnd(n, c0, h >> up, w >> up)
    x1 = rnd(n, c1, h, w) if c1 else None
    wt = rnd(cout, c0 + c1, 3, 3, scale=(9 * (c0 + c1)) ** -0.5)
    bs = rnd(cout, scale=0.1)
    upg = rnd(n, cout, h, w)
    x0k = nhwc(x0).requires_grad_(True)
    x1k = nhwc(x1).requires_grad_(True) if c1 else None
    wk, bk = wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    out = K.SegConv.apply(x0k, x1k, wk, bk, up, act, f32)
    out.backward(upg.permute(0, 2, 3, 1).contiguous() if f32 else nhwc(upg))
    x0r = x0.clone().requires_grad_(True)
    x1r = x1.clone().requires_grad_(True) if c1 else None
    wr, br = wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    v = F.interpolate(x0r, scale_factor=2, mode="nearest") if up else x0r
    if c1:
        v = torch.cat([v, x1r], 1)
    y = ACTS[act](F.conv2d(F.pad(v, [1, 1, 1, 1], mode="reflect"), wr, br))
    y.backward(upg)
    close(nchw(out), y, ACT_TOL, "out")
    close(nchw(x0k.grad), x0r.grad, GRAD_TOL, "dx0")
    if c1:
        close(nchw(x1k.grad), x1r.grad, GRAD_TOL, "dx1")
    close(wk.grad, wr.grad, GRAD_TOL, "dw")
    close(bk.grad, br.grad, GRAD_TOL, "dbias")


@pytest.mark.parametrize("c0,c1,cout,n,h,w", [
    (64, 0, 64, 2, 16, 24),          # decoder.7 shape class: no skip, one 64-cout tile per phase
    (128, 24, 128, 1, 24, 16),       # decoder.5: skip operand joins as a pre-activation addend
    (256, 112, 256, 1, 16, 32),      # decoder.3 (the dominant launch): 128-cout tiles, four K chunks per phase in the data gradient
    (64, 16, 64, 3, 10, 6),          # ragged tiles (output 10x6 low-res cells)
    (256, 112, 256, 8, 48, 96),      # decoder.3 at a many-workgroup size (phase forward with addend, skip conv, phase data gradient)
    (64, 0, 64, 8, 128, 256),        # decoder.7: >= 1024 patches -> the persistent one-phase-per-workgroup forward (whole patches)
    (64, 0, 64, 5, 250, 330),        # ... ragged patches
])
@pytest.mark.parametrize("fwd_phase,dgrad_phase", [(True, True), (False, True), (True, False), (None, None)])
def test_seg_conv_up_phase_form(K, c0, c1, cout, n, h, w, fwd_phase, dgrad_phase):
    """ops.SegConvUp (phase form on the low-resolution grid, 4 of 9 taps per phase) == ELU(Conv3x3(reflect_pad(cat[up2(x0), x1])) + b):
    forward, both data gradients, weight / bias gradients.  h, w = LOW resolution of x0; the output is 2h x 2w."""
    x0 = rnd(n, c0, h, w)
    x1 = rnd(n, c1, 2 * h, 2 * w) if c1 else None
    wt = rnd(cout, c0 + c1, 3, 3, scale=(9 * (c0 + c1)) ** -0.5)
    bs = rnd(cout, scale=0.1)
    upg = rnd(n, cout, 2 * h, 2 * w)
    x0k = nhwc(x0).requires_grad_(True)
    x1k = nhwc(x1).requires_grad_(True) if c1 else None
    wk, bk = wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    assert K.seg_up_phase_ok(x0k, x1k, wk)
    K.SEG_FWD_PHASE, K.SEG_DGRAD_PHASE = fwd_phase, dgrad_phase          # every combination of forms is the same function
    try:
        out = K.SegConvUp.apply(x0k, x1k, wk, bk, False, False)
        out.backward(nhwc(upg))
    finally:
        K.SEG_FWD_PHASE = K.SEG_DGRAD_PHASE = None
    x0r = x0.clone().requires_grad_(True)
    x1r = x1.clone().requires_grad_(True) if c1 else None
    wr, br = wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    v = F.interpolate(x0r, scale_factor=2, mode="nearest")
    if c1:
        v = torch.cat([v, x1r], 1)
    y = F.elu(F.conv2d(F.pad(v, [1, 1, 1, 1], mode="reflect"), wr, br))
    y.backward(upg)
    close(nchw(out), y, ACT_TOL, "out")
    close(nchw(x0k.grad), x0r.grad, GRAD_TOL, "dx0")
    if c1:
        close(nchw(x1k.grad), x1r.grad, GRAD_TOL, "dx1")
    close(wk.grad, wr.grad, GRAD_TOL, "dw")
    close(bk.grad, br.grad, GRAD_TOL, "dbias")


@pytest.mark.parametrize("c,k,n,h,w", [(64, 5, 2, 8, 12), (64, 5, 1, 20, 36), (16, 3, 2, 4, 4),
                                       (64, 5, 5, 250, 330), (64, 5, 4, 256, 256)])   # >= 1024 patches: the persistent form (ragged / whole patches)
def test_seg_out_phase_form(K, c, k, n, h, w):
    """ops.SegOutUp (4-phase conv with replicate padding on the low-resolution grid) == Conv3x3(reflect_pad(nearest_up2(x))); h, w = INPUT size."""
    x = rnd(n, c, h, w)
    wt = rnd(k, c, 3, 3, scale=(9 * c) ** -0.5)
    bs = rnd(k, scale=0.1)
    upg = rnd(n, k, 2 * h, 2 * w)
    xk = nhwc(x).requires_grad_(True)
    wk, bk = wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    out = K.SegOutUp.apply(xk, wk, bk)
    out.backward(upg.permute(0, 2, 3, 1).contiguous())
    xr = x.clone().requires_grad_(True)
    wr, br = wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    y = F.conv2d(F.pad(F.interpolate(xr, scale_factor=2, mode="nearest"), [1, 1, 1, 1], mode="reflect"), wr, br)
    y.backward(upg)
    close(nchw(out), y, ACT_TOL, "out")
    close(nchw(xk.grad), xr.grad, GRAD_TOL, "dx")
    close(wk.grad, wr.grad, GRAD_TOL, "dw")
    close(bk.grad, br.grad, GRAD_TOL, "dbias")


@pytest.mark.parametrize("c0,cout,k,n,h,w", [(64, 64, 5, 2, 12, 20), (64, 64, 5, 16, 32, 64), (128, 64, 5, 2, 8, 8)])
def test_seg_out_gradient_handed_over_in_space_to_depth_order(K, c0, cout, k, n, h, w):
    """SegConvUp -> SegOutUp (the last decoder block and the output conv, head_seg/segmentation.py:100-104) with the output conv's data
    gradient written straight in the block's space-to-depth operand order (hn_conv3x3_dgrad_fold_s2d: no hn_space_to_depth_bf16 pass)
    == the same chain with the plain hand-over, bit for bit: every gradient, incl. the border pixels the ring fix-up touches."""
    x0 = rnd(n, c0, h, w)
    wt = rnd(cout, c0, 3, 3, scale=(9 * c0) ** -0.5)
    bs = rnd(cout, scale=0.1)
    wo = rnd(k, cout, 3, 3, scale=(9 * cout) ** -0.5)
    bo = rnd(k, scale=0.1)
    upg = rnd(n, 4 * h, 4 * w, k)
    K.SEG_DGRAD_PHASE = True
    fold_min = K.seg.SEG_FOLD_MIN_ELEMS
    K.SEG_FOLD_MIN_ELEMS = 0
    try:
        assert K.seg_s2d_handover_ok(nhwc(x0), None, wt)
        got = []
        for s2d in (False, True):
            leaves = [t.clone().requires_grad_(True) for t in (nhwc(x0), wt, bs, wo, bo)]
            y = K.SegConvUp.apply(leaves[0], None, leaves[1], leaves[2], False, True, s2d)
            out = K.SegOutUp.apply(y, leaves[3], leaves[4], True, None, s2d)
            out.backward(upg)
            got.append([t.grad.clone() for t in leaves])
    finally:
        K.SEG_DGRAD_PHASE = None
        K.SEG_FOLD_MIN_ELEMS = fold_min
    for name, a, b in zip(("dx0", "dw", "dbias", "dw_out", "dbias_out"), got[0], got[1]):
        assert torch.equal(a, b), f"{name}: max diff {(a.float() - b.float()).abs().max().item():.3e}"


@pytest.mark.parametrize("c0,c1,cout,k2,n,h,w", [(128, 24, 128, 64, 2, 12, 20), (256, 112, 256, 128, 16, 16, 32), (64, 0, 64, 64, 1, 8, 8)])
def test_seg_block_gradient_left_in_both_orders(K, c0, c1, cout, k2, n, h, w):
    """SegConvUp -> SegConv (decoder blocks 2i+1 -> 2i+2, head_seg/segmentation.py:92-99): the second block's folding data-gradient epilogue
    leaves the gradient a second time in the first block's space-to-depth operand order (GradSlot; hn_conv3x3_dgrad_fold_s2d with both
    outputs) == the chain in which the first block makes that copy itself, bit for bit."""
    x0 = rnd(n, c0, h, w)
    x1 = rnd(n, c1, 2 * h, 2 * w) if c1 else None
    wt = rnd(cout, c0 + c1, 3, 3, scale=(9 * (c0 + c1)) ** -0.5)
    bs = rnd(cout, scale=0.1)
    w2 = rnd(k2, cout, 3, 3, scale=(9 * cout) ** -0.5)
    b2 = rnd(k2, scale=0.1)
    upg = nhwc(rnd(n, k2, 2 * h, 2 * w))
    fold_min = K.seg.SEG_FOLD_MIN_ELEMS
    K.SEG_FOLD_MIN_ELEMS = 0
    try:
        got = []
        for use_slot in (False, True):
            leaves = [t.clone().requires_grad_(True) for t in ([nhwc(x0)] + ([nhwc(x1)] if c1 else []) + [wt, bs, w2, b2])]
            xa, xb = leaves[0], (leaves[1] if c1 else None)
            wa, ba, wb, bb = leaves[-4:]
            slot = K.GradSlot() if use_slot else None
            y = K.SegConvUp.apply(xa, xb, wa, ba, False, True, False, slot)
            out = K.SegConv.apply(y, None, wb, bb, 0, 3, False, True, False, slot)
            out.backward(upg)
            assert slot is None or slot.buf is None                    # taken by the first block's backward
            got.append([t.grad.clone() for t in leaves])
    finally:
        K.SEG_FOLD_MIN_ELEMS = fold_min
    for i, (a, b) in enumerate(zip(got[0], got[1])):
        assert torch.equal(a, b), f"gradient {i}: max diff {(a.float() - b.float()).abs().max().item():.3e}"


def test_pack_plan_matches_single_packs(K):
    """ops.PackPlan (every per-step weight pack of a model in two launches) writes bit for bit what the single-weight entry points write:
    dense 1x1 / 3x3 weights through the tiled transposing kernel (channel counts that are not multiples of 32: zero padding of both
    operand layouts), depthwise taps, grouped-conv stencil and block-diagonal operands, channel-slice and phase-form packs through the
    elementwise kernel."""
    shapes = [(24, 32, 1), (152, 64, 1), (936, 368, 1), (65, 448, 1), (5, 64, 3), (256, 368, 3), (36, 112, 1), (64, 8, 3),
              (24, 30, 1), (7, 13, 3)]                 # rows that are not whole float4s: the element-wise loader
    ws = [rnd(co, ci, k, k) for co, ci, k in shapes]
    dw = [rnd(112, 1, 3, 3), rnd(36, 1, 3, 3)]
    gw = [rnd(24, 8, 3, 3), rnd(152, 8, 3, 3)]
    seg_w, seg_b = rnd(64, 88, 3, 3), rnd(64)
    K.clear_pack_cache()
    K.start_pack_log()
    ref = [K.pack_conv_weight(w) for w in ws] + [K.pack_dw_weight(w) for w in dw] + [K.pack_gconv_weight(gw[0], 0), K.pack_gconv_weight(gw[1], 1)]
    ref += [K.pack_gconv_diag(w) for w in gw] + [K.pack_conv_weight_slice(seg_w, 64, 24), K.pack_phase_weight(seg_w, 64, seg_b)]
    ref = [tuple(t.clone() for t in r) for r in ref]
    log = K.stop_pack_log()
    assert len(log) == len(ref)
    K.clear_pack_cache()
    plan = K.PackPlan(log)
    for (key, _, _), v in zip(plan.entries, plan.values):   # poison: every element (padding included) must be rewritten ...
        if key == "gdiag":
            continue                               # ... except the block-diagonal operands: zero-filled once, only the diagonal blocks are refreshed
        for t in v[:3]:
            if t is not None:
                t.fill_(7.0)
    plan.run()
    for (key, _, _), v, r in zip(plan.entries, plan.values, ref):
        for got, want in zip(v[:len(r)], r):
            assert torch.equal(got.view(-1), want.view(-1)), key
    assert K.pack_conv_weight(ws[2])[0] is plan.packs[2][0]           # run() primes the cache
    assert K.pack_dw_weight(dw[0])[0] is plan.values[len(ws)][0]
    assert K.pack_phase_weight(seg_w, 64, seg_b)[2] is plan.values[-1][2]
    seg_b.add_(1.0)                                                   # a changed bias is noticed (repacked outside the plan)
    assert torch.equal(K.pack_phase_weight(seg_w, 64, seg_b)[2], seg_b.repeat(4))
    K.clear_pack_cache()


@pytest.mark.parametrize("k,c0,c1", [(64, 64, 0), (64, 40, 24), (5, 64, 0), (128, 96, 112)])
def test_phase_weight_pack_and_fold(K, k, c0, c1):
    """hn_pack_weight_ex (phase form / channel slice) and hn_phase_fold against the definition W_eff = W @ T^T of ops._phase_matrix:
    packed operands equal the plain packing of the materialised effective weights bit for bit when the tap sums are exact (bf16-valued
    weights: at most 4 terms of 8 significant bits each), the fold is the transpose map."""
    from multitask_hydranet_amd._lib import lib
    w = rnd(k, c0 + c1, 3, 3, scale=0.25)
    bias = rnd(k, scale=0.1)
    T = K._phase_matrix(dev())
    w_eff = (w[:, :c0].reshape(k * c0, 9) @ T.t()).view(k, c0, 2, 2, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(4 * k, c0, 3, 3).contiguous()
    K.clear_pack_cache()
    wp_ref, wt_ref = K.pack_conv_weight(w_eff)
    wp, wt, b_eff = K.pack_phase_weight(w, c0, bias)
    assert torch.equal(wp.view(-1), wp_ref.view(-1)) and torch.equal(wt.view(-1), wt_ref.view(-1))
    assert torch.equal(b_eff, bias.repeat(4))
    if c1:
        w1 = w[:, c0:].contiguous()
        a_ref, b_ref = K.pack_conv_weight(w1)
        a, b = K.pack_conv_weight_slice(w, c0, c1)
        assert torch.equal(a.view(-1), a_ref.view(-1)) and torch.equal(b.view(-1), b_ref.view(-1))
    # cache: same tensors while weight and bias are unchanged, repacked after an in-place update of either
    assert K.pack_phase_weight(w, c0, bias)[0] is wp
    bias.add_(1.0)
    assert torch.equal(K.pack_phase_weight(w, c0, bias)[2], bias.repeat(4))
    dw_eff = torch.randn(4 * k, c0, 3, 3, device=dev())
    dw1 = torch.randn(k, c1, 3, 3, device=dev()) if c1 else None
    db_eff = torch.randn(4 * k, device=dev())
    dw = torch.empty(k, c0 + c1, 3, 3, device=dev())
    db = torch.empty(k, device=dev())
    lib().call("hn_phase_fold", dw_eff.data_ptr(), dw1.data_ptr() if c1 else None, db_eff.data_ptr(), dw.data_ptr(), db.data_ptr(), k, c0, c1)
    ref0 = (dw_eff.view(2, 2, k, c0, 3, 3).permute(2, 3, 0, 1, 4, 5).reshape(k * c0, 36).double() @ T.double()).view(k, c0, 3, 3)
    torch.testing.assert_close(dw[:, :c0].double(), ref0, rtol=1e-6, atol=1e-6)
    if c1:
        assert torch.equal(dw[:, c0:], dw1)
    torch.testing.assert_close(db.double(), db_eff.view(4, k).double().sum(0), rtol=1e-6, atol=1e-6)
    K.clear_pack_cache()


@pytest.mark.parametrize("ca,cb", [(65, 65), (8, 3)])
def test_head_out_cat(K, ca, cb):
    """ops.HeadOutCat == cat([conv1x1(ta), conv1x1(tb)], channel) flattened to [N, H*W, ca+cb] (lane location head, lanedetect.py:86-93)"""
    n, c, h, w = 2, 48, 4, 6
    ta, tb = rnd(n, c, h, w), rnd(n, c, h, w)
    wa, wb = rnd(ca, c, 1, 1, scale=0.2), rnd(cb, c, 1, 1, scale=0.2)
    ba, bb = rnd(ca, scale=0.1), rnd(cb, scale=0.1)
    ak, bk = nhwc(ta).requires_grad_(True), nhwc(tb).requires_grad_(True)
    prm = [t.clone().requires_grad_(True) for t in (wa, ba, wb, bb)]
    out = K.HeadOutCat.apply(*prm, ak, bk)
    up = torch.randn_like(out)
    out.backward(up)
    ar, br = ta.clone().requires_grad_(True), tb.clone().requires_grad_(True)
    ref_p = [t.clone().requires_grad_(True) for t in (wa, ba, wb, bb)]
    y = torch.cat([F.conv2d(ar, ref_p[0], ref_p[1]), F.conv2d(br, ref_p[2], ref_p[3])], 1).permute(0, 2, 3, 1).reshape(n, h * w, ca + cb)
    y.backward(up)
    close(out, y, ACT_TOL, "out")
    close(nchw(ak.grad), ar.grad, GRAD_TOL, "dta")
    close(nchw(bk.grad), br.grad, GRAD_TOL, "dtb")
    for got, ref, nm in zip(prm, ref_p, ("dwa", "dba", "dwb", "dbb")):
        close(got.grad, ref.grad, GRAD_TOL, nm)


@pytest.mark.parametrize("c,k,n,h,w", [(64, 5, 2, 8, 12), (64, 5, 1, 20, 36), (64, 3, 2, 16, 18), (48, 8, 1, 4, 6),
                                       (64, 5, 5, 250, 330), (64, 8, 4, 256, 256)])   # >= 1024 patches: the persistent form
def test_seg_out_argmax_fused(K, c, k, n, h, w):
    """hn_conv3x3_out_argmax (deploy: the output conv's epilogue takes the arg-max over the classes, the logits are never written) ==
    arg-max of the logits SegOutUp writes, bit for bit (same accumulation, first maximum wins); h, w = INPUT size."""
    x = nhwc(rnd(n, c, h, w))
    wt = rnd(k, c, 3, 3, scale=(9 * c) ** -0.5)
    bs = rnd(k, scale=0.1)
    with torch.no_grad():
        assert K.seg_out_argmax_ok(x, wt)
        logits = K.SegOutUp.apply(x, wt, bs)
        want = K.argmax_channels(logits.permute(0, 3, 1, 2))
        got = K.seg_out_argmax(x, wt, bs)
    assert got.dtype == torch.int64 and got.shape == (n, 2 * h, 2 * w)
    assert torch.equal(got, want)
    assert torch.equal(want, torch.argmax(logits, dim=3))


@pytest.mark.parametrize("with_dw,cout,k,act", [(True, 36, 4, 0), (True, 81, 9, 4), (False, 65, 65, 0), (False, 2, 2, 0)])
def test_head_out(K, with_dw, cout, k, act):
    n, c = 2, 16
    sizes = [(8, 8), (4, 4), (2, 2)] if with_dw else [(4, 6)]
    feats = [rnd(n, c, *s) for s in sizes]
    dw = rnd(c, 1, 3, 3, scale=0.3) if with_dw else None
    pw, bs = rnd(cout, c, 1, 1, scale=0.25), rnd(cout, scale=0.1)
    fk = [nhwc(t).requires_grad_(True) for t in feats]
    dwk = dw.clone().requires_grad_(True) if with_dw else None
    pwk, bk = pw.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    out = K.HeadOut.apply(dwk, pwk, bk, k, act, *fk)
    up = torch.randn_like(out)
    out.backward(up)
    fr = [t.clone().requires_grad_(True) for t in feats]
    dwr = dw.clone().requires_grad_(True) if with_dw else None
    pwr, br = pw.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    outs = []
    for f in fr:
        t = F.conv2d(F.pad(f, [1, 1, 1, 1]), dwr, None, 1, 0, 1, c) if with_dw else f
        t = F.conv2d(t, pwr, br).permute(0, 2, 3, 1).contiguous()
        outs.append(t.view(n, -1, k))
    y = ACTS[act](torch.cat(outs, 1))
    y.backward(up)
    close(out, y, ACT_TOL, "out")
    for a, b in zip(fk, fr):
        close(nchw(a.grad), b.grad, GRAD_TOL, "dfeat")
    close(pwk.grad, pwr.grad, GRAD_TOL, "dpw")
    close(bk.grad, br.grad, GRAD_TOL, "dbias")
    if with_dw:
        close(dwk.grad, dwr.grad, GRAD_TOL, "ddw")


def test_lane_concat(K):
    n, c = 2, 16
    p3, p4, p5, p6 = rnd(n, c, 16, 16), rnd(n, c, 8, 8), rnd(n, c, 4, 4), rnd(n, c, 2, 2)
    ks = [nhwc(t).requires_grad_(True) for t in (p3, p4, p5, p6)]
    out = K.LaneConcat.apply(*ks)
    up = rnd(n, 4 * c, 4, 4)
    out.backward(nhwc(up))
    rs = [t.clone().requires_grad_(True) for t in (p3, p4, p5, p6)]
    mp = lambda t: F.max_pool2d(t, 3, 2, 1)
    y = torch.cat([mp(mp(rs[0])), mp(rs[1]), rs[2], F.interpolate(rs[3], scale_factor=2, mode="nearest")], 1)
    y.backward(up)
    assert torch.equal(nchw(out), y)
    for a, b in zip(ks, rs):
        close(nchw(a.grad), b.grad, 1e-2, "dpyr")


def test_eval_mode_bn_uses_running_stats(K):
    n, c, h, w = 2, 24, 6, 6
    x = rnd(n, c, h, w)
    wt = rnd(c, c, 1, 1, scale=0.2)
    g, b, rm, rv, nbt = bn_tuple(c)
    with torch.no_grad():
        out = K.conv_bn_act(nhwc(x), wt, None, (g, b, rm, rv, nbt), act=1, training=False)
        y = F.relu(F.batch_norm(F.conv2d(x, wt), rm, rv, g, b, False, 0.1, 1e-5))
    close(nchw(out), y, ACT_TOL, "eval out")
    assert int(nbt) == 0


@pytest.mark.parametrize("use_top_k,n,h,w", [(True, 2, 32, 64), (False, 2, 16, 16), (True, 3, 128, 128)])
def test_seg_loss_topk_radix_select(K, use_top_k, n, h, w):
    """weighted CE + ignore_index + top-k mean: HIP radix select vs torch.sort (fp32: rtol 1e-5 on the scalar, 1e-4 on gradients)."""
    logits = torch.randn(n, h, w, 5, device=dev()) * 2
    tgt = torch.randint(0, 5, (n, h, w), device=dev())
    tgt[0, :2] = 255
    cw = torch.tensor([0.1, 0.5, 1.0, 5.0, 5.0], device=dev())
    lk = logits.clone().requires_grad_(True)
    for target in (tgt, tgt.float()):
        lk.grad = None
        out = K.SegLoss.apply(lk, target, cw, use_top_k, 0.3, 255)
        (out * 3.0).backward()
        lr = logits.clone().requires_grad_(True)
        loss = F.cross_entropy(lr.permute(0, 3, 1, 2), tgt, weight=cw, ignore_index=255, reduction="none").reshape(n, -1)
        if use_top_k:
            k = int(0.3 * h * w)
            loss = torch.sort(loss, dim=1, descending=True)[0][:, :k]
        ref = loss.mean()
        (ref * 3.0).backward()
        assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref)), (float(out), float(ref))
        close(lk.grad, lr.grad, 1e-4, "dlogits")


def test_argmax_channels_is_exact(K):
    logits = torch.randn(2, 16, 24, 5, device=dev())
    logits[0, 0, 0, :] = 1.0                              # a tie: the first maximum wins, like torch.argmax
    got = K.argmax_channels(logits.permute(0, 3, 1, 2))
    assert torch.equal(got, torch.argmax(logits.permute(0, 3, 1, 2), dim=1))


def test_det_loss_matches_reference_loop(K):
    """HIP detection loss vs the oracle's per-image loop (the reference algorithm) on the recorded KAT inputs: values 1e-5, grads 1e-4."""
    import numpy as np
    from oracle import hydranet_oracle as O
    from tests.helpers import load_npz
    z = load_npz("loss_kats.npz")
    t = lambda k: torch.from_numpy(z[k])
    anc = t("det/anchors")
    for ann in (t("det/ann"), torch.ones(3, 16, 5)):
        cls = (t("det/cls") * 4.5).clamp(0, 1)                   # spread over (0, 0.9]: exercises both focal branches and the clamp
        cls[0, :50] = 0.0
        cls[0, 50:100] = 1.0
        c1, r1 = cls.clone().requires_grad_(True), t("det/reg").clone().requires_grad_(True)
        ref_c, ref_r = O.det_loss(c1, r1, anc, ann)
        (ref_c.sum() + 50 * ref_r.sum()).backward()
        c2, r2 = cls.clone().to(dev()).requires_grad_(True), t("det/reg").clone().to(dev()).requires_grad_(True)
        got_c, got_r = K.det_loss_hip(c2, r2, anc.to(dev()), ann.to(dev()))
        (got_c.sum() + 50 * got_r.sum()).backward()
        for a, b in ((got_c, ref_c), (got_r, ref_r)):
            assert abs(float(a) - float(b)) <= 2e-5 * max(abs(float(b)), 1e-6), (float(a), float(b))
        close(c2.grad.cpu(), c1.grad, 2e-4, "dcls")
        close(r2.grad.cpu(), r1.grad if r1.grad is not None else torch.zeros_like(r1), 2e-4, "dreg")   # no positives: zero gradient


@pytest.mark.parametrize("case", ["random", "all_background", "no_negative", "few_pos_ties"])
def test_lane_losses_hip_vs_torch(K, case):
    """HIP lane losses (radix-select OHEM threshold, masked Huber) against the static-shape torch forms of losses.py (which
    tests/test_host_cpu.py pins to the oracle / reference KATs), values and gradients."""
    from tests import torch_losses as L
    g = torch.Generator(device="cuda").manual_seed(3)
    n, hw, ppl = 4, 96, 20
    Lc = 2 * ppl + 2
    fgm = torch.rand(n, hw, device="cuda", generator=g) < 0.05
    if case == "all_background":
        fgm[:] = False
    elif case == "no_negative":
        fgm[:] = True
    tgt = torch.stack([(~fgm).float(), fgm.float()], -1)
    z = torch.randn(n, hw, 2, device="cuda", generator=g)
    if case == "few_pos_ties":
        z = (z * 2).round() / 2                       # many exactly equal log-probs around the OHEM threshold
    lt = torch.randn(n, hw, Lc, device="cuda", generator=g) * (torch.rand(n, hw, Lc, device="cuda", generator=g) < 0.7)
    lp = torch.randn(n, hw, Lc, device="cuda", generator=g) * 2
    wts = torch.tensor([1.3, 0.7, 2.1], device="cuda")
    res = []
    for impl in ("hip", "torch"):
        zz, pp = z.clone().requires_grad_(True), lp.clone().requires_grad_(True)
        if impl == "hip":
            pos, neg, pm, pn = K.lane_cls_loss_hip(tgt, zz)
            loc = K.lane_loc_loss_hip(pm, pn, lt, pp, points_per_line=ppl)
        else:
            pos, neg, pm, pn = L.lane_cls_loss(tgt, zz)
            loc = L.lane_loc_loss(pm, pn, lt, pp, points_per_line=ppl)
        (pos * wts[0] + neg * wts[1] + loc * wts[2]).backward()
        res.append((pos.detach(), neg.detach(), loc.detach(), zz.grad, pp.grad))
    for a, b, nm in zip(res[0], res[1], ("pos", "neg", "loc", "dcls", "dloc")):
        err = float((a - b).abs().max())
        assert err <= 2e-5 * max(float(b.abs().max()), 1.0), (case, nm, err)
    with pytest.raises(IndexError):                   # the reference's hard-coded column 160/161 on a narrower tensor
        K.lane_loc_loss_hip(pm, pn, lt, lp)


@pytest.mark.parametrize("k,spread", [(1, 50.0), (37, 30.0), (700, 200.0), (3000, 400.0)])
def test_device_nms_bit_exact_vs_oracle(K, k, spread):
    """hn_nms_sorted (IoU bit-mask + one-wave scan) keeps exactly the indices of the ORACLE's greedy NMS (oracle.nms_greedy, the restated
    torchvision semantics the CPU tests pin with known answers), incl. many exact score ties (stable order)."""
    from multitask_hydranet_amd._lib import lib
    from oracle import hydranet_oracle as O
    g = torch.Generator(device="cuda").manual_seed(k)
    ctr = torch.rand(k, 2, device="cuda", generator=g) * spread
    wh = torch.rand(k, 2, device="cuda", generator=g) * 40 + 4
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    scores = (torch.rand(k, device="cuda", generator=g) * 20).round() / 20
    for thr in (0.3, 0.5):
        order = torch.argsort(scores, descending=True, stable=True)
        b = boxes[order].contiguous()
        mask = torch.empty((lib().query("hn_nms_mask_words", k),), device="cuda", dtype=torch.int64)
        keep = torch.empty((k,), device="cuda", dtype=torch.uint8)
        lib().call("hn_nms_sorted", b.data_ptr(), k, float(thr), mask.data_ptr(), keep.data_ptr())
        dev_keep = order[keep.bool()].cpu()
        ref = O.nms_greedy(boxes.cpu(), scores.cpu(), thr)
        assert torch.equal(dev_keep, ref), (k, thr)
        assert 0 < len(ref) <= k


@pytest.mark.parametrize("c,n,hs,ws", [(112, 2, (8,), (12,)), (16, 1, (5,), (5,)), (112, 16, (64,), (128,)), (64, 3, (7,), (9,)),
                                       (112, 4, (20, 10, 5, 3, 2), (20, 10, 5, 3, 2)), (112, 16, (64, 32, 16, 8, 4), (128, 64, 32, 16, 8))])
def test_depthwise_backward_one_pass(K, c, n, hs, ws):
    """hn_dwconv_bwd_levels (data + weight gradient from one pass over (dz, x)) against the separate launches: dx bit-identical (same
    products in the same order), dweight to fp32 summation order; ragged level packing (alignment rows zeroed) and accumulation."""
    geom = (n, list(hs), list(ws)) if len(hs) > 1 else None
    wt = rnd(c, 1, 3, 3, scale=0.3)
    _, wf = K.pack_dw_weight(wt)
    if geom is None:
        x, dz = nhwc(rnd(n, c, hs[0], ws[0])), nhwc(rnd(n, c, hs[0], ws[0]))
        dx_ref, dw_ref = K.k_dwconv(dz, wf), K.k_dwconv_wgrad(x, dz)
    else:
        rows_ = sum(K._pad_rows(n * h * w) for h, w in zip(hs, ws))
        x = (torch.randn(1, 1, rows_, c, device=dev())).bfloat16()
        dz = (torch.randn(1, 1, rows_, c, device=dev())).bfloat16()
        dx_ref, dw_ref = K.k_dwconv_levels(dz, wf, geom), K.k_dwconv_wgrad_levels(x, dz, geom)
    dx, dw = K.k_dwconv_bwd(dz, x, wf, geom)
    assert torch.equal(dx, dx_ref)
    close(dw, dw_ref, 1e-4, "dweight")
    base = torch.randn_like(dx.float()).bfloat16()
    got = base.clone()
    K.k_dwconv_bwd(dz, x, wf, geom, into=got)
    if geom is None:
        want = (dx_ref.float() + base.float()).bfloat16()
        # the accumulate path adds the fp32 sum to the stored bf16 value: one rounding, not two
        assert float((got.float() - want.float()).abs().max()) <= 2e-2 * float(want.float().abs().max())
    else:
        want = base.clone()
        K.k_dwconv_levels(dz, wf, geom, into=want)
        assert torch.equal(got, want)
    none_dx, dw2 = K.k_dwconv_bwd(dz, x, wf, geom, want_dx=False)
    assert none_dx is None and torch.equal(dw2, dw)
    # against the fp32 definition
    if geom is None:
        xr = nchw(x).clone().requires_grad_(True)
        wr = wt.clone().requires_grad_(True)
        F.conv2d(xr, wr, None, 1, 1, 1, c).backward(nchw(dz))
        close(nchw(dx), xr.grad, GRAD_TOL, "dx vs torch")
        close(dw, wr.grad, GRAD_TOL, "dw vs torch")


@pytest.mark.parametrize("n,h,w,cz,nout,clamp,elu", [
    (2, 16, 16, 64, 64, 0, True), (1, 4, 4, 64, 64, 0, False), (3, 5, 7, 128, 112, 0, True), (2, 33, 18, 64, 256, 0, True),
    (2, 16, 32, 24, 64, 1, True), (1, 4, 5, 24, 64, 1, False), (2, 128, 256, 24, 64, 1, True), (2, 31, 47, 256, 128, 0, False),
])
def test_dgrad_fold_direct_vs_padded_grid_and_fold(K, n, h, w, cz, nout, clamp, elu):
    """hn_conv3x3_dgrad_fold (folding epilogue + ring fix-up) against hn_conv_gemm_nt(mode 3) on the padded grid + hn_seg_fold: identical
    away from the border (same bf16 values through the same operations), one extra bf16 rounding on the 2 (H + W) border pixels."""
    from multitask_hydranet_amd._lib import lib
    dz = nhwc(rnd(n, cz, h, w))
    wgt = rnd(cz, nout, 3, 3, scale=0.2)                     # forward weight [cout = cz][cin = nout]
    _, wt = K.pack_conv_weight(wgt)
    yp = nhwc(rnd(n, nout, h, w)) if elu else None
    kp = K.kp32(cz)
    dvp, _, _ = K.k_gemm_nt(dz, None, 3, (n, h + 2, w + 2), wt, nout, kp, 9, c0=cz, c1=0)
    ref = K.new_act(n, h, w, nout, dev())
    lib().call("hn_seg_fold", K.ptr(dvp), K.ld(dvp), 0, K.ptr(ref), K.ld(ref), K.ptr(yp), K.ld(yp) if yp is not None else 0, n, h, w, nout,
               2 if clamp else 0)
    assert K.dgrad_fold_ok(nout, h, w)
    got = K.k_dgrad_fold(dz, wt, n, h, w, nout, kp, 0, clamp, yp)
    inner = (slice(None), slice(2, h - 2), slice(2, w - 2))
    if h > 4 and w > 4:
        assert torch.equal(got[inner], ref[inner])
    close(got, ref, 1e-2, "folded dgrad")


@pytest.mark.parametrize("n,h,w,k,c0", [(2, 16, 16, 64, 64), (1, 8, 12, 64, 128), (2, 20, 36, 128, 256)])
def test_dgrad_fold_direct_phase_form(K, n, h, w, k, c0):
    """the same for the phase-form data gradient (space-to-depth gradient operand, 4 taps per phase, replicate-padding fold)"""
    from multitask_hydranet_amd._lib import lib
    wgt = rnd(k, c0, 3, 3, scale=0.1)
    _, wt_eff, _ = K.pack_phase_weight(wgt, c0, rnd(k))
    dzs = nhwc(rnd(n, 4 * k, h, w))
    yp = nhwc(rnd(n, c0, h, w))
    dvp = K.new_act(n, h + 2, w + 2, c0, dev())
    lib().call("hn_conv3x3_phase", K.ptr(dzs), 3, n, h + 2, w + 2, 4 * k, K.ld(dzs), K.ptr(wt_eff), c0, K.kp32(4 * k), None, 0, K.ptr(dvp),
               K.ld(dvp), k, None, 0)
    ref = K.new_act(n, h, w, c0, dev())
    lib().call("hn_seg_fold", K.ptr(dvp), K.ld(dvp), 0, K.ptr(ref), K.ld(ref), K.ptr(yp), K.ld(yp), n, h, w, c0, 2)
    got = K.k_dgrad_fold(dzs, wt_eff, n, h, w, c0, K.kp32(4 * k), k, 1, yp)
    assert torch.equal(got[:, 2:h - 2, 2:w - 2], ref[:, 2:h - 2, 2:w - 2])
    close(got, ref, 1e-2, "folded phase dgrad")


def test_wgrad_reduces_batched_in_one_launch(K):
    """k_gemm_tn(..., defer=batch) + batch.flush(): the slab reduces of several weight gradients (1x1 with few / many splits, the grouped
    3x3 block-diagonal form) in one launch give the gradients of the one-by-one launches"""
    cases = [((16, 8, 16), 936, 936, 0), ((16, 64, 128), 56, 56, 0), ((16, 16, 32), 368, 368, 5), ((4, 32, 32), 152, 64, 0)]
    batch = K.WgradBatch()
    got, want = [], []
    for (n, h, w), cin, cout, mode in cases:
        x = nhwc(rnd(n, cin, h, w))
        if mode == 5:
            dz = nhwc(rnd(n, cin, h, w))
            args = (x, None, 5, (n, h, w), dz, cin, 64, 9, 8)
            kw = dict(kh=3)
        else:
            dz = nhwc(rnd(n, cout, h, w))
            args = (x, None, 0, (n, h, w), dz, cout, K.kp32(cin), 1, cin)
            kw = {}
        want.append(K.k_gemm_tn(*args, **kw))
        got.append(K.k_gemm_tn(*args, defer=batch, **kw))
    batch.flush()
    for g, r in zip(got, want):
        close(g, r, 1e-5, "batched reduce")


@pytest.mark.parametrize("stage", ["stage4", "stage2", "stage0", "many"])
def test_grouped_deferred_wgrad(K, stage):
    """ops.WgradGroup / hn_wgrad_group: the 1x1 weight gradients of a whole backbone stage in one grouped launch (pixel splits + one
    grouped slab reduce only where the jobs cannot fill the chip) == the fp32 einsum on the bf16 operands (2e-3 of the gradient's max:
    fp32 accumulation in another order) and == the one-by-one k_gemm_tn launches (1e-5).  Shapes: a stage-4-like group (no split, tiles
    write the gradient itself), a stage-2-like group (splits + reduce; the first block's stride-2 shortcut reads x on the stride-2
    sub-grid, mode 1), stage 0 (24 / 32 channels: K padding and ragged couts), and more jobs than one launch holds."""
    if stage == "stage4":
        jobs = [((4, 8, 16), 376, 936, 1), ((4, 16, 32), 376, 936, 0)] + [((4, 8, 16), 936, 936, 0)] * 5
    elif stage == "stage2":
        jobs = [((4, 32, 64), 64, 152, 1), ((4, 64, 128), 64, 152, 0)] + [((4, 32, 64), 152, 152, 0)] * 3
    elif stage == "stage0":
        jobs = [((2, 64, 128), 32, 24, 1), ((2, 128, 256), 32, 24, 0), ((2, 64, 128), 24, 24, 0)]
    else:
        jobs = [((2, 8, 16), 152, 152, 0)] * 37
    group = K.WgradGroup()
    ws, refs, ones = [], [], []
    for i, ((n, h, w), cin, cout, mode) in enumerate(jobs):
        hi, wi = (2 * h, 2 * w) if mode == 1 else (h, w)
        x = nhwc(rnd(n, cin, hi, wi))
        dz = nhwc(rnd(n, cout, h, w, scale=0.1))
        wgt = torch.empty(cout, cin, 1, 1, device=dev())
        ws.append(wgt)
        group.add(wgt, x, dz, mode, (n, h, w), cin, cout)
        xs = x[:, ::2, ::2] if mode == 1 else x
        refs.append(torch.einsum("nhwo,nhwi->oi", dz.float(), xs.float()).reshape(cout, cin, 1, 1))
        ones.append(K.k_gemm_tn(x, None, mode, (n, h, w), dz, cout, K.kp32(cin), 1, cin))
    group.weights = tuple(ws)
    got = group.flush()
    assert len(got) == len(jobs) and not group.jobs
    for g, r, o in zip(got, refs, ones):
        assert g.shape == r.shape and g.dtype == torch.float32
        close(g, r, 2e-3, "grouped wgrad vs einsum")
        close(g, o, 1e-5, "grouped wgrad vs one-by-one")
    # weights without a queued job get no gradient
    group.weights = (ws[0], torch.empty(1))
    assert group.flush() == [None, None]


def test_grouped_deferred_gconv_wgrad(K):
    """ops.GradQueue.add_gconv / hn_gconv_wgrad_group: the grouped 3x3 (group width 8) weight gradients of several identity XBlocks in one
    patch-kernel launch + one extract launch == the one-by-one launches (hn_conv_gemm_tn mode 5; 1e-5: same kernel body, other patch
    splits) and == torch's grouped conv weight gradient on the bf16 operands (2e-3)"""
    import torch.nn.functional as F
    q = K.GradQueue()
    ws, want, ones = [], [], []
    for (n, h, w), c in [((16, 8, 16), 936), ((16, 8, 16), 936), ((4, 16, 32), 376), ((2, 32, 64), 152), ((2, 20, 24), 64)]:
        x = nhwc(rnd(n, c, h, w))
        dz = nhwc(rnd(n, c, h, w, scale=0.1))
        wgt = torch.empty(c, 8, 3, 3, device=dev())
        q.add_gconv(wgt, x, dz, (n, h, w), c)
        ws.append(wgt)
        ones.append(K.k_gemm_tn(x, None, 5, (n, h, w), dz, c, 64, 9, 8, kh=3))
        xx = x.float().permute(0, 3, 1, 2).requires_grad_(False)
        wz = torch.zeros(c, 8, 3, 3, device=dev(), requires_grad=True)
        F.conv2d(xx, wz, None, 1, 1, 1, c // 8).backward(dz.float().permute(0, 3, 1, 2))
        want.append(wz.grad)
    q.weights = tuple(ws)
    got = q.flush()
    for g, o, r in zip(got, ones, want):
        close(g, o, 1e-5, "grouped gconv wgrad vs one-by-one")
        close(g, r, 2e-3, "grouped gconv wgrad vs torch")


def test_grad_tail_jobs(K):
    """hn_grad_tail through ops.GradQueue: partial-row folds (short and tall), BiFPN fusion-weight Jacobians (2 and 3 weights, one clamped
    by the relu) and SE outer products, > 64 jobs (two launches), against torch in fp32"""
    q = K.GradQueue()
    ws, want = [], []
    g = torch.Generator(device=dev()).manual_seed(5)
    for rows, cols, shape in [(37, 1008, (112, 1, 3, 3)), (768, 1008, (112, 1, 3, 3)), (5, 64 * 72, (64, 8, 3, 3)), (2049, 72, (8, 1, 3, 3))] * 12:
        part = torch.randn(rows, cols, device=dev(), generator=g)
        w = torch.empty(shape, device=dev())
        q.add_rows(w, part, rows, cols, shape)
        ws.append(w)
        want.append(part.double().sum(0).float().view(shape))
    for nw, blocks in [(2, 1024), (3, 100), (3, 1)] * 6:
        praw = torch.tensor([0.7, -0.2, 1.3][:nw], device=dev())
        pw = torch.randn(blocks, 3, device=dev(), generator=g)
        q.add_fuse(praw, pw, blocks, 1e-4)
        ws.append(praw)
        p64 = praw.double().clone().requires_grad_(True)
        r = torch.relu(p64)
        wn = r / (r.sum() + 1e-4)
        (wn * pw.double().sum(0)[:nw]).sum().backward()
        want.append(p64.grad.float())
    for n, pi, qj in [(16, 234, 936), (16, 936, 234), (2, 6, 24)] * 3:
        p_, q_ = torch.randn(n, pi, device=dev(), generator=g), torch.randn(n, qj, device=dev(), generator=g)
        w, b = torch.empty(pi, qj, 1, 1, device=dev()), torch.empty(pi, device=dev())
        q.add_outer(w, b, p_, q_)
        ws += [w, b]
        want += [(p_.double().t() @ q_.double()).float().view(pi, qj, 1, 1), p_.double().sum(0).float()]
    assert len(q.tail) > K.GradQueue.MAX_TAIL
    q.weights = tuple(ws)
    got = q.flush()
    assert not q.tail and len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        # fp32 accumulation of up to 2049 terms vs float64; the fusion Jacobian is a difference of nearly equal sums
        close(a, b, 1e-3 if b.numel() <= 3 else 1e-4, "grad tail job %d %s" % (i, tuple(b.shape)))


def test_full_model_with_deferred_gradients_equals_immediate(K):
    """the whole tiny HydraNet step with ops.DEFER_WGRAD (stage-boundary and backbone-tail GradQueue flushes) vs every parameter gradient
    launched where autograd asks for it: identical losses, all gradients equal to fp32 summation-order noise, same set of parameters"""
    from multitask_hydranet_amd import HydraNet
    from tests.helpers import load_cfg, load_npz, tiny_state
    z = load_npz("tiny_hydranet.npz")
    cfgs = load_cfg("hydranet_tiny.yml")
    net = HydraNet(cfgs)
    net.load_state_dict(tiny_state(z))
    net = net.cuda().train()
    net.lane_points_per_line = int(z["meta/lane_points_per_line"])
    batch = {k[3:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("in/")}
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    res = []
    for defer in (False, True):
        K.DEFER_WGRAD = defer
        try:
            net.load_state_dict(sd)
            net.zero_grad(set_to_none=True)
            out = net(batch["image"])
            ld = net.cal_loss(out, batch)
            net.total_loss(ld).backward()
            res.append(({k: float(v) for k, v in ld.items()}, {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
        finally:
            K.DEFER_WGRAD = True
    (l0, g0), (l1, g1) = res
    assert l0 == l1 and g0.keys() == g1.keys() and len(g0) > 300
    for k in g0:
        close(g1[k], g0[k], 2e-5, k)


def test_backbone_stage_with_deferred_wgrad_equals_immediate(K):
    """HydraNet._backbone_shared with ops.DEFER_WGRAD: a stage's 1x1 weight gradients come out of the DeferredGrads node at the stage
    boundary.  Same forward (bit-identical outputs), every parameter gradient equal to the immediate path's to fp32 summation-order noise,
    identical input gradient."""
    from multitask_hydranet_amd import HydraNet
    from tests.helpers import load_cfg
    cfgs = load_cfg("hydranet_tiny.yml")
    torch.manual_seed(3)
    net = HydraNet(cfgs).cuda().train()
    x = torch.randn(4, 3, 128, 256, device=dev())
    res = []
    for defer in (False, True):
        K.DEFER_WGRAD = defer
        try:
            net.zero_grad(set_to_none=True)
            sd = {k: v.clone() for k, v in net.state_dict().items()}
            feats = net._backbone(x)
            net._flush_nbt()
            loss = sum((f.float() ** 2).mean() for f in feats)
            loss.backward()
            res.append(([f.detach().clone() for f in feats], {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
            net.load_state_dict(sd)
        finally:
            K.DEFER_WGRAD = True
    (f0, g0), (f1, g1) = res
    assert all(torch.equal(a, b) for a, b in zip(f0, f1))
    assert g0.keys() == g1.keys() and any(".conv_block_1.0.weight" in k for k in g0)
    for k in g0:
        close(g1[k], g0[k], 1e-5, k)


@pytest.mark.parametrize("n,h,w,c,mode", [(16, 8, 16, 936, 1), (4, 16, 32, 376, 1), (2, 32, 64, 152, 1), (2, 32, 64, 152, 2), (1, 64, 128, 64, 1),
                                          (16, 8, 16, 936, 5), (4, 16, 32, 376, 5), (3, 20, 24, 64, 5), (2, 33, 47, 152, 5)])
def test_backward_reductions_in_producer_epilogues(K, n, h, w, c, mode):
    """hn_conv_gemm_nt_stat: the statistics rows of a data-gradient producer carry the reduce pass that used to follow it.
    emode 1 (1x1 GEMM dbg = dz3 W3): sum over the rows of q * bf16(relu(sc z + sh)) == hn_se_bwd_reduce_fused's partials;
    emode 2 (1x1 GEMM, or mode 5 = grouped 3x3 data gradient on the direct kernel): (sum g, sum g * xhat), g = q * [sc z + sh > 0]
    == hn_bn_bwd_reduce_fused.  Checked against fp64 sums over the kernel's own bf16 output (1e-5 of the column's absolute sum) and the
    output must be bit-identical to the launch without the epilogue operand."""
    torch.manual_seed(n * 1000 + c)
    m = n * h * w
    z = nhwc(rnd(n, c, h, w))
    coef = torch.stack([torch.rand(c, device=dev()) + 0.5, torch.randn(c, device=dev()) * 0.3, torch.randn(c, device=dev()) * 0.2,
                        torch.rand(c, device=dev()) + 0.5]).contiguous()
    dz = nhwc(rnd(n, c, h, w, scale=0.1))
    if mode == 5:
        wgt = rnd(c, 8, 3, 3, scale=0.2)
        _, wd = K.pack_gconv_diag(wgt)
        args = (dz, None, 5, (n, h, w), wd, c, 64, 9)
        emode = 2
    else:
        wgt = rnd(c, c, 1, 1, scale=0.05)
        _, wt = K.pack_conv_weight(wgt)
        args = (dz, None, 0, (n, h, w), wt, c, K.kp32(c), 1)
        emode = mode
    plain, _, _ = K.k_gemm_nt(*args)
    out, ps, pq = K.k_gemm_nt(*args, estat=(emode, z, coef))
    assert torch.equal(out, plain)
    q = out.double().view(m, c)
    # the kernel's fp32 fused multiply-add decides the mask (one rounding)
    pre32 = (coef[0].double() * z.double().view(m, c) + coef[1].double()).float()
    if emode == 1:
        b = torch.relu(pre32).bfloat16().double()
        want1 = (q * b).sum(0)
        scale1 = (q * b).abs().sum(0)
        assert pq is None
        close_cols = [(ps.double().sum(0), want1, scale1)]
    else:
        g = torch.where(pre32 > 0, q, torch.zeros_like(q))
        xh = ((z.float().view(m, c) - coef[2]) * coef[3]).double()
        close_cols = [(ps.double().sum(0), g.sum(0), g.abs().sum(0)), (pq.double().sum(0), (g * xh).sum(0), (g * xh).abs().sum(0))]
    for got, want, scale in close_cols:
        assert float(((got - want).abs() / (scale + 1e-6)).max()) < 1e-5
    # rows: one per pixel tile (tiles never straddle the launch's rows), so per-image sums can be taken from whole rows when hw % tile == 0
    if mode != 5 and (h * w) % (m // ps.shape[0]) == 0:
        per_img = ps.double().view(n, -1, c).sum(1)
        want_img = ((q * b) if emode == 1 else g).view(n, h * w, c).sum(1)
        assert float(((per_img - want_img).abs() / (want_img.abs() + 1e-3)).max()) < 1e-3


def test_backbone_with_epilogue_reductions_equals_reduce_passes(K):
    """the XBlock backward with ops.EPILOGUE_STATS (SE gate-gradient partials and BatchNorm-1 backward sums from the producers' epilogues)
    vs the separate reduce passes: same forward, gradients equal up to the bf16 roundings a 1e-7 change of a channel mean can flip"""
    from multitask_hydranet_amd import HydraNet
    from tests.helpers import load_cfg
    cfgs = load_cfg("hydranet_tiny.yml")
    torch.manual_seed(5)
    net = HydraNet(cfgs).cuda().train()
    x = torch.randn(4, 3, 128, 256, device=dev())
    res = []
    for ep in (False, True):
        K.EPILOGUE_STATS = ep
        try:
            net.zero_grad(set_to_none=True)
            sd = {k: v.clone() for k, v in net.state_dict().items()}
            feats = net._backbone(x)
            net._flush_nbt()
            loss = sum((f.float() ** 2).mean() for f in feats)
            loss.backward()
            res.append(([f.detach().clone() for f in feats], {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
            net.load_state_dict(sd)
        finally:
            K.EPILOGUE_STATS = True
    (f0, g0), (f1, g1) = res
    assert all(torch.equal(a, b) for a, b in zip(f0, f1))
    assert g0.keys() == g1.keys()
    # round 5: every block of the tiny cfg's five stages takes the fused node now (4 x 8 and 2 x 4 maps included: BatchNorm over 32-128
    # samples), so a flipped rounding travels through up to 8 more blocks before it reaches the stem: direction to 1e-3, size to 6e-2 (measured: 4.5e-2 on one BatchNorm bias of stage 0)
    for k in g0:
        a, b = g1[k].float().flatten(), g0[k].float().flatten()
        if float(b.abs().max()) > 0:
            assert float(F.cosine_similarity(a, b, dim=0)) >= 0.999, k
        close(g1[k], g0[k], 6e-2, k)


@pytest.mark.parametrize("n,h,w,c,ck", [(16, 16, 32, 376, 376), (16, 8, 16, 936, 936), (2, 8, 8, 152, 96), (1, 5, 13, 72, 72)])
def test_next_blocks_dgrad_gemm_makes_the_batchnorm3_backward_partials(K, n, h, w, c, ck):
    """hn_conv_gemm_nt_stat3 (GemmNT::emode 3): dx = dz1 W1 + g, and the statistics rows = the reduce pass of the previous block's masked
    BatchNorm backward over that dx: (sum g', sum g' xhat), g' = dx [y > 0].  The output is bit-identical to the launch without the
    operand; the sums are checked against fp64 sums over the kernel's own bf16 output (1e-5 of the column's absolute sum) and against
    hn_bn_bwd_reduce_fused's partials; K = 936 runs the two-K-group form, the 5 x 13 grid a ragged last pixel tile."""
    torch.manual_seed(c + h)
    m = n * h * w
    z = nhwc(rnd(n, c, h, w))
    y = nhwc(rnd(n, c, h, w))
    coef = torch.stack([torch.rand(c, device=dev()) + 0.5, torch.randn(c, device=dev()) * 0.3, torch.randn(c, device=dev()) * 0.2,
                        torch.rand(c, device=dev()) + 0.5]).contiguous()
    dz = nhwc(rnd(n, ck, h, w, scale=0.1))
    g = nhwc(rnd(n, c, h, w, scale=0.1))
    _, wt = K.pack_conv_weight(rnd(ck, c, 1, 1, scale=0.05))
    args = (dz, None, 0, (n, h, w), wt, c, K.kp32(ck), 1)
    plain, _, _ = K.k_gemm_nt(*args, addend=g)
    out, ps, pq = K.k_gemm_nt(*args, addend=g, estat=(3, z, coef, y))
    assert torch.equal(out, plain)
    assert ps.shape == ((m + 63) // 64, c) == pq.shape
    q = out.double().view(m, c)
    gm = torch.where(y.view(m, c) > 0, q, torch.zeros_like(q))
    xh = ((z.float().view(m, c) - coef[2]) * coef[3]).double()
    for got, want, scale in [(ps.double().sum(0), gm.sum(0), gm.abs().sum(0)), (pq.double().sum(0), (gm * xh).sum(0), (gm * xh).abs().sum(0))]:
        assert float(((got - want).abs() / (scale + 1e-6)).max()) < 1e-5
    # the pass it replaces
    rb = K.lib().query("hn_fused_row_block", m, c, 0, 0, 1)
    pr = (m + rb - 1) // rb
    pg, pgx = torch.empty((pr, c), device=dev()), torch.empty((pr, c), device=dev())
    K.lib().call("hn_bn_bwd_reduce_fused", K.ptr(out), K.ld(out), K.ptr(z), K.ld(z), K.ptr(y), K.ld(y), K.ptr(coef), K.ACT_RELU, None, None,
                 0, m, c, rb, K.ptr(pg), K.ptr(pgx))
    for got, want, scale in [(ps.double().sum(0), pg.double().sum(0), gm.abs().sum(0)), (pq.double().sum(0), pgx.double().sum(0), (gm * xh).abs().sum(0))]:
        assert float(((got - want).abs() / (scale + 1e-6)).max()) < 1e-5


def test_backbone_with_batchnorm3_partials_from_the_next_block_equals_reduce_passes(K):
    """ops.BN3_PARTS_FROM_DGRAD (an identity XBlock's last backward GEMM makes the previous block's BatchNorm-3 backward sums) vs the
    separate reduce passes on the big cfg's backbone: same forward; the hand-over is taken once per identity block of stages 2..4.  The sums
    themselves are pinned by the kernel test above (1e-5 against hn_bn_bwd_reduce_fused); this test pins the WIRING (whose z3 / coefficients /
    mask reach which block).  A different summation order flips bf16 roundings of dz3, and 30 randomly initialised blocks with batch
    statistics over 256..8192 samples amplify that: tools/scratch/bn3_parts_err.py shows the same 1e-3 (last blocks) .. 3e-2 (stem) of the
    gradient's maximum for EPILOGUE_STATS on/off.  So: the last block is untouched (bit-equal), the block before it sees only the direct
    effect (5e-3), everything else 8e-2 (a wrong operand gives O(1))."""
    from multitask_hydranet_amd import HydraNet
    from multitask_hydranet_amd.ops import backbone as B
    from tests.helpers import load_cfg
    cfgs = load_cfg("hydranet_big.yml")
    torch.manual_seed(6)
    net = HydraNet(cfgs).cuda().train()
    x = torch.randn(2, 3, 512, 1024, device=dev())             # stage 4: 8 x 16 pixels per image, the smallest grid of the fused node
    res, taken = [], []
    real = B.k_gemm_nt

    def counting(*a, **kw):
        if kw.get("estat") is not None and kw["estat"][0] == 3:
            taken[-1] += 1
        return real(*a, **kw)

    for on in (False, True):
        K.BN3_PARTS_FROM_DGRAD = on
        B.k_gemm_nt = counting
        taken.append(0)
        try:
            net.zero_grad(set_to_none=True)
            sd = {k: v.clone() for k, v in net.state_dict().items()}
            feats = net._backbone(x)
            net._flush_nbt()
            loss = sum((f.float() ** 2).mean() for f in feats)
            loss.backward()
            res.append(([f.detach().clone() for f in feats], {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
            net.load_state_dict(sd)
        finally:
            K.BN3_PARTS_FROM_DGRAD = True
            B.k_gemm_nt = real
    assert taken == [0, 9 + 13 + 3], taken                    # stage depths (1, 1, 4, 10, 14): the identity blocks of stages 2..4 (>64 ch)
    (f0, g0), (f1, g1) = res
    assert all(torch.equal(a, b) for a, b in zip(f0, f1))
    assert g0.keys() == g1.keys()
    for k in g0:
        if "stage_4.blocks.block_13." in k:
            assert torch.equal(g1[k], g0[k]), k
        else:
            close(g1[k], g0[k], 5e-3 if "stage_4.blocks.block_12." in k else 8e-2, k)


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_hip_adam_tracks_torch_adam(K, wd):
    """multitask_hydranet_amd.optim.Adam (one launch for all parameters) against torch.optim.Adam: same update rule and operation order;
    odd sizes, a late-starting parameter (its own step count), a changing learning rate, interchangeable state_dict"""
    from multitask_hydranet_amd.optim import Adam
    torch.manual_seed(1)
    shapes = [(936, 936, 1, 1), (7,), (3, 5), (1,), (64, 8, 3, 3), (1023,), (1025,)]
    pa = [torch.nn.Parameter(torch.randn(s, device=dev())) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = Adam(pa, 1e-2, weight_decay=wd)
    ob = torch.optim.Adam(pb, 1e-2, weight_decay=wd)
    for it in range(6):
        for o in (oa, ob):
            o.param_groups[0]["lr"] = 1e-2 / (1 + it)
        for a, b in zip(pa, pb):
            g = torch.randn_like(a) * (10.0 ** (it - 3))
            a.grad = None if (it < 2 and a.numel() == 7) else g.clone()          # no gradient in the first iterations: its step starts later
            b.grad = None if (it < 2 and b.numel() == 7) else g.clone()
        oa.step()
        ob.step()
        for a, b in zip(pa, pb):
            assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-3), (it, tuple(a.shape))
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k in sb:
        assert float(sa[k]["step"]) == float(sb[k]["step"])
        close(sa[k]["exp_avg"], sb[k]["exp_avg"], 1e-5, "exp_avg")
        close(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"], 1e-5, "exp_avg_sq")
    oc = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in pa], 1e-2, weight_decay=wd)
    import copy
    # (deep copies: Optimizer.load_state_dict keeps tensors that already have the parameter's dtype / device, i.e. it would ALIAS the moment
    # buffers of the optimizer the state came from, and stepping both would update them twice)
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))                            # interchangeable state
    # ... also for STEPPING: torch's Adam bumps every parameter's `step` on its own (a shared tensor would advance by #params per step),
    # and a load into the HIP optimizer after a step must not keep writing into the old moment buffers (cached job tables)
    pc = oc.param_groups[0]["params"]
    od = Adam([torch.nn.Parameter(p.detach().clone()) for p in pa], 1e-2, weight_decay=wd)
    pd = od.param_groups[0]["params"]
    for q in pd:
        q.grad = torch.zeros_like(q)
    od.step()                                                                     # builds the cached tables on a throw-away state
    for q, a in zip(pd, pa):
        q.data.copy_(a.detach())
    od.load_state_dict(copy.deepcopy(ob.state_dict()))
    for it in range(2):
        for a, b, c, d in zip(pa, pb, pc, pd):
            g = torch.randn_like(a)
            a.grad, b.grad, c.grad, d.grad = g.clone(), g.clone(), g.clone(), g.clone()
        for o in (oa, ob, oc, od):
            o.step()
        for a, b, c, d in zip(pa, pb, pc, pd):
            tol = 2e-6 * max(float(b.abs().max()), 1e-3)
            assert float((c - b).abs().max()) <= tol and float((d - b).abs().max()) <= tol and float((a - b).abs().max()) <= tol, (it, tuple(a.shape))
    for k, v in oc.state_dict()["state"].items():
        assert float(v["step"]) == float(ob.state_dict()["state"][k]["step"])


def test_direct_conv_unstageable_output_takes_the_32_cout_tile(K):
    """a bf16 output whose rows cannot leave through the LDS-staged epilogue (cout not a multiple of 8, or a channel-slice view that is
    not 16-byte aligned) must not reach the >= 64-cout instantiations (compact staged epilogue only): the host picks the 32-cout tile,
    whose generic epilogue stores 4 couts per lane directly.  Forward of Conv3x3(reflect_pad(x)) + bias + ELU, cout = 100."""
    n, c0, cout, h, w = 2, 64, 100, 12, 20
    x0 = rnd(n, c0, h, w)
    wt = rnd(cout, c0, 3, 3, scale=(9 * c0) ** -0.5)
    bs = rnd(cout, scale=0.1)
    wp, _ = K.pack_conv_weight(wt)
    out, _, _ = K.k_gemm_nt(nhwc(x0), None, 2, (n, h, w), wp, cout, K.kp32(c0), 9, bias=bs, act=3)
    y = F.elu(F.conv2d(F.pad(x0.bfloat16().float(), [1, 1, 1, 1], mode="reflect"), wt.bfloat16().float(), bs))
    close(nchw(out), y, ACT_TOL, "out")


@pytest.mark.parametrize("n,h,w,cin,cout", [(3, 16, 24, 64, 64), (2, 8, 16, 152, 376)])
def test_conv1x1_with_gate_folded_into_per_image_weights(K, n, h, w, cin, cout):
    """inference form of an XBlock's conv_block_3 (net/anynet.py:68-76): relu(W (b * gate_n) + bias + identity) with the SE gate folded into
    the packed weights per image (hn_scale_weight_gate + hn_conv_gemm_nt_imgw; h * w is a multiple of 128) == the gated activation through
    the shared weights (the switch off), and both == the fp32 composition."""
    b = rnd(n, cin, h, w)
    res = rnd(n, cout, h, w)
    wt = rnd(cout, cin, 1, 1, scale=cin ** -0.5)
    bias = rnd(cout, scale=0.1)
    gate = torch.rand(n, cin, device=b.device)
    wp, _ = K.pack_conv_weight(wt)
    outs = {}
    for on in (True, False):
        K.INFER_GATE_IN_WEIGHTS = on
        try:
            outs[on] = K.conv_infer(nhwc(b), wp, bias, cout, "1x1", 1, 1, res=nhwc(res), gate=gate)
        finally:
            K.INFER_GATE_IN_WEIGHTS = True
    ref = F.relu(F.conv2d(b.bfloat16().float() * gate.view(n, cin, 1, 1), wt.bfloat16().float(), bias) + res.bfloat16().float())
    close(nchw(outs[True]), ref, ACT_TOL, "gate in weights")
    close(nchw(outs[False]), ref, ACT_TOL, "gate on the activation")


def test_tower_pointwise_conv_with_level_batchnorm_epilogue(K):
    """hn_conv_gemm_nt_lvl: level-packed rows, out = swish(scale_l * (x W^T + bias) + shift_l) with the level's eval-mode BatchNorm rows
    (head_detect/detection.py:60-75) == the GEMM followed by hn_bn_act_levels, and == the fp32 composition per level."""
    import ctypes
    n, c, cout = 2, 64, 64
    hs, ws = (16, 8, 4), (24, 12, 6)
    geom = (n, hs, ws)
    nl, H, W, R, CNT = K._geom_arrays(geom)
    total = sum(R)
    x = torch.zeros(1, 1, total, c, device="cuda", dtype=torch.bfloat16)
    feats = [rnd(n, h, w, c) for h, w in zip(hs, ws)]
    off = 0
    for f, r in zip(feats, R):
        x[0, 0, off:off + f.numel() // c] = f.reshape(-1, c).to(torch.bfloat16)
        off += r
    wt = rnd(cout, c, 1, 1, scale=c ** -0.5)
    bias = rnd(cout, scale=0.1)
    coef = torch.zeros(nl, 4, cout, device="cuda")
    coef[:, 0] = torch.rand(nl, cout, device="cuda") + 0.5
    coef[:, 1] = torch.randn(nl, cout, device="cuda") * 0.2
    wp, _ = K.pack_conv_weight(wt)
    out = torch.empty(1, 1, total, cout, device="cuda", dtype=torch.bfloat16)
    K.lib().call("hn_conv_gemm_nt_lvl", x.data_ptr(), c, total, c, wp.data_ptr(), cout, K.kp32(c), bias.data_ptr(), 2, out.data_ptr(), cout,
                 coef.data_ptr(), nl, ctypes.addressof(R))
    off = 0
    for l, (f, r) in enumerate(zip(feats, R)):
        m = f.numel() // c
        z = f.reshape(-1, c).bfloat16().float() @ wt.view(cout, c).bfloat16().float().t() + bias
        y = z * coef[l, 0] + coef[l, 1]
        y = y * torch.sigmoid(y)
        close(out[0, 0, off:off + m].float(), y, ACT_TOL, f"level {l}")
        off += r


@pytest.mark.parametrize("cout,act", [(36, 0), (10, 4), (72, 0)])       # 4 = sigmoid (the classifier)
def test_head_output_conv_of_all_levels_in_one_launch(K, cout, act):
    """hn_conv_gemm_nt_lvlout: level-packed rows (ragged: the levels are padded to 128-row boundaries with rows that must not be stored) ->
    the per-image concatenation [N][sum_l H_l W_l][cout] in fp32, bit-identical to one hn_conv_gemm_nt per level with the per-image row
    mapping (the path it replaces), and every element of the output written exactly by it (poisoned buffer)."""
    import ctypes
    n, c = 3, 64
    hs, ws = (12, 6, 3, 2), (20, 10, 5, 3)
    geom = (n, hs, ws)
    nl, H, W, R, CNT = K._geom_arrays(geom)
    total = sum(R)
    x = torch.full((1, 1, total, c), 3.0, device="cuda", dtype=torch.bfloat16)     # alignment rows: garbage that must not leak
    feats = [rnd(n, h, w, c) for h, w in zip(hs, ws)]
    off = 0
    for f, r in zip(feats, R):
        x[0, 0, off:off + f.numel() // c] = f.reshape(-1, c).to(torch.bfloat16)
        off += r
    wt = rnd(cout, c, 1, 1, scale=c ** -0.5)
    bias = rnd(cout, scale=0.1)
    wp, _ = K.pack_conv_weight(wt)
    rows_total = sum(h * w for h, w in zip(hs, ws))
    ldc, img_stride = cout, rows_total * cout
    out = torch.full((n, rows_total, cout), float("nan"), device="cuda")
    K.lib().call("hn_conv_gemm_nt_lvlout", x.data_ptr(), c, total, c, wp.data_ptr(), cout, K.kp32(c), bias.data_ptr(), act, out.data_ptr(), ldc,
                 img_stride, n, nl, ctypes.addressof(H), ctypes.addressof(W), K.LEVEL_ALIGN)
    ref = torch.full((n, rows_total, cout), float("nan"), device="cuda")
    off = 0
    for v, h, w in zip(K.level_views(x, geom), hs, ws):
        K.k_gemm_nt(v, None, 0, (n, h, w), wp, cout, K.kp32(c), 1, bias=bias, act=act, out=ref.view(-1)[off * ldc:], out_f32=True, ldc=ldc,
                    rpi=h * w, img_stride=img_stride)
        off += h * w
    assert not torch.isnan(ref).any()
    assert torch.equal(out, ref)



@pytest.mark.parametrize("n,h,w,c,cs,act", [(16, 8, 16, 936, 234, 1), (4, 16, 32, 376, 94, 1), (3, 10, 10, 152, 16, 1), (2, 20, 20, 64, 6, 0),
                                            (2, 18, 30, 936, 234, 0), (5, 7, 9, 24, 300, 1)])
def test_se_excite_and_gated_apply_in_one_launch(K, n, h, w, c, cs, act):
    """hn_se_gate_apply (second SE layer + Sigmoid + BatchNorm / ReLU / product, net/anynet.py:44-48,68-69) against the two-launch
    composition hn_se_mlp_fwd's second layer + hn_bn_apply_fused(gate=...) (same summation order: equal to an ulp), and against fp32 torch"""
    from multitask_hydranet_amd.ops import core as C
    lib, ptr, ld = C.lib, C.ptr, C.ld
    hw, m = h * w, n * h * w
    z = (torch.randn(n, h, w, c, device=dev()) * 2).bfloat16()
    coef = torch.stack([torch.rand(c, device=dev()) + 0.5, torch.randn(c, device=dev()) * 0.3, torch.zeros(c, device=dev()),
                        torch.ones(c, device=dev())]).contiguous()
    pooled = torch.rand(n, c, device=dev())
    w1 = torch.randn(cs, c, device=dev()) / c ** 0.5
    b1 = torch.randn(cs, device=dev()) * 0.2
    w2 = torch.randn(c, cs, device=dev()) / cs ** 0.5
    b2 = torch.randn(c, device=dev()) * 0.2
    hid = torch.empty(n, cs, device=dev())
    gate2 = torch.empty(n, c, device=dev())
    lib().call("hn_se_mlp_fwd", ptr(pooled), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(hid), ptr(gate2), n, c, cs)     # both layers: the composition
    for use_coef in (True, False):
        cf = coef if use_coef else None
        gate1 = torch.empty(n, c, device=dev())
        out1 = torch.empty_like(z)
        lib().call("hn_se_gate_apply", ptr(z), ld(z), ptr(cf), act, ptr(hid), ptr(w2), ptr(b2), ptr(gate1), ptr(out1), ld(out1), n, hw, c, cs)
        rb = lib().query("hn_fused_row_block", m, c, hw, 0, 0)
        out2 = torch.empty_like(z)
        lib().call("hn_bn_apply_fused", ptr(z), ld(z), m, c, None, None, 0, m, None, None, 0.0, 0.0, None, None, ptr(cf), None, 0, act,
                   ptr(out2), ld(out2), None, ptr(gate2), hw, rb)
        torch.cuda.synchronize()
        assert float((gate1 - gate2).abs().max()) <= 2.5e-7              # an ulp of a value in (0.5, 1): the compiler contracts the two
        close(out1, out2, 2.0 ** -7, "out vs composition")               # dot products differently; one bf16 ulp where a product rounds the other way
        g = torch.sigmoid(hid @ w2.t() + b2)
        v = z.float() * (coef[0] if use_coef else 1.0) + (coef[1] if use_coef else 0.0)
        v = torch.relu(v) if act == 1 else v
        ref = v.bfloat16().float() * g[:, None, None, :]
        close(gate1, g, 1e-5, "gate")
        close(out1, ref, ACT_TOL, "out")

