"""The optional HIP backend of the denoisers' plain conv stacks (csrc/kernels_conv.hip, `Denoiser(backend='hip')`): the
64 -> 64 conv3x3 (+ bias, + skip, + ReLU) implicit GEMM on the fp32 matrix cores against a plain PyTorch reference of the
same op.  fp32 MFMA is an exact f32 fma chain over the 576 products of an output value, so one layer sits at ~5e-7 from the
float64 result -- the same distance as MIOpen's own fp32 kernels; tolerances: one layer <= 2e-6 (vs float64), whole networks
<= 1e-5 (vs the PyTorch / MIOpen forward with the same weights).

The same layer in split-half arithmetic on the f16 matrix cores (csrc/kernels_conv_f16x3.hip, `backend='hip_f16x3'`: every
float32 operand as two halves, three exact-product v_mfma_f32_16x16x32_f16 per product, float32 accumulation) is held to the
SAME tolerances by the same tests (parameter `math`), plus its own: magnitudes from 1e-6 to 1e4, and a loud failure beyond the
half range."""
import ctypes as C
C_ = C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    import torch.nn.functional as F
    import pnp_admm_cnc_mri_amd as P
    from pnp_admm_cnc_mri_amd import _lib, denoisers
    assert torch.cuda.is_available() and _lib.device_count() >= 1
    return dict(torch=torch, F=F, P=P, L=_lib.lib(), lib=_lib, D=denoisers)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def _conv(env, x_nhwc, w_oihw, bias, skip, relu, dilation=1, math='f32'):
    torch, L, lib = env['torch'], env['L'], env['lib']
    pack, conv = ((L.pnp_conv3x3_c64_pack, L.pnp_conv3x3_c64_nhwc) if math == 'f32' else
                  (L.pnp_conv3x3_c64_pack_f16x3, L.pnp_conv3x3_c64_nhwc_f16x3))
    y = torch.empty_like(x_nhwc)
    n, H, W, _ = x_nhwc.shape
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    w9 = torch.empty(9 * 64 * 64, device='cuda')
    lib.check(pack(C.c_void_p(torch.cuda.current_stream().cuda_stream), p(w_oihw.contiguous()), p(w9)))
    lib.check(conv(C.c_void_p(torch.cuda.current_stream().cuda_stream), p(x_nhwc), p(w9), p(bias), p(skip), p(y),
                   n, H, W, 1 if relu else 0, dilation))
    return y


@pytest.mark.parametrize('n,H,W', [(3, 136, 136), (2, 16, 16), (1, 8, 16), (2, 5, 23), (1, 1, 1), (5, 128, 128)])
@pytest.mark.parametrize('variant', ['bias_relu', 'plain', 'skip_relu'])
@pytest.mark.parametrize('math', ['f32', 'f16x3'])
def test_conv3x3_c64_against_pytorch(env, n, H, W, variant, math):
    """tiles of 8 x 16 pixels: shapes that are no multiple of the tile, smaller than one tile, a single pixel; more tiles than
    resident workgroups (5 x 128 x 128 = 640 tiles on 512 persistent workgroups: the loop's second trip)"""
    torch, F = env['torch'], env['F']
    g = torch.Generator(device='cuda').manual_seed(1000 * n + 10 * H + W)
    x = torch.randn(n, 64, H, W, device='cuda', generator=g)
    w = torch.randn(64, 64, 3, 3, device='cuda', generator=g) * (2.0 / 576) ** 0.5
    b = torch.randn(64, device='cuda', generator=g) * 0.1 if variant != 'plain' else None
    sk = torch.randn(n, 64, H, W, device='cuda', generator=g) if variant == 'skip_relu' else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if sk is not None:
        ref = ref + sk.double()
    if variant != 'plain':
        ref = F.relu(ref)
    xn = x.permute(0, 2, 3, 1).contiguous()
    skn = None if sk is None else sk.permute(0, 2, 3, 1).contiguous()
    y = _conv(env, xn, w, b, skn, variant != 'plain', math=math)
    assert _rel(y.permute(0, 3, 1, 2), ref) <= 2e-6
    # asymmetric weights + an identity-like input catch a transposed tap or channel map: one hot input channel / pixel
    x1 = torch.zeros(1, 64, H, W, device='cuda')
    x1[0, 7, H // 2, W // 2] = 1.0
    y1 = _conv(env, x1.permute(0, 2, 3, 1).contiguous(), w, None, None, False, math=math)
    assert _rel(y1.permute(0, 3, 1, 2), F.conv2d(x1.double(), w.double(), padding=1)) <= 1e-6


@pytest.mark.parametrize('dilation', [2, 3, 4])
@pytest.mark.parametrize('n,H,W', [(3, 136, 136), (2, 5, 23), (1, 1, 1), (5, 128, 128)])
@pytest.mark.parametrize('math', ['f32', 'f16x3'])
def test_dilated_conv3x3_c64_against_pytorch(env, n, H, W, dilation, math):
    """IRCNN's layers (models/network_dncnn.py:87-101): dilation d = 2, 3, 4 with zero padding d -- a halo of d pixels, taps d apart;
    images smaller than the halo, not a multiple of the tile, and more tiles than resident workgroups (one per unit at d = 3, 4)"""
    torch, F = env['torch'], env['F']
    g = torch.Generator(device='cuda').manual_seed(100 * dilation + n + H)
    x = torch.randn(n, 64, H, W, device='cuda', generator=g)
    w = torch.randn(64, 64, 3, 3, device='cuda', generator=g) * (2.0 / 576) ** 0.5
    b = torch.randn(64, device='cuda', generator=g) * 0.1
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=dilation, dilation=dilation))
    y = _conv(env, x.permute(0, 2, 3, 1).contiguous(), w, b, None, True, dilation, math)
    assert _rel(y.permute(0, 3, 1, 2), ref) <= 2e-6
    x1 = torch.zeros(1, 64, H, W, device='cuda')
    x1[0, 5, H // 2, W // 2] = 1.0                                   # one-hot input, asymmetric weights: every tap must land where PyTorch puts it
    y1 = _conv(env, x1.permute(0, 2, 3, 1).contiguous(), w, None, None, False, dilation, math)
    assert _rel(y1.permute(0, 3, 1, 2), F.conv2d(x1.double(), w.double(), padding=dilation, dilation=dilation)) <= 1e-6


@pytest.mark.parametrize('xs,ws', [('mixed', 1.0), (1e-3, 1e-3), (1.0, 1.0), (300.0, 0.01), (1e4, 1.0), (1.0, 50.0)])
def test_f16x3_holds_float32_accuracy_over_magnitudes(env, xs, ws):
    """x = hi + lo / 2048 with both parts halves: relative precision 2^-22 for |x| >= 2^-14 (hi a normal half), an ABSOLUTE
    precision of 2^-36 = 1.5e-11 below (subnormal halves, which the matrix cores take as they are; the factor 2048 is the largest
    that keeps `lo` inside the half range for every |x| <= 65504).  Over five decades of operand magnitude -- and with tiny
    values mixed into ordinary ones -- the distance from the float64 result is no larger than the float32-MFMA kernel's."""
    torch, F = env['torch'], env['F']
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(2, 64, 48, 40, device='cuda', generator=g)
    if xs == 'mixed':                                                # one value in eight six decades below the others
        x = x * torch.where(torch.rand(x.shape, device='cuda', generator=g) < 0.125, 1e-6, 1.0)
    else:
        x = x * xs
    w = torch.randn(64, 64, 3, 3, device='cuda', generator=g) * (2.0 / 576) ** 0.5 * ws
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    e16 = _rel(_conv(env, xn, w, None, None, False, math='f16x3').permute(0, 3, 1, 2), ref)
    e32 = _rel(_conv(env, xn, w, None, None, False, math='f32').permute(0, 3, 1, 2), ref)
    assert e16 <= 5e-7 and e16 <= 1.25 * e32, (e16, e32)


@pytest.mark.parametrize('ch,n,H,W', [(128, 3, 40, 56), (256, 2, 24, 24), (512, 2, 32, 32), (192, 1, 5, 23), (128, 40, 64, 64), (1024, 1, 9, 9)])
@pytest.mark.parametrize('variant', ['bias_relu', 'skip'])
def test_f16x3_wide_layers_against_pytorch(env, ch, n, H, W, variant):
    """C -> C channels, C a multiple of 64 (DRUNet's residual blocks at 128 / 256 / 512 channels, models/network_unet.py:36-58):
    items (tile, 64 output channels) x chunks of 64 input channels.  Shapes off the tile grid, more items than resident
    workgroups (40 x 32 tiles x 2 blocks), a channel count that is no power of two, the largest count accepted."""
    torch, F, L, lib = env['torch'], env['F'], env['L'], env['lib']
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    g = torch.Generator(device='cuda').manual_seed(ch + 10 * n + H)
    x = torch.randn(n, H, W, ch, device='cuda', generator=g)
    w = torch.randn(ch, ch, 3, 3, device='cuda', generator=g) * (2.0 / (9 * ch)) ** 0.5
    b = torch.randn(ch, device='cuda', generator=g) * 0.1 if variant == 'bias_relu' else None
    sk = torch.randn(n, H, W, ch, device='cuda', generator=g) if variant == 'skip' else None
    pk = torch.empty(9 * ch * ch, device='cuda')
    lib.check(L.pnp_conv3x3_pack_f16x3(s, p(w), p(pk), ch))
    y = torch.empty_like(x)
    lib.check(L.pnp_conv3x3_nhwc_f16x3(s, p(x), p(pk), p(b), p(sk), p(y), n, ch, H, W, 1 if variant == 'bias_relu' else 0))
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None if b is None else b.double(), padding=1)
    ref = F.relu(ref) if variant == 'bias_relu' else ref + sk.permute(0, 3, 1, 2).double()
    assert _rel(y.permute(0, 3, 1, 2), ref) <= 2e-6
    x1 = torch.zeros(1, H, W, ch, device='cuda')
    x1[0, H // 2, W // 2, ch - 3] = 1.0                             # one hot value in the LAST chunk of input channels
    y1 = torch.empty_like(x1)
    lib.check(L.pnp_conv3x3_nhwc_f16x3(s, p(x1), p(pk), None, None, p(y1), 1, ch, H, W, 0))
    assert _rel(y1.permute(0, 3, 1, 2), F.conv2d(x1.permute(0, 3, 1, 2).double(), w.double(), padding=1)) <= 1e-6
    for bad_c in (0, 32, 96, 1088):
        with pytest.raises(lib.PnpError):
            lib.check(L.pnp_conv3x3_nhwc_f16x3(s, p(x), p(pk), None, None, p(y), n, bad_c, H, W, 0))
        with pytest.raises(lib.PnpError):
            lib.check(L.pnp_conv3x3_pack_f16x3(s, p(w), p(pk), bad_c))


def test_f16x3_precision_floor_of_uniformly_tiny_operands(env):
    """the documented floor: a tensor whose EVERY value is ~1e-6 (far below 2^-14) is carried to 2^-36 absolute, i.e. ~1e-5 of
    its own scale -- the one regime where f16x3 is not float32-like.  (A denoiser's activations are O(0.01 .. 1).)"""
    torch, F = env['torch'], env['F']
    g = torch.Generator(device='cuda').manual_seed(6)
    x = torch.randn(2, 64, 48, 40, device='cuda', generator=g) * 1e-6
    w = torch.randn(64, 64, 3, 3, device='cuda', generator=g) * (2.0 / 576) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    y = _conv(env, x.permute(0, 2, 3, 1).contiguous(), w, None, None, False, math='f16x3').permute(0, 3, 1, 2)
    assert 1e-6 < _rel(y, ref) <= 3e-5
    assert float((y.double() - ref).abs().max()) <= 576 ** 0.5 * 6 * 2.0 ** -36 * float(w.abs().max())


def test_f16x3_is_loud_outside_the_half_range(env):
    """an activation beyond +-65504 has no half representation: the outputs it reaches are inf / NaN, never a plausible number;
    the Python backend refuses such WEIGHTS when it packs them"""
    torch, D = env['torch'], env['D']
    x = torch.zeros(1, 64, 16, 16, device='cuda')
    x[0, 3, 8, 8] = 7e4
    w = torch.randn(64, 64, 3, 3, device='cuda') * 0.05
    y = _conv(env, x.permute(0, 2, 3, 1).contiguous(), w, None, None, False, math='f16x3').permute(0, 3, 1, 2)
    assert not torch.isfinite(y[0, :, 7:10, 7:10]).any()
    far = torch.ones(16, 16, dtype=torch.bool, device='cuda')
    far[7:10, 7:10] = False
    assert torch.isfinite(y[0][:, far]).all()
    net = D.DnCNN(nb=4)
    net.load_state_dict(D.seeded_state_dict(net, 1))
    with torch.no_grad():
        net.model[2].weight[0, 0, 0, 0] = 1e5
    net.backend = 'hip_f16x3'
    with pytest.raises(ValueError):
        net.cuda()(torch.rand(1, 1, 16, 16, device='cuda'))


@pytest.mark.parametrize('n,H,W', [(3, 136, 136), (2, 16, 16), (2, 5, 23), (1, 1, 1), (4, 128, 128)])
def test_head_and_tail_layers_against_pytorch(env, n, H, W):
    """the direct kernels of the stacks' first (cin <= 8 -> 64, NCHW -> NHWC, + bias + ReLU) and last layer (64 -> cout <= 4, NHWC ->
    NCHW, + bias): every channel count the reference's models use (DnCNN 1, FDnCNN 2, FFDNet 5 in; 1 and 4 out) and the rest"""
    torch, F, L, lib = env['torch'], env['F'], env['L'], env['lib']
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    g = torch.Generator(device='cuda').manual_seed(7 * n + H + W)
    for cin in (1, 2, 5, 8):
        x = torch.randn(n, cin, H, W, device='cuda', generator=g)
        w = torch.randn(64, cin, 3, 3, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(64, device='cuda', generator=g) * 0.1
        for relu, bias in ((1, b), (0, None)):
            y = torch.empty(n, H, W, 64, device='cuda')
            lib.check(L.pnp_conv3x3_head_nhwc(s, p(x), p(w), p(bias), p(y), n, cin, H, W, relu))
            ref = F.conv2d(x.double(), w.double(), None if bias is None else bias.double(), padding=1)
            ref = F.relu(ref) if relu else ref
            assert _rel(y.permute(0, 3, 1, 2), ref) <= 2e-6, (cin, relu)
    xn = torch.randn(n, H, W, 64, device='cuda', generator=g)
    for cout in (1, 2, 3, 4):
        w = torch.randn(cout, 64, 3, 3, device='cuda', generator=g) * (2.0 / 576) ** 0.5
        b = torch.randn(cout, device='cuda', generator=g) * 0.1
        for bias in (b, None):
            y = torch.empty(n, cout, H, W, device='cuda')
            ref = F.conv2d(xn.permute(0, 3, 1, 2).double(), w.double(), None if bias is None else bias.double(), padding=1)
            for tail in (L.pnp_conv3x3_tail_nchw, L.pnp_conv3x3_tail_nchw_f16x3):       # vector units; f16x3 on the matrix cores
                y = torch.full((n, cout, H, W), float('nan'), device='cuda')
                lib.check(tail(s, p(xn), p(w), p(bias), p(y), n, cout, H, W))
                assert _rel(y, ref) <= 2e-6, cout
    with pytest.raises(lib.PnpError):
        lib.check(L.pnp_conv3x3_head_nhwc(s, p(xn), p(xn), None, p(xn), n, 9, H, W, 0))       # cin > 8
    with pytest.raises(lib.PnpError):
        lib.check(L.pnp_conv3x3_tail_nchw(s, p(xn), p(xn), None, p(xn), n, 5, H, W))          # cout > 4
    with pytest.raises(lib.PnpError):
        lib.check(L.pnp_conv3x3_tail_nchw_f16x3(s, p(xn), p(xn), None, p(xn), n, 5, H, W))


@pytest.mark.parametrize('name,backend', [('ffdnet_gray', 'hip'), ('ffdnet_gray', 'hip_f16x3'), ('drunet_gray', 'hip_f16x3')])
def test_hip_backend_makes_no_miopen_call(env, name, backend):
    """FFDNet with a HIP backend: head, 13 body layers and tail all run on libpnpmri.so; DRUNet with 'hip_f16x3': first / last layer, the
    28 residual blocks, the 2 x 2 strided and transposed convolutions and the skip sums -- torch.nn.Conv2d / ConvTranspose2d are never
    reached (checked by making both raise for the duration of the call), and the process-global cudnn.benchmark flag is left alone."""
    torch, D, F = env['torch'], env['D'], env['F']
    net, nlm, sched = D.build(name)
    net.load_state_dict(D.seeded_state_dict(net, 3))
    sig = torch.tensor([20.0 / 255]) if sched else None
    den = D.Denoiser(name, net.eval(), nlm, sigmas=sig, backend=backend, miopen_find='auto').to('cuda')
    x = torch.rand(17, 1, 64, 64, device='cuda')                   # >= 16 slices: 'auto' would switch MIOpen's find mode on for a forward that needs it
    ref = D.Denoiser(name, net, nlm, sigmas=sig, backend='torch', miopen_find=False).to('cuda')(x, 0).clone()
    for m in [net] + list(net.modules()):
        if hasattr(m, 'backend'):
            m.backend = backend
    orig, orig_t, seen = torch.nn.Conv2d.forward, torch.nn.ConvTranspose2d.forward, []

    def boom(self, inp, *a):
        seen.append(torch.backends.cudnn.benchmark)
        raise AssertionError('a PyTorch convolution was called')
    torch.nn.Conv2d.forward = torch.nn.ConvTranspose2d.forward = boom
    before = torch.backends.cudnn.benchmark
    try:
        out = den(x, 0)
    finally:
        torch.nn.Conv2d.forward, torch.nn.ConvTranspose2d.forward = orig, orig_t
    assert torch.backends.cudnn.benchmark == before and not seen
    assert _rel(out, ref) <= 1e-5, _rel(out, ref)


@pytest.mark.parametrize('up,C,n,H,W', [(0, 64, 3, 40, 56), (0, 128, 2, 24, 24), (0, 256, 2, 16, 32), (0, 64, 1, 2, 2), (0, 192, 1, 6, 34),
                                       (1, 128, 3, 20, 28), (1, 256, 2, 12, 12), (1, 512, 2, 8, 16), (1, 128, 1, 1, 1), (1, 384, 1, 3, 17),
                                       (0, 64, 8, 128, 128), (1, 128, 8, 64, 64)])
@pytest.mark.parametrize('with_x2', [False, True])
def test_pix2x2_layers_against_pytorch(env, up, C, n, H, W, with_x2):
    """DRUNet's scale changes on csrc/kernels_pix2x2_f16x3.hip: Conv2d(C, 2C, 2, 2, 0) and ConvTranspose2d(C, C/2, 2, 2, 0), bias-free, NHWC,
    with and without the second input that is added while the operand is staged -- against the float64 PyTorch operator on the same
    data, at the tolerance of the 3 x 3 layers (one layer <= 2e-6); tiles that overhang the image, one-tile and one-pixel images."""
    torch, F, L, lib = env['torch'], env['F'], env['L'], env['lib']
    g = torch.Generator(device='cuda').manual_seed(100 * C + 10 * H + up)
    x = torch.randn(n, H, W, C, device='cuda', generator=g)
    x2 = torch.randn(n, H, W, C, device='cuda', generator=g) if with_x2 else None
    if up:
        w = torch.randn(C, C // 2, 2, 2, device='cuda', generator=g) * (1.0 / C) ** 0.5
        ref = F.conv_transpose2d((x + x2 if with_x2 else x).double().permute(0, 3, 1, 2), w.double(), stride=2).permute(0, 2, 3, 1)
        y = torch.full((n, 2 * H, 2 * W, C // 2), float('nan'), device='cuda')
    else:
        w = torch.randn(2 * C, C, 2, 2, device='cuda', generator=g) * (0.25 / C) ** 0.5
        ref = F.conv2d((x + x2 if with_x2 else x).double().permute(0, 3, 1, 2), w.double(), stride=2).permute(0, 2, 3, 1)
        y = torch.full((n, H // 2, W // 2, 2 * C), float('nan'), device='cuda')
    s = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else C_.c_void_p(t.data_ptr())
    wp = torch.empty(w.numel(), device='cuda')
    lib.check(L.pnp_conv2x2_pack_f16x3(s, p(w), p(wp), C, up))
    lib.check((L.pnp_convT2x2s2_nhwc_f16x3 if up else L.pnp_conv2x2s2_nhwc_f16x3)(s, p(x), p(x2), p(wp), p(y), n, C, H, W))
    assert bool(torch.isfinite(y).all())                            # every output element was written
    assert _rel(y, ref) <= 2e-6, _rel(y, ref)


def test_pix2x2_argument_errors(env):
    torch, L, lib = env['torch'], env['L'], env['lib']
    s = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.zeros(1, 6, 6, 128, device='cuda')
    p = lambda t: None if t is None else C_.c_void_p(t.data_ptr())
    for fn, args in ((L.pnp_conv2x2s2_nhwc_f16x3, (s, p(x), None, p(x), p(x), 1, 128, 6, 6)),             # y aliases x
                     (L.pnp_conv2x2s2_nhwc_f16x3, (s, p(x), None, p(x), None, 1, 128, 6, 6)),             # null y
                     (L.pnp_conv2x2s2_nhwc_f16x3, (s, p(x), None, p(x), p(x[:, :5]), 1, 128, 5, 6)),      # odd H
                     (L.pnp_conv2x2s2_nhwc_f16x3, (s, p(x), None, p(x), p(x[:, :4]), 1, 100, 6, 6)),      # C no multiple of 64
                     (L.pnp_convT2x2s2_nhwc_f16x3, (s, p(x), None, p(x), p(x[:, :4]), 1, 64, 6, 6)),      # transposed: C must be a multiple of 128
                     (L.pnp_conv2x2_pack_f16x3, (s, p(x), p(x), 128, 0))):                                # aliased
        with pytest.raises(lib.PnpError):
            lib.check(fn(*args))
    # round 6 (advisor): the skip input x2 must not be the output either, and images beyond 2 GiB are an argument error with a message
    # (PNP_E_ARG), not a bare HIP error from the launcher
    y = torch.zeros(1, 3, 3, 256, device='cuda')
    w4 = torch.zeros(4, 64, 3, 3, device='cuda')
    x64 = torch.zeros(1, 6, 6, 64, device='cuda')
    y1 = torch.zeros(1, 4, 6, 6, device='cuda')
    big = 1 << 12                                             # 4096 x 4096 x 64 floats = 4 GiB: only the sizes are checked, nothing is touched
    for fn, args in ((L.pnp_conv2x2s2_nhwc_f16x3, (s, p(x), p(y), p(x), p(y), 1, 128, 6, 6)),              # y aliases x2
                     (L.pnp_conv3x3_tail_add_nchw_f16x3, (s, p(x64), p(y1), p(w4), None, p(y1), 1, 4, 6, 6)),      # y aliases x2
                     (L.pnp_conv3x3_tail_add_nchw_f16x3, (s, p(x64), p(x64), p(w4), None, p(y1), 1, 4, big, big)),
                     (L.pnp_ffdnet_tail_f16x3, (s, p(x64), p(w4), None, p(y1), 1, 2 * big, 2 * big)),
                     (L.pnp_ffdnet_head_nhwc, (s, p(y1), p(y1), 0, p(w4), None, p(x64), 1, 2 * big, 2 * big, 1))):
        with pytest.raises(lib.PnpError) as e:
            lib.check(fn(*args))
        assert e.value.code == -1, (fn.__name__, e.value.code)              # PNP_E_ARG (include/pnp_mri.h)


def test_relayout_round_trip_and_argument_errors(env):
    torch, L, lib = env['torch'], env['L'], env['lib']
    x = torch.randn(3, 64, 37, 21, device='cuda')
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    xn = torch.empty(3, 37, 21, 64, device='cuda')
    lib.check(L.pnp_relayout_c64(s, C.c_void_p(x.data_ptr()), C.c_void_p(xn.data_ptr()), 3, 37, 21, 1))
    assert torch.equal(xn, x.permute(0, 2, 3, 1).contiguous())
    back = torch.empty_like(x)
    lib.check(L.pnp_relayout_c64(s, C.c_void_p(xn.data_ptr()), C.c_void_p(back.data_ptr()), 3, 37, 21, 0))
    assert torch.equal(back, x)
    with pytest.raises(lib.PnpError):                        # y must not alias x: a tile reads its neighbours' halo
        lib.check(L.pnp_conv3x3_c64_nhwc(s, C.c_void_p(xn.data_ptr()), C.c_void_p(xn.data_ptr()), None, None, C.c_void_p(xn.data_ptr()), 3, 37, 21, 0, 1))
    with pytest.raises(lib.PnpError):
        lib.check(L.pnp_conv3x3_c64_nhwc(s, None, None, None, None, None, 1, 8, 8, 0, 1))
    y = torch.empty_like(xn)
    with pytest.raises(lib.PnpError):                        # dilation outside 1..4
        lib.check(L.pnp_conv3x3_c64_nhwc(s, C.c_void_p(xn.data_ptr()), C.c_void_p(xn.data_ptr()), None, None, C.c_void_p(y.data_ptr()), 3, 37, 21, 0, 5))
    for bad in ((xn, xn, xn, 1), (None, xn, y, 1), (xn, xn, y, 0)):          # the f16x3 entry point checks the same things
        a_, w_, y_, d_ = bad
        with pytest.raises(lib.PnpError):
            lib.check(L.pnp_conv3x3_c64_nhwc_f16x3(s, None if a_ is None else C.c_void_p(a_.data_ptr()), C.c_void_p(w_.data_ptr()), None, None,
                                                   C.c_void_p(y_.data_ptr()), 3, 37, 21, 0, d_))


@pytest.mark.parametrize('hip', ['hip', 'hip_f16x3'])
@pytest.mark.parametrize('name', ['ffdnet_gray', 'dncnn_15', 'fdncnn_gray', 'dncnn_gray_blind', 'drunet_gray', 'ircnn_gray'])
def test_hip_backend_matches_the_pytorch_forward(env, name, hip):
    """Same seeded weights, `Denoiser(backend='hip')` against `backend='torch'` (MIOpen) on 5 slices of 256 x 256 -- and the
    parameters are still the module's own: a load_state_dict after the first call reaches the kernel."""
    torch, D = env['torch'], env['D']
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.rand(5, 1, 256, 256, device='cuda', generator=g)
    noises = (np.random.default_rng(3).standard_normal((256, 256)) + 1j * np.random.default_rng(4).standard_normal((256, 256))) * 5
    outs = {}
    for backend in ('torch', hip):
        net, nlm, _ = D.build(name)
        net.load_state_dict(D.seeded_state_dict(net, 11))
        sig = torch.tensor([30.0 / 255, 20.0 / 255]) if name.startswith(('drunet', 'ircnn')) else None
        den = D.Denoiser(name, net.eval(), nlm, sigmas=sig, noises=noises, backend=backend, miopen_find=False).to('cuda')
        outs[backend] = den(x, 0).clone()
        if backend == hip:
            net.load_state_dict(D.seeded_state_dict(net, 12))
            again = den(x, 0).clone()
            net2, _, _ = D.build(name)
            net2.load_state_dict(D.seeded_state_dict(net2, 12))
            ref2 = D.Denoiser(name, net2.eval(), nlm, sigmas=sig, noises=noises, miopen_find=False).to('cuda')(x, 0)
            assert _rel(again, ref2) <= 1e-5 and _rel(again, outs[hip]) > 1e-3
    assert _rel(outs[hip], outs['torch']) <= 1e-5, _rel(outs[hip], outs['torch'])
    assert torch.isfinite(outs[hip]).all()


@pytest.mark.parametrize('hip', ['hip', 'hip_f16x3'])
def test_pnp_entry_point_with_the_hip_backend(env, tmp_path, hip):
    """PNP_ADMM_CNC_D(..., cnn_backend='hip') -- FFDNet, S6:573 presets, 6 iterations on 3 synthetic slices -- ends where the
    PyTorch / MIOpen backend ends (the loop contracts: 2e-5 covers the 12 forwards' accumulated difference)."""
    P, D = env['P'], env['D']
    from pnp_admm_cnc_mri_amd import synthetic as S, solvers_pnp as SP
    mask = S.reference_masks()['Q_Radial30'].astype(np.uint8)
    img, noise = S.batch(300, 3)
    res = {}
    for backend in ('torch', hip):
        net, _, _ = D.build('ffdnet_gray')
        net.load_state_dict(D.seeded_state_dict(net, 1))
        opts = dict(SP.PRESETS['PNP_ADMM_CNC_D']['ffdnet'], iter_num=6)
        out, _ = SP.PNP_ADMM_CNC_D('ffdnet_gray', mask, noise[0], images=img, model=net, results=str(tmp_path), cnn_backend=backend,
                                   miopen_find=False, **opts)
        res[backend] = np.stack([out[b] for b in range(3)])
    err = np.linalg.norm(res[hip] - res['torch']) / np.linalg.norm(res['torch'])
    assert err <= 2e-5, err


@pytest.mark.parametrize('name,backend', [('ffdnet_gray', 'hip_f16x3'), ('ffdnet_gray', 'torch'), ('dncnn_15', 'hip'), ('drunet_gray', 'hip_f16x3')])
def test_graph_replay_of_a_forward_equals_the_eager_forward(env, name, backend):
    """`Denoiser(graph=True)`: the forward of one call (the reference's own usage is ONE slice per call, S6:231) captured into a HIP graph
    and replayed -- the same kernels on the same data: equal results for new inputs, for another iteration's sigma (DRUNet), for
    another batch size (a second graph), and after the weights changed (re-captured)."""
    torch, D = env['torch'], env['D']
    g = torch.Generator(device='cuda').manual_seed(21)
    noises = (np.random.default_rng(3).standard_normal((256, 256)) + 1j * np.random.default_rng(4).standard_normal((256, 256))) * 5
    sig = torch.tensor([30.0 / 255, 20.0 / 255, 10.0 / 255]) if name.startswith('drunet') else None
    net, nlm, _ = D.build(name)
    net.load_state_dict(D.seeded_state_dict(net, 5))
    eager = D.Denoiser(name, net.eval(), nlm, sigmas=sig, noises=noises, backend=backend, miopen_find=False).to('cuda')
    graph = D.Denoiser(name, net, nlm, sigmas=sig, noises=noises, backend=backend, miopen_find=False, graph=True).to('cuda')
    # stacks that run on libpnpmri.so alone replay bit for bit; where MIOpen takes part (the PyTorch backend, DRUNet's strided and
    # transposed convolutions) it may choose another kernel under capture
    tol = 0.0 if (backend != 'torch' and not name.startswith('drunet')) else 1e-5
    for it, B in ((0, 1), (1, 1), (2, 1), (1, 3), (0, 1)):
        x = torch.rand(B, 1, 256, 256, device='cuda', generator=g)
        a, b = eager(x, it).clone(), graph(x, it).clone()
        assert _rel(b, a) <= tol, (it, B, _rel(b, a))
    assert len(graph._graphs) == 2
    net.load_state_dict(D.seeded_state_dict(net, 6))             # the graph holds the OLD packed weights: it must be captured again
    x = torch.rand(1, 1, 256, 256, device='cuda', generator=g)
    e2 = _rel(graph(x, 0), eager(x, 0))
    assert e2 <= tol, e2
    big = D.Denoiser(name, net, nlm, sigmas=sig, noises=noises, backend=backend, miopen_find=False, graph=True, cnn_batch=2).to('cuda')
    x = torch.rand(5, 1, 256, 256, device='cuda', generator=g)   # more slices than one forward takes: the eager path
    assert _rel(big(x, 0), eager(x, 0)) <= max(tol, 1e-6) and not big._graphs      # (MIOpen: other kernels for other batch sizes)


def test_graph_and_eager_calls_of_one_denoiser_each_use_their_own_sigma(env):
    """A DRUNet `Denoiser(graph=True)` serves calls of at most cnn_batch slices from the graph (noise level = a device scalar refreshed per
    call) and larger ones eagerly; the eager calls must read iteration i's own sigma, not the scalar the last graph call left behind
    -- neither after a graph call with another i, nor with the graph switched off afterwards."""
    torch, D = env['torch'], env['D']
    g = torch.Generator(device='cuda').manual_seed(5)
    sig = torch.tensor([40.0 / 255, 25.0 / 255, 8.0 / 255])
    net, nlm, _ = D.build('drunet_gray')
    net.load_state_dict(D.seeded_state_dict(net, 5))
    ref = D.Denoiser('drunet_gray', net.eval(), nlm, sigmas=sig, backend='hip_f16x3', miopen_find=False, cnn_batch=2).to('cuda')
    den = D.Denoiser('drunet_gray', net, nlm, sigmas=sig, backend='hip_f16x3', miopen_find=False, cnn_batch=2, graph=True).to('cuda')
    small, large = torch.rand(2, 1, 64, 64, device='cuda', generator=g), torch.rand(5, 1, 64, 64, device='cuda', generator=g)
    apart = _rel(ref(large, 0), ref(large, 2))
    assert apart > 1e-3, apart                                      # the noise level matters: a wrong one would show
    for i_small, i_large in ((0, 2), (2, 1), (1, 0)):
        assert _rel(den(small, i_small), ref(small, i_small)) <= 1e-5
        assert den._graphs
        assert _rel(den(large, i_large), ref(large, i_large)) <= 1e-5, (i_small, i_large)
    den(small, 2)
    den.graph = False
    assert _rel(den(small, 0), ref(small, 0)) <= 1e-5


@pytest.mark.parametrize('ch,n,H,W,dil', [(64, 3, 40, 56, 1), (64, 2, 5, 23, 3), (128, 2, 24, 24, 1), (256, 1, 17, 9, 1)])
@pytest.mark.parametrize('fmt', [1, 2, 3, 4, 5, 6, 7])
def test_f16x3_layer_with_tensors_in_the_split_activation_format(env, ch, n, H, W, dil, fmt):
    """pnp_conv3x3_nhwc_f16x3_fmt: any of x / skip / y in the split activation format ([64 hi halves][64 lo halves] per block of 64
    channels; include/pnp_mri.h).  A split INPUT carries exactly the operand values the layer would have formed itself, so the result
    equals the all-float32 call bit for bit; a split SKIP is the float32 one rounded to 2^-22; a split OUTPUT is the float32 result
    rounded to 2^-22 (an absolute 2.4e-7 of the tensor's scale at most) -- and against float64 the layer stays at the tolerance of
    the float32-format one (2e-6)."""
    torch, F, L, lib, D = env['torch'], env['F'], env['L'], env['lib'], env['D']
    g = torch.Generator(device='cuda').manual_seed(7 * ch + fmt)
    x = torch.randn(n, H, W, ch, device='cuda', generator=g)
    skip = torch.randn(n, H, W, ch, device='cuda', generator=g)
    w = torch.randn(ch, ch, 3, 3, device='cuda', generator=g) * (2.0 / (9 * ch)) ** 0.5
    bias = torch.randn(ch, device='cuda', generator=g) * 0.1
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    wp = torch.empty(9 * ch * ch, device='cuda')
    lib.check(L.pnp_conv3x3_pack_f16x3(s, p(w), p(wp), ch))
    x_q = D.unsplit_activations(D.split_activations(x))           # what a split tensor can hold
    assert _rel(x_q, x) <= 3e-7 and _rel(D.unsplit_activations(D.split_activations(x * 1e-3)), x * 1e-3) <= 3e-7
    xin = D.split_activations(x) if fmt & 1 else x
    kin = D.split_activations(skip) if fmt & 2 else skip
    y = torch.full_like(x, float('nan'))
    lib.check(L.pnp_conv3x3_nhwc_f16x3_fmt(s, p(xin), p(wp), p(bias), p(kin), p(y), n, ch, H, W, 1, dil, fmt))
    got = D.unsplit_activations(y) if fmt & 4 else y
    assert bool(torch.isfinite(got).all())
    ref = torch.relu(F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=dil, dilation=dil).permute(0, 2, 3, 1) + skip.double())
    assert _rel(got, ref) <= 2e-6, _rel(got, ref)
    # against the all-float32-format call on the values the split tensors really hold
    y0 = torch.empty_like(x)
    k_q = D.unsplit_activations(kin) if fmt & 2 else skip
    lib.check(L.pnp_conv3x3_nhwc_f16x3_fmt(s, p(x), p(wp), p(bias), p(k_q), p(y0), n, ch, H, W, 1, dil, 0))
    if fmt == 1:
        assert torch.equal(got, y0)                                # same operands, same epilogue: the same bits
    else:
        # the split-format epilogue adds the bias before the skip value, the float32 one after: one rounding apart at most -- plus, for a
        # split output, its own rounding to 2^-22
        want = D.unsplit_activations(D.split_activations(y0)) if fmt & 4 else y0
        assert float((got - want).abs().max()) <= 3.5e-7 * float(want.abs().max()) + 1e-7, float((got - want).abs().max())
    with pytest.raises(lib.PnpError):
        lib.check(L.pnp_conv3x3_nhwc_f16x3_fmt(s, p(x), p(wp), p(bias), p(kin), p(y), n, ch, H, W, 1, dil, 8))


def test_conv_check_range_knob(env, monkeypatch):
    """PNP_CONV_CHECK_RANGE=1: an activation beyond the half range entering an f16x3 layer raises at that layer (bring-up with real
    weights) instead of surfacing as NaN at the end of the network"""
    torch, D = env['torch'], env['D']
    net, nlm, _ = D.build('dncnn_15')
    sd = D.seeded_state_dict(net, 3)
    sd['model.0.weight'][0, 0, 1, 1] = 1e6
    net.load_state_dict(sd)
    den = D.Denoiser('dncnn_15', net.eval(), nlm, backend='hip_f16x3').to('cuda')
    x = torch.rand(2, 1, 32, 32, device='cuda')
    assert not bool(torch.isfinite(den(x, 0)).all())              # without the knob: loud only at the output
    monkeypatch.setenv('PNP_CONV_CHECK_RANGE', '1')
    with pytest.raises(FloatingPointError):
        den(x, 0)


@pytest.mark.parametrize('shape', [(2, 1, 64, 64), (3, 1, 31, 34), (1, 1, 17, 16), (5, 1, 128, 96)])
def test_ffdnet_with_its_input_and_output_stages_inside_the_first_and_last_layer(env, shape):
    """FFDNet under 'hip_f16x3': the replicate pad to even size, the pixel-unshuffle, the concatenation of the noise-level map
    (models/network_ffdnet.py:58-68) are folded into the first layer's kernel (pnp_ffdnet_head_nhwc), the pixel-shuffle and the crop
    (:70-73) into the last layer's (pnp_ffdnet_tail_f16x3), which writes into the caller's tensor: against the PyTorch forward with the
    same weights (<= 1e-5), odd sizes, one noise level for the batch and one per image, no PyTorch convolution or copy involved."""
    torch, D = env['torch'], env['D']
    g = torch.Generator(device='cuda').manual_seed(sum(shape))
    net, nlm, _ = D.build('ffdnet_gray')
    net.load_state_dict(D.seeded_state_dict(net, 9))
    net = net.eval().cuda()
    x = torch.rand(*shape, device='cuda', generator=g)
    for sigma in (torch.full((1, 1, 1, 1), 15 / 255., device='cuda'), torch.rand(shape[0], 1, 1, 1, device='cuda', generator=g) * 0.2):
        net.backend = 'torch'
        with torch.no_grad():
            ref = net(x, sigma).clone()
        net.backend = 'hip_f16x3'
        assert net._fused_ok(x, sigma)
        out = torch.full_like(x, float('nan'))
        orig = torch.nn.Conv2d.forward

        def boom(self, inp):
            raise AssertionError('a PyTorch convolution was called')
        torch.nn.Conv2d.forward = boom
        try:
            with torch.no_grad():
                got = net(x, sigma, out=out)
        finally:
            torch.nn.Conv2d.forward = orig
        assert got is out and bool(torch.isfinite(out).all())
        assert _rel(out, ref) <= 1e-5, _rel(out, ref)
    den = D.Denoiser('ffdnet_gray', net, nlm, backend='hip_f16x3', cnn_batch=2).to('cuda')      # batches beyond cnn_batch: slice by slice into `out`
    net.backend = 'torch'
    ref = D.Denoiser('ffdnet_gray', net, nlm, backend='torch', miopen_find=False).to('cuda')(x, 0).clone()
    for m in [net]:
        m.backend = 'hip_f16x3'
    assert _rel(den(x, 0), ref) <= 1e-5


# ---- round 6: the WIDE f16x3 kernel (csrc/kernels_conv_f16x3_wide.hip: 16 x 16 tiles, 64 x 64 wave tiles, compute + helper waves) ----
def test_wide_f16x3_kernel_is_bit_equal_to_the_narrow_one(env):
    """Both kernels issue the same products in the same order per output value; which one a launch gets depends on its size only
    (pnp_conv3x3_f16x3_set_variant: -1 by size, 0 narrow, 1 wide).  Every combination of activation formats x {skip, none} x {ReLU, none}
    x {bias, none}, on shapes with ragged edges (rows / columns that are no multiple of 16, a 5-row image, one tile exactly), C = 64 /
    128 / 256 (one, two, four chunks of input channels per item), several items per workgroup (5 x 64 x 64 on 256 compute units is
    not enough for that: the 40-image case is): BIT-equal, and the float32 -> float32 case within the layer's bar of float64 PyTorch."""
    torch, L, lib, D = env['torch'], env['L'], env['lib'], env['D']
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    prev = L.pnp_conv3x3_f16x3_set_variant(-1)
    try:
        for (nn, Cc, H, W) in ((3, 64, 37, 53), (2, 64, 16, 16), (1, 64, 5, 130), (2, 128, 40, 24), (1, 256, 17, 33), (40, 64, 96, 96)):
            g = torch.Generator(device='cuda').manual_seed(nn * 1000 + Cc)
            w = torch.randn(Cc, Cc, 3, 3, device='cuda', generator=g) * (2.0 / (9 * Cc)) ** 0.5
            wp = torch.empty(9 * Cc * Cc, device='cuda')
            lib.check(L.pnp_conv3x3_pack_f16x3(s, p(w), p(wp), Cc))
            x = torch.randn(nn, H, W, Cc, device='cuda', generator=g)
            k = torch.randn(nn, H, W, Cc, device='cuda', generator=g)
            b = torch.randn(Cc, device='cuda', generator=g) * 0.1
            xs, ks = D.split_activations(x), D.split_activations(k)
            for fmt in range(8):
                for skip in (0, 1):
                    if (fmt & 2) and not skip:
                        continue
                    for relu, bias in ((0, 0), (1, 1)) if nn == 40 else ((0, 0), (0, 1), (1, 0), (1, 1)):
                        out = []
                        for variant in (0, 1):
                            L.pnp_conv3x3_f16x3_set_variant(variant)
                            y = torch.full_like(x, 7.0)
                            lib.check(L.pnp_conv3x3_nhwc_f16x3_fmt(s, p(xs if fmt & 1 else x), p(wp), p(b) if bias else None,
                                                                   p(ks if fmt & 2 else k) if skip else None, p(y), nn, Cc, H, W, relu, 1, fmt))
                            out.append(y)
                        assert torch.equal(out[0], out[1]), ((nn, Cc, H, W), fmt, skip, relu, bias)
                        if fmt == 0 and skip and relu and bias:
                            ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)
                                             + k.permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1)
                            assert _rel(out[1], ref) <= 6e-7, _rel(out[1], ref)
    finally:
        L.pnp_conv3x3_f16x3_set_variant(prev)
