"""GPU suite: the persistent tile loops of the matrix-core conv kernels against a float64 oracle.

A conv workgroup walks a contiguous RANGE of 16x16-pixel tiles (forward / data gradient) or of
8x32 / 4x32-pixel tiles (weight gradient): the next tile's halo is requested while the current one
computes, tile coordinates advance by carries across rows and images, the ranges of a launch are
uneven, and the per-channel sums accumulate across a workgroup's tiles.  The per-kernel tests with
small tensors give every workgroup ONE tile; here every workgroup gets several, two ways:

* shapes of the benchmark's own size class, launched as the library sizes them (16 384-tile
  layers over <= 1024 workgroups in the benchmark; 1200-1984 tiles over 256-768 here);
* small shapes with the workgroup count forced down through mis_set_conv_launch (3 and 7
  workgroups over 12-60 tiles: long, ragged ranges that cross rows and images), where the CPU
  oracle costs nothing and every kernel form can be swept.

Oracle: torch.nn.functional.conv2d in float64 on the dtype-rounded operands (the arithmetic of
model/utils/modules.py:86-99's nn.Conv2d).  Bounds: float32 1e-5 of the tensor's scale, bfloat16
1e-2; sums a few 1e-4 (long float32 reductions of bf16-rounded products on the kernel side).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import datagen as dg

pytestmark = pytest.mark.gpu

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]
conv2d = torch.nn.functional.conv2d


@pytest.fixture(scope="module")
def F_():
    if not torch.cuda.is_available():
        pytest.fail("GPU suite needs a HIP device")
    from medical_image_segmentation_amd import _lib, functional
    _lib.load()
    return functional


@pytest.fixture
def launch():
    """launch(n): force n workgroups (None = the library's own sizing); always reset afterwards."""
    from medical_image_segmentation_amd import _lib

    def _set(n):
        _lib.call("mis_set_conv_launch", n or 0, n or 0)

    try:
        yield _set
    finally:
        _lib.call("mis_set_conv_launch", 0, 0)


def act(a, dtype):
    t = a if torch.is_tensor(a) else T(a)
    return t.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)


def q(a, dtype):
    t = a if torch.is_tensor(a) else T(a)
    return t.to(dtype).to(torch.float32)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def lazy_pair(C, seed):
    sc = dg.hash_uniform((C,), seed, 0.4, 1.6) * np.where(dg.hash_uniform((C,), seed + 1, 0, 1) < 0.25, -1.0, 1.0)
    sh = dg.hash_uniform((C,), seed + 2, -0.5, 0.5)
    return T(np.stack([sc, sh]).astype(np.float32))


def lazy_value(z, ss, dtype):
    """relu(z*scale + shift) from the dtype-rounded z, rounded to the dtype again (the matrix-core
    kernels use the applied activation as an operand of the compute dtype)."""
    zq = q(z, dtype).double()
    a = zq * ss[0].double()[None, :, None, None] + ss[1].double()[None, :, None, None]
    return q(torch.relu(a).float(), dtype)


def taken(dtype, Ci):
    G = 8 if dtype == torch.bfloat16 else 4
    return Ci % G == 0 and (Ci <= 4 * G or Ci % (4 * G) == 0)


# B, Cin, Cout, H, W — sizes of the benchmark's class, launched as the library sizes them
BIG = [(2, 32, 16, 384, 392), (2, 8, 32, 512, 496), (1, 64, 8, 512, 512), (4, 128, 64, 160, 176)]
# small tensors for forced launches: one- and multi-chunk K loops, every output-slice width, the
# wave-specialised form (32-channel chunks with >= 32 output channels), short-K forms (8/16/24 channels)
SMALL = [(2, 32, 16, 40, 56), (2, 8, 32, 48, 40), (1, 64, 8, 64, 48), (3, 128, 64, 24, 40), (2, 32, 64, 40, 24),
         (2, 16, 64, 36, 52), (1, 24, 8, 50, 30), (2, 64, 128, 20, 36)]
CASES = [(s, None) for s in BIG] + [(s, n) for s in SMALL for n in (3, 7)]
IDS = [f"{s[0]}x{s[1]}to{s[2]}@{s[3]}x{s[4]}-wg{n or 'auto'}" for s, n in CASES]


def _check_conv(F_, z, stats, zo, dtype):
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(z.float(), zo) < tol
    slack = 1e-4 + (0 if dtype == torch.float32 else 1e-3)
    assert rel_err(stats[0], zo.sum(dim=(0, 2, 3))) < slack
    assert rel_err(stats[1], (zo * zo).sum(dim=(0, 2, 3))) < slack


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("transposed", [False, True], ids=["fwd", "dgrad"])
@pytest.mark.parametrize("lazy", [False, True], ids=["plain", "lazy"])
def test_conv3x3_tile_ranges(F_, launch, dtype, case, transposed, lazy):
    """mis_conv3x3_fwd with the forward operand and with the data-gradient operand (W^T, taps
    mirrored), plain and lazy input: z, sum z, sum z^2."""
    (B, Ci, Co, H, W), n = case
    if not taken(dtype, Ci):
        pytest.skip("shape not taken by the matrix-core kernel in this dtype")
    x = dg.hash_normalish((B, Ci, H, W), 2701)
    wm = dg.hash_normalish((Ci, Co, 3, 3) if transposed else (Co, Ci, 3, 3), 2702, std=(2.0 / (9 * Ci)) ** 0.5)
    b = dg.hash_uniform((Co,), 2703, -0.1, 0.1)
    ss = lazy_pair(Ci, 2710) if lazy else None
    xv = lazy_value(x, ss, dtype) if lazy else q(x, dtype)
    launch(n)
    wp = F_.pack_weights(T(wm).to(DEV), dtype, transposed=transposed)
    z, stats = F_.conv3x3_raw(act(x, dtype), wp, None if transposed else T(b).to(DEV), Co, True,
                              ss.to(DEV) if lazy else None)
    torch.cuda.synchronize()
    wq = q(wm, dtype).double()
    w_eff = wq.flip(2, 3).transpose(0, 1) if transposed else wq
    zo = conv2d(xv.double(), w_eff, None if transposed else T(b).double(), padding=1)
    _check_conv(F_, z, stats.cpu(), zo, dtype)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_dgrad_bn_sums_tile_ranges(F_, launch, dtype, case):
    """mis_conv3x3_dgrad_bn_sums: da = conv(dz, W^T) and, from the same kernel, the BatchNorm-backward
    sums of the layer whose lazy output the conv consumed: sum da*[y>0] and sum da*[y>0]*xhat with
    y = z_prev*scale + shift, xhat = (z_prev - mean)*rstd, `da` as stored (rounded to the dtype)."""
    from medical_image_segmentation_amd import _lib
    (B, Ci, Co, H, W), n = case         # Ci = channels of dz, Co = channels of da and z_prev
    if not taken(dtype, Ci):
        pytest.skip("shape not taken by the matrix-core kernel in this dtype")
    code = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
    if not _lib.query("mis_conv3x3_dgrad_bn_sums_supported", code, B, H, W, Ci, Co):
        pytest.skip("fused-sums data gradient not offered for this shape")
    dz = dg.hash_normalish((B, Ci, H, W), 2801)
    zp = dg.hash_normalish((B, Co, H, W), 2802)
    wm = dg.hash_normalish((Ci, Co, 3, 3), 2803, std=(2.0 / (9 * Ci)) ** 0.5)      # master weights of the forward conv Co -> Ci
    aff = np.stack([dg.hash_uniform((Co,), 2804, 0.5, 1.5) * np.where(dg.hash_uniform((Co,), 2805, 0, 1) < 0.3, -1, 1),
                    dg.hash_uniform((Co,), 2806, -0.4, 0.4), dg.hash_uniform((Co,), 2807, -0.2, 0.2),
                    dg.hash_uniform((Co,), 2808, 0.5, 2.0)]).astype(np.float32)
    launch(n)
    wt = F_.pack_weights(T(wm).to(DEV), dtype, transposed=True)
    dzd, zpd, affd = act(dz, dtype), act(zp, dtype), T(aff).to(DEV)
    da = torch.empty_like(zpd)
    sums = torch.empty((2, Co), dtype=torch.float32, device=DEV)
    ws = F_.workspace(_lib.query("mis_conv3x3_workspace_bytes", B, H, W, Ci, Co), zpd.device)
    _lib.call("mis_conv3x3_dgrad_bn_sums", dzd.data_ptr(), wt.data_ptr(), code, B, H, W, Ci, Co, da.data_ptr(),
              zpd.data_ptr(), affd.data_ptr(), sums.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    w_eff = q(wm, dtype).double().flip(2, 3).transpose(0, 1)
    dao = conv2d(q(dz, dtype).double(), w_eff, None, padding=1)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(da.float(), dao) < tol
    # the sums are taken from the values as the kernel stored them
    das = da.float().cpu().double()
    zq = q(zp, dtype)
    a4 = T(aff)
    on = (zq * a4[0][None, :, None, None] + a4[1][None, :, None, None]) > 0          # float32, like the kernel
    xhat = (zq.double() - a4[2].double()[None, :, None, None]) * a4[3].double()[None, :, None, None]
    s0 = (das * on).sum(dim=(0, 2, 3))
    s1 = (das * on * xhat).sum(dim=(0, 2, 3))
    assert rel_err(sums[0], s0) < 2e-4
    assert rel_err(sums[1], s1) < 2e-4


CAT_BIG = [(2, 32, 8, 512, 256), (2, 64, 32, 160, 176)]          # B, Ca (= Cb), Cout, H, W
CAT_SMALL = [(2, 32, 8, 40, 56), (2, 64, 32, 24, 40), (1, 32, 64, 48, 40), (2, 128, 16, 20, 36)]
CAT_CASES = [(s, None) for s in CAT_BIG] + [(s, n) for s in CAT_SMALL for n in (3, 7)]
CAT_IDS = [f"{s[0]}x{s[1]}+{s[1]}to{s[2]}@{s[3]}x{s[4]}-wg{n or 'auto'}" for s, n in CAT_CASES]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CAT_CASES, ids=CAT_IDS)
def test_virtual_concat_tile_ranges(F_, launch, dtype, case):
    """mis_conv3x3_cat_fwd (two input tensors, the second lazy), mis_conv3x3_split_fwd (two output
    tensors) and mis_conv3x3_cat_wgrad of the decoder's first conv (model/unet.py:66-68)."""
    (B, Ca, Co, H, W), n = case
    xa, xb = dg.hash_normalish((B, Ca, H, W), 2901), dg.hash_normalish((B, Ca, H, W), 2902)
    w = dg.hash_normalish((Co, 2 * Ca, 3, 3), 2903, std=(2.0 / (18 * Ca)) ** 0.5)
    b = dg.hash_uniform((Co,), 2904, -0.1, 0.1)
    gz = dg.hash_normalish((B, Co, H, W), 2905)
    ssb = lazy_pair(Ca, 2920)
    xad, xbd, gd = act(xa, dtype), act(xb, dtype), act(gz, dtype)
    if not F_.cat_supported(xad, xbd, Co):
        pytest.skip("shape not taken by the two-tensor kernels in this dtype")
    launch(n)
    wd = T(w).to(DEV)
    wf, wt = F_.pack_weights(wd, dtype), F_.pack_weights(wd, dtype, transposed=True)
    z, st = F_.conv3x3_raw(xad, wf, T(b).to(DEV), Co, True, None, xbd, ssb.to(DEV))
    da, db = F_.conv3x3_split_raw(gd, wt, Ca, Ca)
    dw = F_.conv3x3_wgrad_raw(xad, gd, None, xbd, ssb.to(DEV))
    torch.cuda.synchronize()
    cat = torch.cat([q(xa, dtype), lazy_value(xb, ssb, dtype)], dim=1).double().requires_grad_(True)
    wo = q(w, dtype).double().requires_grad_(True)
    zo = conv2d(cat, wo, T(b).double(), padding=1)
    zo.backward(q(gz, dtype).double())
    _check_conv(F_, z, st.cpu(), zo.detach(), dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(torch.cat([da, db], dim=1).float(), cat.grad) < tol
    assert rel_err(dw, wo.grad) < (2e-5 if dtype == torch.float32 else 3e-3)


DIL_CASES = [((2, 32, 32, 256, 264), None, 2), ((1, 32, 128, 192, 200), None, 3)] + \
            [(s, n, d) for s in [(2, 32, 32, 40, 56), (1, 32, 128, 36, 52), (2, 16, 64, 30, 44)] for n in (3, 7) for d in (2, 3)]
DIL_IDS = [f"{s[0]}x{s[1]}to{s[2]}@{s[3]}x{s[4]}-d{d}-wg{n or 'auto'}" for s, n, d in DIL_CASES]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", DIL_CASES, ids=DIL_IDS)
def test_dilated_conv_tile_ranges(F_, launch, dtype, case):
    """mis_conv3x3_dilated_fwd (forward and data-gradient operands) and mis_conv3x3_dilated_wgrad
    (ResDConv's dilation 2 / 3 convs, model/utils/modules.py:436-445)."""
    from medical_image_segmentation_amd import _lib
    (B, Ci, Co, H, W), n, dil = case
    code = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
    if not _lib.query("mis_conv3x3_dilated_supported", code, B, H, W, Ci, Co, dil):
        pytest.skip("shape not taken by the dilated matrix-core kernels in this dtype")
    x = dg.hash_normalish((B, Ci, H, W), 3001)
    w = dg.hash_normalish((Co, Ci, 3, 3), 3002, std=(2.0 / (9 * Ci)) ** 0.5)
    b = dg.hash_uniform((Co,), 3003, -0.1, 0.1)
    gz = dg.hash_normalish((B, Co, H, W), 3004)
    ss = lazy_pair(Ci, 3010)
    launch(n)
    wd = T(w).to(DEV)
    wf, wt = F_.pack_weights(wd, dtype), F_.pack_weights(wd, dtype, transposed=True)
    xd, gd = act(x, dtype), act(gz, dtype)
    z, st = F_.conv3x3_raw(xd, wf, T(b).to(DEV), Co, True, ss.to(DEV), dilation=dil)
    dx, _ = F_.conv3x3_raw(gd, wt, None, Ci, False, dilation=dil)
    dw = F_.conv3x3_wgrad_raw(xd, gd, ss.to(DEV), dilation=dil)
    torch.cuda.synchronize()
    xo = lazy_value(x, ss, dtype).double().requires_grad_(True)
    wo = q(w, dtype).double().requires_grad_(True)
    zo = conv2d(xo, wo, T(b).double(), padding=dil, dilation=dil)
    zo.backward(q(gz, dtype).double())
    _check_conv(F_, z, st.cpu(), zo.detach(), dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(dx.float(), xo.grad) < tol
    assert rel_err(dw, wo.grad) < (2e-5 if dtype == torch.float32 else 3e-3)


WG_CASES = [((2, 8, 32, 512, 496), None), ((1, 64, 8, 512, 512), None), ((4, 128, 64, 160, 176), None),
            ((2, 32, 32, 256, 264), None)] + \
           [(s, n) for s in [(2, 8, 32, 40, 72), (1, 64, 8, 64, 96), (3, 128, 64, 24, 40), (2, 32, 32, 40, 56),
                             (2, 16, 16, 36, 52), (1, 64, 256, 20, 36)] for n in (3, 8, 24)]
WG_IDS = [f"{s[0]}x{s[1]}to{s[2]}@{s[3]}x{s[4]}-wg{n or 'auto'}" for s, n in WG_CASES]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", WG_CASES, ids=WG_IDS)
@pytest.mark.parametrize("lazy", [False, True], ids=["plain", "lazy"])
def test_wgrad_tile_ranges(F_, launch, dtype, case, lazy):
    """mis_conv3x3_wgrad: every K-split workgroup accumulates several pixel tiles (both the
    single-role and the wave-specialised forms), slabs reduced in a fixed order."""
    (B, Ci, Co, H, W), n = case
    G = 8 if dtype == torch.bfloat16 else 4
    if Ci % G or Co % G:
        pytest.skip("shape not taken by the matrix-core kernel in this dtype")
    x = dg.hash_normalish((B, Ci, H, W), 3101)
    gz = dg.hash_normalish((B, Co, H, W), 3102)
    ss = lazy_pair(Ci, 3110) if lazy else None
    xv = lazy_value(x, ss, dtype) if lazy else q(x, dtype)
    launch(n)
    dw = F_.conv3x3_wgrad_raw(act(x, dtype), act(gz, dtype), ss.to(DEV) if lazy else None)
    torch.cuda.synchronize()
    wo = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    conv2d(xv.double(), wo, None, padding=1).backward(q(gz, dtype).double())
    assert rel_err(dw, wo.grad) < 2e-5


def test_conv_launch_hook_rejects_bad_counts(F_):
    from medical_image_segmentation_amd import _lib
    with pytest.raises(RuntimeError, match="0..8192"):
        _lib.call("mis_set_conv_launch", -1, 0)
    with pytest.raises(RuntimeError, match="0..8192"):
        _lib.call("mis_set_conv_launch", 0, 100000)
    _lib.call("mis_set_conv_launch", 0, 0)
