"""CPU: host-side arithmetic policy of the generator's conv launches (warpedganspace_amd/conv.py) — which mode a layer runs in
under 'auto' / 'mixed', forward and backward, and which launches take the fused up-sampling kernel.  No kernel runs here."""
import pytest

from warpedganspace_amd import conv as C


def test_precision_names_round_trip():
    for name, code in C.PRECISION_NAMES.items():
        assert C.precision_code(name) == code
        assert C.precision_name(code) == name
    with pytest.raises(Exception):
        C.precision_code('fp8')


def test_auto_resolves_per_architecture():
    # StyleGAN2: the per-layer table CALIBRATED by the step engine on its own generator (round 6; the fixed 'mixed' table stays an explicit mode)
    assert C.precision_name(C.resolve('auto', 'stylegan2', 256)) == 'mixed-strict'
    assert C.precision_name(C.resolve(None, 'stylegan2', 256)) == 'mixed-strict'         # None = auto
    assert C.precision_name(C.resolve('auto', 'stylegan2', 1024)) == 'mixed-strict'
    assert C.precision_name(C.resolve('auto', 'proggan', 256)) == 'f16'
    assert C.precision_name(C.resolve('auto', 'biggan', 128)) == 'bf16x3'
    assert C.precision_name(C.resolve('auto', 'sngan', 32)) == C.AUTO_FALLBACK
    assert C.precision_name(C.resolve('auto', 'stylegan2', 64)) == 'bf16x3w'            # no measurement behind it: fp32-class (StyleGAN2: with the F(2,3) form)
    assert C.resolve('fp32', 'stylegan2', 256) == 0 and C.resolve(3, 'proggan', 256) == 3   # an explicit mode wins everywhere
    for key, name in C.AUTO_TABLE.items():
        assert name in C.PRECISION_NAMES and name != 'auto', key


def test_no_process_wide_arithmetic_state():
    """The arithmetic is an attribute / argument of generators and step engines: the module holds no mutable mode."""
    for gone in ('PRECISION', 'set_precision', 'resolved', 'last_resolved', '_LAST_RESOLVED', 'grad_operands', '_GRAD_CTX'):
        assert not hasattr(C, gone), gone
    from warpedganspace_amd import reconstructor as RR
    for gone in ('R_PRECISION', 'R_DGRAD_PRECISION', 'R_WGRAD_PRECISION', 'forward_precision'):
        assert not hasattr(RR, gone), gone
    from warpedganspace_amd.stylegan2 import Generator
    a, b = Generator(32, 512, 2), Generator(32, 512, 2)
    a.precision = 'f16'
    assert a.resolve_precision() == 2 and b.resolve_precision() == 0 and b.resolve_precision('mixed') == 4
    assert Generator(256, 512, 2).resolve_precision('auto') == C.MIXED_STRICT


def test_mixed_policy_per_layer():
    M, W16 = C.MIXED, C.BF16W
    # StyleGAN2-256 (the default table): below 128 x 128 split-bf16 (stride-1 convs in the F(2,3) form where the kernel covers them: code 7
    # falls back to a direct split-bf16 launch elsewhere), stride-1 convs fp16, up-sampling layers fp16 x2
    assert C.layer_precision(M, 64, False) == W16 and C.layer_precision(M, 64, True) == 1 and C.layer_precision(M, 32, True) == 1 and C.layer_precision(M, 8, False) == W16
    assert C.layer_precision(M, 128, False) == 2 and C.layer_precision(M, 256, False) == 2
    assert C.layer_precision(M, 128, True) == 3 and C.layer_precision(M, 256, True) == 3
    # backward: the up-sampling layers' input-gradient convs in plain fp16, everything else as the forward
    assert C.layer_precision_bwd(M, 128, True) == 2
    assert C.layer_precision_bwd(M, 128, False) == 2 and C.layer_precision_bwd(M, 16, True) == 1
    # ... and the two 64 x 64 layers: split-bf16 forward (image gate), plain fp16 input-gradient convs (gradient gate)
    assert C.layer_precision_bwd(M, 64, False) == 2 and C.layer_precision_bwd(M, 64, True) == 2 and C.layer_precision_bwd(M, 32, False) == W16
    # StyleGAN2-1024: fp16 x2 in the HBM-bound 512^2 / 1024^2 layers only
    pol = C.mixed_policy(1024)
    assert pol is C.MIXED_1024 and C.mixed_policy(256) is C.MIXED_256
    assert C.layer_precision(M, 1024, False, pol) == 3 and C.layer_precision(M, 512, True, pol) == 3
    assert C.layer_precision(M, 256, False, pol) == W16 and C.layer_precision(M, 64, True, pol) == 1
    assert C.layer_precision_bwd(M, 1024, True, pol) == 2 and C.layer_precision_bwd(M, 1024, False, pol) == 3
    assert C.layer_precision_bwd(M, 256, False, pol) == 2 and C.layer_precision_bwd(M, 64, True, pol) == 2 and C.layer_precision_bwd(M, 32, True, pol) == 1
    # an explicit table, and a policy without the plain-fp16 backward rule
    p2 = C.MixedPolicy({64: (3, 2)}, below=0, bwd_up_f16=False)
    assert p2.fwd(64, False) == 3 and p2.fwd(64, True) == 2 and p2.fwd(8, True) == 0 and p2.bwd(64, True) == 2
    assert C.MixedPolicy({64: (2, 3)}, bwd_up_f16=False).bwd(64, True) == 3
    # any fixed mode is the same for every layer, forward and backward
    for code in (0, 1, 2, 3):
        assert C.layer_precision(code, 8, True) == code and C.layer_precision_bwd(code, 256, True) == code
    # 'bf16x3w': split-bf16 everywhere, the stride-1 layers carry code 7 to conv.launch (which routes what conv_wino_bf16.hip covers)
    for res in (4, 64, 256):
        assert C.layer_precision(W16, res, False) == W16 and C.layer_precision(W16, res, True) == 1
        assert C.layer_precision_bwd(W16, res, False) == W16 and C.layer_precision_bwd(W16, res, True) == 1
    assert C.is_reduced(W16) and not C.is_f16_operand(W16)
    # the strict ladder (ordered by measured step time): starts at the default table, ends without any fp16-rounded layer; distinct names
    for size, ladder in C.STRICT_LADDER.items():
        sp = [pol_.spends() for _, pol_ in ladder]
        assert ladder[0][1] is C.MIXED_POLICIES[size] and sp[0] == max(sp) and sp[-1] == 0, (size, sp)
        assert len({n_ for n_, _ in ladder}) == len(ladder)
        assert C.mixed_policy(size, C.MIXED_STRICT) in [pol_ for _, pol_ in ladder]


def test_fused_upconv_selection():
    assert C.upconv_fused_ok(128, 256, 128, 3) and C.upconv_fused_ok(32, 512, 512, 2)
    assert C.upconv_fused_ok(128, 256, 128, 1) and C.upconv_fused_ok(32, 512, 512, 1)       # round 5: split-bf16 through the fused kernel from 32 x 32 inputs
    assert not C.upconv_fused_ok(16, 512, 512, 1)           # ... below that the phase GEMMs + blur kernel
    assert not C.upconv_fused_ok(128, 256, 128, 0)
    assert not C.upconv_fused_ok(8, 512, 512, 2)            # maps below 16 x 16
    assert C.upconv_fused_ok(512, 64, 32, 2)                # 32 output channels: one half-filled 64-column tile (StyleGAN2-1024's last up-sampling layer)
    assert not C.upconv_fused_ok(512, 32, 16, 2) and not C.upconv_fused_ok(512, 64, 96, 2)
    assert not C.upconv_fused_ok(64, 24, 64, 2)


def test_fp32w_is_fp32_with_the_winograd_form_on_the_stride_1_layers_only():
    """'fp32w': the 3x3 stride-1 layers (forward and input-gradient) carry code 5 to conv.launch, which routes what the Winograd
    kernel covers; up-sampling layers, and everything conv._desc hands to the library, are plain exact fp32 (code 0)."""
    W = C.FP32W
    assert C.precision_code('fp32w') == W and C.resolve('fp32w', 'stylegan2', 256) == W
    for res in (4, 64, 256, 1024):
        assert C.layer_precision(W, res, False) == W and C.layer_precision(W, res, True) == 0
        assert C.layer_precision_bwd(W, res, False) == W and C.layer_precision_bwd(W, res, True) == 0
    assert C.SplitCache(None).get(W) is None            # no 16-bit planes: U is kept under its own key
    assert not C.upconv_fused_ok(128, 256, 128, W) and not C.upconv_fused_ok(128, 256, 128, 0)
    from warpedganspace_amd import reconstructor as RR
    assert RR.r_arith('auto', W) == RR.R_FP32_WINO == RR.RArith(5, 5, 0) and RR.r_arith('fp32w') == RR.R_FP32_WINO
    assert RR.r_arith('fp32', W) == RR.R_EXACT


def test_reconstructor_arithmetic_follows_the_generator():
    """r_precision 'auto': fp32-class (split-bf16 x3) convs inside a step whose generator runs in a 16-bit mode, the reference's
    exact fp32 for an fp32 generator and for a Reconstructor used on its own; 'fp32' / 'bf16x3' pin it; an RArith passes through."""
    from warpedganspace_amd import reconstructor as RR
    assert RR.r_arith('auto') == RR.R_EXACT and RR.r_arith('auto', None) == RR.R_EXACT and RR.r_arith('auto', 0) == RR.R_EXACT
    for code in (1, 2, 3, 4):
        assert RR.r_arith('auto', code) == RR.R_FP32_CLASS
    assert all(RR.r_arith('fp32', c) == RR.R_EXACT for c in (None, 0, 1, 2, 3, 4))
    assert all(RR.r_arith('bf16x3', c) == RR.R_FP32_CLASS for c in (None, 0, 1, 2, 3, 4))
    hybrid = RR.RArith(forward=0, dgrad=1, wgrad=1)
    assert RR.r_arith(hybrid, 2) is hybrid
    with pytest.raises(Exception):
        RR.r_arith('fp8')
    assert RR.Reconstructor('LeNet', 4, channels=1).arith == RR.R_EXACT


def test_plane_route_gates():
    # transposed-blur plane: plain fp16 gradient launches that fill the chip with 256-row tiles
    assert C.blur_bwd_f16_ok(32, 256, 128, 256, 2) and C.blur_bwd_f16_ok(32, 64, 512, 512, 2)
    assert not C.blur_bwd_f16_ok(32, 256, 128, 256, 3) and not C.blur_bwd_f16_ok(32, 256, 128, 256, 1)
    assert not C.blur_bwd_f16_ok(2, 32, 512, 512, 2) and not C.blur_bwd_f16_ok(32, 256, 24, 256, 2)
    # dy plane of the stride-1 layers: from 128 channels up since round 4 (a 128-column plane goes through the patch kernel's XF16 form;
    # round 3 had only the LDS-DMA kernel for planes, whose 128-column tile loses to the patch form)
    assert C.dy_plane_ok(32, 128, 256, 256, 2) and C.dy_plane_ok(32, 64, 512, 512, 2) and C.dy_plane_ok(32, 256, 128, 128, 2)
    assert not C.dy_plane_ok(32, 512, 64, 64, 2) and not C.dy_plane_ok(32, 128, 256, 256, 1)
    # forward planes: the up-conv's output for a plain-fp16 stride-1 conv with >= 128 output columns that fills the chip
    assert C.fwd_plane_ok(32, 256, 128, 128, 2) and C.fwd_plane_ok(32, 128, 256, 256, 2)
    assert not C.fwd_plane_ok(32, 256, 128, 128, 3) and not C.fwd_plane_ok(32, 512, 64, 64, 2) and not C.fwd_plane_ok(1, 16, 512, 512, 2)
    # a plane must stay addressable through one buffer descriptor (< 2^31 bytes at 2 bytes per element, which is how the library
    # sizes it): per-GPU batch 64 at 256^2 is fine ([64, 257, 257, 128] fp16 = 1.08e9 bytes), batch 128 keeps the fp32 route
    assert C.blur_bwd_f16_ok(64, 256, 128, 256, 2) and not C.blur_bwd_f16_ok(128, 256, 128, 256, 2)
    assert C.dy_plane_ok(64, 128, 256, 256, 2) and not C.dy_plane_ok(512, 128, 256, 256, 2)
