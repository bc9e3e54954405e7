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
    old = C.set_precision('auto')
    try:
        assert C.precision_name(C.resolve_auto('stylegan2', 256)) == 'mixed'
        assert C.precision_name(C.resolve_auto('stylegan2', 1024)) == 'bf16x3'        # f16 / f16x2 miss the gate at 1024^2
        assert C.precision_name(C.resolve_auto('proggan', 256)) == 'f16'
        assert C.precision_name(C.resolve_auto('biggan', 128)) == 'bf16x3'
        assert C.precision_name(C.resolve_auto('sngan', 32)) == C.AUTO_FALLBACK
        C.set_precision('fp32')
        assert C.resolve_auto('stylegan2', 256) == 0                                    # an explicit mode wins everywhere
    finally:
        C.PRECISION = old


def test_mixed_policy_per_layer():
    M = C.MIXED
    # forward: below 64 x 64 split-bf16, stride-1 convs fp16, up-sampling layers fp16 x2
    assert C.layer_precision(M, 32, False) == 1 and C.layer_precision(M, 32, True) == 1
    assert C.layer_precision(M, 64, False) == 2 and C.layer_precision(M, 256, False) == 2
    assert C.layer_precision(M, 64, True) == 3 and C.layer_precision(M, 256, True) == 3
    # backward: the up-sampling layers' input-gradient convs in plain fp16, everything else as the forward
    assert C.layer_precision_bwd(M, 128, True) == 2
    assert C.layer_precision_bwd(M, 128, False) == 2 and C.layer_precision_bwd(M, 16, True) == 1
    # any fixed mode is the same for every layer, forward and backward
    for code in (0, 1, 2, 3):
        assert C.layer_precision(code, 8, True) == code and C.layer_precision_bwd(code, 256, True) == code


def test_fused_upconv_selection():
    assert C.upconv_fused_ok(128, 256, 128, 3) and C.upconv_fused_ok(32, 512, 512, 2)
    assert not C.upconv_fused_ok(128, 256, 128, 1)          # split-bf16 keeps the phase GEMMs + blur kernel
    assert not C.upconv_fused_ok(128, 256, 128, 0)
    assert not C.upconv_fused_ok(8, 512, 512, 2)            # maps below 16 x 16
    assert not C.upconv_fused_ok(512, 64, 32, 2)            # 32 output channels: not a multiple of the 64-column tile
    assert not C.upconv_fused_ok(64, 24, 64, 2)


def test_reconstructor_auto_forward_mode_follows_the_generator(monkeypatch):
    """R_PRECISION 'auto': split-bf16 forward convs inside a step whose generator ran in a 16-bit mode, exact fp32 for an fp32
    generator and for a Reconstructor used on its own; 'fp32' / 'bf16x3' pin it."""
    from warpedganspace_amd import reconstructor as RR
    monkeypatch.setattr(RR, 'R_PRECISION', 'auto')
    assert RR.forward_precision() == 0 and RR.forward_precision(None) == 0 and RR.forward_precision(0) == 0
    for code in (1, 2, 3, 4):
        assert RR.forward_precision(code) == 1
    monkeypatch.setattr(RR, 'R_PRECISION', 'fp32')
    assert all(RR.forward_precision(c) == 0 for c in (None, 0, 1, 2, 3, 4))
    monkeypatch.setattr(RR, 'R_PRECISION', 'bf16x3')
    assert all(RR.forward_precision(c) == 1 for c in (None, 0, 1, 2, 3, 4))


def test_last_resolved_tracks_the_generator_context():
    old = C.last_resolved()
    with C.resolved(3):
        assert C.PRECISION == 3
    assert C.last_resolved() == 3
    with C.resolved(0):
        pass
    assert C.last_resolved() == 0
    C._LAST_RESOLVED = old


def test_plane_route_gates():
    # transposed-blur plane: plain fp16 gradient launches that fill the chip with 256-row tiles
    assert C.blur_bwd_f16_ok(32, 256, 128, 256, 2) and C.blur_bwd_f16_ok(32, 64, 512, 512, 2)
    assert not C.blur_bwd_f16_ok(32, 256, 128, 256, 3) and not C.blur_bwd_f16_ok(32, 256, 128, 256, 1)
    assert not C.blur_bwd_f16_ok(2, 32, 512, 512, 2) and not C.blur_bwd_f16_ok(32, 256, 24, 256, 2)
    # dy plane of the stride-1 layers: from 256 channels up (the 128-column DMA tile loses to the patch form)
    assert C.dy_plane_ok(32, 128, 256, 256, 2) and C.dy_plane_ok(32, 64, 512, 512, 2)
    assert not C.dy_plane_ok(32, 256, 128, 128, 2) and not C.dy_plane_ok(32, 128, 256, 256, 1)
